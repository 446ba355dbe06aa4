import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{"metric"')][-1]
print(len(l)); d = json.loads(l); print(d["summary"]["batch_curve"]); print(d["summary"].get("batch_curve_in_flight")); print(d["summary"].get("leg_errors"))
