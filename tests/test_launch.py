"""The rank launcher behind `python bench.py --gpus N` (turbo-metrics_amd/launch.py), on CPU: N children with the torchrun
environment, rank 0's stdout forwarded, failures propagated, a dead rank does not leave the others hanging."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body, n, tmp_path, timeout_s=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    drv = tmp_path / "drv.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from tm_pkg import tm
        assert "torch" not in sys.modules  # the launcher must not pull in torch (nothing may touch the GPU before the ranks start)
        sys.exit(tm.launch.spawn_ranks([sys.executable, {str(script)!r}], {n}, timeout_s={timeout_s!r}))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=120, env=env)


def test_ranks_get_the_torchrun_environment_and_rank0_stdout_is_forwarded(tmp_path):
    r = _run("""
        import os
        print("rank", os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) > 0)
    """, 3, tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "rank 0 0 3 127.0.0.1 True"  # ranks 1, 2 print to /dev/null: ONE line reaches the caller


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    r = _run("""
        import os, torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
        dist.reduce(t, dst=0)
        if dist.get_rank() == 0: print("sum", t.item(), dist.get_world_size())
        dist.destroy_process_group()
    """, 2, tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == "sum 3.0 2"  # (gloo itself prints a connection line on stdout)


def test_failing_rank_fails_the_launch_and_stops_the_others(tmp_path):
    r = _run("""
        import os, sys, time
        if os.environ["RANK"] == "1": sys.exit(7)
        time.sleep(60)  # would hang the launch if the dead rank went unnoticed
    """, 2, tmp_path)
    assert r.returncode == 7
    assert "a rank failed" in r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in r.stderr
