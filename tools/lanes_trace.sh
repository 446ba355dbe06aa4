#!/bin/bash
# GPU box: kernel timelines of the 128-pair step with an RCCL communicator created BEFORE the engine (TM_BENCH_FORCE_DIST=1: a one-rank
# process group, what bench.py --gpus N does on every rank), with and without the hardware-queue lanes -> gpurun_out/<TAG>_lanes{1,0}_timeline.txt
set -u
TAG=${1:-r06y15}
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 TM_BENCH_FORCE_DIST=1
R=$GRAFT_REPO_ROOT
for q in 1 0; do
  export TM_QUEUE_LANES=$q
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_lanes${q}_prof -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-extras > $R/gpurun_out/${TAG}_lanes${q}_bench.log 2>&1
  cd $R
  python3 tools/trace_timeline.py "$(ls -t gpurun_out/${TAG}_lanes${q}_prof/*/*_kernel_trace.csv | head -1)" 2 > gpurun_out/${TAG}_lanes${q}_timeline.txt
  grep '^{' gpurun_out/${TAG}_lanes${q}_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('TM_QUEUE_LANES=$q', d['value'], d['summary'].get('stage_ms'))"
done
