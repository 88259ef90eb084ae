#!/bin/bash
# GPU box with ONE GPU: bench.py's N > 1 entry as N rank processes on that GPU (HOMER_BENCH_ONE_DEVICE=1: the ring goes over gloo and page-locked host buffers instead of
# RCCL) - started by bench.py itself (--gpus N without a launcher).  Checks the ring's code path and every access unit, not its speed.   tools/ring_one_device.sh [tag]
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
mkdir -p gpurun_out
for N in 2 4; do
  HOMER_BENCH_ONE_DEVICE=1 timeout 900 python3 bench.py --gpus $N --sequences 24 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_ring_one_device_n$N.json 2> gpurun_out/${TAG}_ring_one_device_n$N.err
  echo "N=$N rc=$?"; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${TAG}_ring_one_device_n$N.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['config']['sequences_per_gpu'], d['stream_matches_reference'], d['access_units_checked_against_reference'], d['access_units_differing'], d['roofline'] and d['roofline']['ms_per_launch'])" || tail -5 gpurun_out/${TAG}_ring_one_device_n$N.err
done
