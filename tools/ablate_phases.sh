#!/bin/bash
cd $GRAFT_REPO_ROOT
cp homerhevc_amd/libhomer_gpu.so /tmp/prod.so
for n in ${VARIANTS:-cur noi nome noboth}; do
  cp build/variants/$n/libhomer_gpu.so homerhevc_amd/libhomer_gpu.so
  timeout 300 python3 bench.py --sequences 1 --steps 8 --warmup 3 --no-cpu-baseline --no-single-thread-order > gpurun_out/abl_$n.json 2>/dev/null
  python3 - $n <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/abl_{n}.json").read().strip().splitlines()[-1]); print(n, "single fps", d["value"], "ms/frame", d["ms_per_step"], d["stream_matches_reference"])
except Exception as ex: print(n,"FAILED",ex)
P
done
cp /tmp/prod.so homerhevc_amd/libhomer_gpu.so
