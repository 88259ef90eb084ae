#!/bin/bash
# GPU box: run the short bench with each variant library of build/variants/ (tools/build_variant.sh) in place of the product library.
#   tools/variant_bench.sh "w2 w3 w4" [bench args...]   ->  gpurun_out/variants/NAME.json (+ .log)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
NAMES=$1; shift
ARGS=${@:---sequences 128 --steps 5 --warmup 3 --no-cpu-baseline --no-single-thread-order}
mkdir -p gpurun_out/variants
cp homerhevc_amd/libhomer_gpu.so /tmp/libhomer_gpu.product.so
for n in $NAMES; do
  cp build/variants/$n/libhomer_gpu.so homerhevc_amd/libhomer_gpu.so
  timeout 600 python3 bench.py $ARGS > gpurun_out/variants/$n.json 2> gpurun_out/variants/$n.log
  python3 - $n <<'P'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/variants/{n}.json").read().strip().splitlines()[-1])
    ss = d.get("single_sequence", {})
    print(n, "fps", d["value"], "ms/step", d["ms_per_step"], "match", d["stream_matches_reference"], "kernel ms", d["roofline"]["ms_per_launch"], "single", ss.get("value"), ss.get("stream_matches_reference"))
except Exception as ex:
    print(n, "FAILED", ex)
P
done
cp /tmp/libhomer_gpu.product.so homerhevc_amd/libhomer_gpu.so
