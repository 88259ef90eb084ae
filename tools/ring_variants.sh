#!/bin/bash
# GPU box: the engine ring of bench.py --gpus N as N processes on ONE MI355X (HOMER_BENCH_ONE_DEVICE=1: gloo, the pictures cross page-locked host buffers), repeated with
# different rank counts, warm-up lengths and call kinds; every access unit is checked against the reference's engine digests.  Prints one line per run.
# usage: bash tools/ring_variants.sh [repeats]        (HOMER_BENCH_DUMP_UNITS=dir / HOMER_RING_TRACE=dir: keep the access units / the checksums of what was sent and received)
export HOMER_BENCH_ONE_DEVICE=1 HENC_WATCHDOG_S=60
run() { N=$1; shift; timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --sequences 4 "$@" > gpurun_out/rv.json 2> gpurun_out/rv.err; python - "$N" "$@" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/rv.json").read().strip().splitlines()[-1])
    print(sys.argv[1:], d["value"], d["stream_matches_reference"], d["access_units_differing"], d["access_units_produced"], [(x["sequence"],x["frame"]) for x in d["first_differences_on_rank_0"]])
except Exception as ex:
    print(sys.argv[1:], "failed", ex, open("gpurun_out/rv.err").read()[-400:])
PY
}
mkdir -p gpurun_out
for k in $(seq 1 ${1:-2}); do
  run 2 --steps 6 --warmup 3
  run 3 --steps 6 --warmup 3
  run 4 --steps 6 --warmup 3
  run 4 --steps 6 --warmup 3 --no-pipeline
  run 4 --steps 9 --warmup 0
done
