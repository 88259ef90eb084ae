#!/bin/bash
# Collect the round's profiles on the GPU box: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE PMC passes (separate runs).
# usage (from the repo root on the GPU box): bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
# kernel durations: the eager replay (one launch at a time - these are the durations bench.py's roofline uses) and the default graph replay
# (launches of different graph branches overlap, so individual durations stretch)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python3 "$ROOT/bench.py" $ARGS --mode eager > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ktg -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_under_rocprof_graph.json" 2> "$OUT/ktg.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o f -- python3 "$ROOT/bench.py" $ARGS --mode eager --launch-order "$OUT/order.json" > "$OUT/bench_pmc_f.json" 2> "$OUT/f.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o w -- python3 "$ROOT/bench.py" $ARGS --mode eager > "$OUT/bench_pmc_w.json" 2> "$OUT/w.err"
cd "$ROOT"
cp "$(find "$OUT" -name 'kt_kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
cp "$(find "$OUT" -name 'ktg_kernel_stats.csv' | head -1)" "$OUT/kernel_stats_graph.csv"
F=$(find "$OUT" -name 'f_counter_collection.csv' | head -1); W=$(find "$OUT" -name 'w_counter_collection.csv' | head -1)
python3 tools/pmc_summary.py "$F" "$W" "$OUT/order.json" > "$OUT/hbm_traffic.json" 2> "$OUT/pmc_summary.err"
# keep only the small summaries (the raw traces exceed the merge limit)
python3 - "$F" "$W" "$OUT" <<'PY'
import csv, sys, collections
for path, name in ((sys.argv[1], "pmc_FETCH_SIZE.csv"), (sys.argv[2], "pmc_WRITE_SIZE.csv")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    with open(sys.argv[3] + "/" + name, "w") as f:
        f.write("Kernel_Name,Dispatches,Counter_Sum_KiB,Counter_Avg_KiB\n")
        for k, (n, s) in agg.items():
            f.write('"%s",%d,%.1f,%.2f\n' % (k, n, s, s / n))
PY
find "$OUT" -name '*_counter_collection.csv' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
ls -la "$OUT" | head -30
