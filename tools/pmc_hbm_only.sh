set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o f -- python3 "$ROOT/bench.py" $ARGS --mode eager --launch-order "$OUT/order.json" > "$OUT/bench_pmc_f.json" 2> "$OUT/f.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o w -- python3 "$ROOT/bench.py" $ARGS --mode eager > "$OUT/bench_pmc_w.json" 2> "$OUT/w.err"
cd $ROOT
F=$(find "$OUT" -name 'f_counter_collection.csv' | head -1); W=$(find "$OUT" -name 'w_counter_collection.csv' | head -1)
python3 tools/pmc_summary.py "$F" "$W" "$OUT/order.json" > "$OUT/hbm_traffic.json" 2> "$OUT/pmc_summary.err"
find "$OUT" -name '*_counter_collection.csv' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
cat $OUT/pmc_summary.err; head -c 400 $OUT/hbm_traffic.json
