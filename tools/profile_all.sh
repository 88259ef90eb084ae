#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box: kernel stats (eager + graph), HBM counter passes, SQ counter passes, a frame timeline,
# then the bench records.  usage: bash tools/profile_all.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash tools/profile_round.sh $TAG > gpurun_out/profile_round.log 2>&1
bash tools/pmc_sq.sh $TAG > gpurun_out/pmc_sq.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tl -o tl -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/tl.err)
F=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
L=$(python3 -c "import json;print(len(json.load(open('gpurun_out/prof_$TAG/order.json'))))")
python3 tools/timeline.py $F $L > gpurun_out/timeline.md; rm -rf gpurun_out/tl
P=gpurun_out/prof_$TAG
mkdir -p gpurun_out/profiles_$TAG
O=gpurun_out/profiles_$TAG
cp $P/kernel_stats.csv $O/${TAG}_kernel_stats.csv; cp $P/kernel_stats_graph.csv $O/${TAG}_kernel_stats_graph.csv
cp $P/pmc_FETCH_SIZE.csv $O/${TAG}_pmc_FETCH_SIZE.csv; cp $P/pmc_WRITE_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv
cp $P/hbm_traffic.json $O/hbm_traffic.json; cp $P/order.json $O/${TAG}_launch_order.json
cp $P/bench_under_rocprof.json $O/${TAG}_bench_under_rocprof.json; cp $P/bench_under_rocprof_graph.json $O/${TAG}_bench_under_rocprof_graph.json
cp gpurun_out/pmc_$TAG/sq_summary.csv $O/sq_summary.csv; cp gpurun_out/pmc_$TAG/sq_summary.csv $O/${TAG}_sq_summary.csv
cp gpurun_out/timeline.md $O/${TAG}_timeline.md
# bench records with the fresh counter files in place
cp $O/hbm_traffic.json $O/sq_summary.csv profiles/
python3 bench.py > $O/${TAG}_bench.json 2> gpurun_out/bench_default.err
python3 bench.py --workload cfg4-2160p-P-frame-replay > $O/${TAG}_bench_2160p.json 2> /dev/null
python3 bench.py --workload cfg5-2160p-all-intra-replay --callmix-frame 1 > $O/${TAG}_bench_2160p_all_intra.json 2> /dev/null
python3 bench.py --cu-driver --no-cpu-baseline > $O/${TAG}_bench_cu_driver.json 2> /dev/null
python3 tools/kernel_table.py $O/${TAG}_bench.json $O/hbm_traffic.json $O/sq_summary.csv > $O/${TAG}_kernel_table.md
ls -la $O; tail -2 $P/pmc_summary.err
for f in bench bench_2160p bench_2160p_all_intra bench_cu_driver; do python3 -c "
import json
d=json.loads(open('$O/${TAG}_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['config']['launches_per_frame'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'), (d['roofline'].get('valu_issue') or {}).get('frac_of_measured_issue_rate'))
"; done
