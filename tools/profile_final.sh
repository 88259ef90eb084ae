#!/bin/bash
# GPU box: rocprofv3 kernel-trace statistics of the default bench command (256 sequences, 20 timed steps; the side measurements left out) on the build in the tree:
# kernel_stats.csv, the pool kernel's launches against the bench's own HIP-event figure, the bench line of that run.  usage: tools/profile_final.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04_final}
ARGS="--no-cpu-baseline --no-single-thread-order"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.log
python3 $R/tools/kernel_launches.py $(find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1) $R/gpurun_out/${TAG}_bench_under_rocprof.json > $R/gpurun_out/${TAG}_k_encode_pool_launches.json
cp $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
head -3 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
