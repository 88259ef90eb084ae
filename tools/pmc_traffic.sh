#!/bin/bash
# GPU box: the two memory-side counter passes (TCC_EA0 read / write requests by width) of a short default-size batch run on the build in the tree;
# summed per kernel by tools/pmc_kernels.py.  usage: tools/pmc_traffic.sh [tag]     (each pass is bounded by `timeout`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05_final_traffic}
ARGS="--steps 2 --warmup 2 --no-cpu-baseline --no-single-thread-order --sequences 256"
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 380 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${TAG}_$tag -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_$tag.json 2> $R/gpurun_out/pmc_${TAG}_$tag.log
  echo "$tag rc=$?"
done
python3 $R/tools/pmc_kernels.py $R/gpurun_out $TAG "python3 bench.py $ARGS" $((4 * 256 + 4)) > $R/gpurun_out/${TAG}_pmc_kernels.json      # (frames: four steps of the batch + the four frames of the single sequence beside it)
find $R/gpurun_out/pmc_${TAG}_* -name "*counter_collection.csv" -delete
