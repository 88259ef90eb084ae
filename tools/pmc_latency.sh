#!/bin/bash
# GPU box: how long the pool kernel's memory instructions, LDS instructions and instruction fetches stay in flight (SQ_*_LEVEL counters: the sum over
# cycles of the operations outstanding; divided by the operations issued = mean time in flight in the counter's unit), one rocprofv3 --pmc pass per set.
# Three passes of the 180-sequence bench take about 20 minutes on the box (counter collection serialises the launches).  A TCP_TCC_*_REQ_LATENCY pass aborted
# inside rocprofv3 on this pool and is not part of the list.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04lat}
SEQ=${SEQ:-180}
ARGS="--steps 2 --warmup 2 --streams 0 --no-cpu-baseline --no-single-thread-order --sequences $SEQ"
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TCC\|TA\|TD\|SQ\|SQC\|GRBM\|SPI\)_[A-Za-z_0-9]*" | sort -u > $R/gpurun_out/counters_all.txt
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_FLAT_LDS_ONLY" "SQ_IFETCH_LEVEL SQ_IFETCH SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_MISSES SQC_ICACHE_REQ"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${TAG}_$tag -o pmc --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/pmc_${TAG}_$tag.log
done
python3 $R/tools/pmc_kernels.py $R/gpurun_out $TAG "python3 bench.py $ARGS" > $R/gpurun_out/${TAG}_pmc_kernels.json
find $R/gpurun_out/pmc_${TAG}_* -name "*counter_collection.csv" -delete
