#!/bin/bash
# GPU box: rocprofv3 kernel-trace statistics and PMC passes of the bench command (short run); summaries land in gpurun_out/ for copying into profiles/.
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters_sq.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
SEQ=${SEQ:-180}
ARGS="--steps 3 --warmup 3 --streams 0 --no-cpu-baseline --no-single-thread-order --sequences $SEQ"
FRAMES=$((6 * SEQ + 6))      # six steps of the batch + six frames of the single sequence measured beside it
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.log
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VSKIPPED" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${TAG}_$tag -o pmc --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/pmc_${TAG}_$tag.log
done
python3 $R/tools/pmc_kernels.py $R/gpurun_out $TAG "python3 bench.py $ARGS" $FRAMES > $R/gpurun_out/${TAG}_pmc_kernels.json
python3 $R/tools/kernel_launches.py $R/gpurun_out/prof_$TAG/bench_kernel_trace.csv $R/gpurun_out/${TAG}_bench_under_rocprof.json > $R/gpurun_out/${TAG}_k_encode_pool_launches.json
# the raw per-dispatch tables of a 120-sequence run exceed what comes back from the GPU box: the summaries above are what is kept
find $R/gpurun_out/pmc_${TAG}_* -name "*counter_collection.csv" -delete
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_$TAG -name "*stats*" | head
# request widths of the memory-side counters on known byte counts (tools/ubench/tcc_probe.hip): what makes `traffic` an absolute figure
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $R/gpurun_out/tccprobe_$tag -o p --output-format csv -- $R/tools/ubench/tcc_probe > $R/gpurun_out/tcc_probe.txt 2> $R/gpurun_out/tccprobe_$tag.log
done
python3 $R/tools/tcc_calibrate.py $R/gpurun_out/tcc_probe.txt $R/gpurun_out/tccprobe_*/p_counter_collection.csv > $R/gpurun_out/${TAG}_tcc_calibration.json
