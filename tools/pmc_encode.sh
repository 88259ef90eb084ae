cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_$tag -o pmc --output-format csv -- python3 $R/tools/enc_run.py --width 416 --height 240 --frames 3 > $R/gpurun_out/pmc_$tag.log 2>&1
done
ls -R $R/gpurun_out | head -30
