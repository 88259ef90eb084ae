#!/bin/bash
# GPU box: host-trap PC sampling of the encoder (library built with HENC_EXTRA_FLAGS=-gline-tables-only), aggregated per source line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pcs
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${INTERVAL:-100} --output-format csv -d $OUT -o pcs -- python3 $R/tools/enc_run.py --width ${WIDTH:-1920} --height ${HEIGHT:-1080} --frames ${FRAMES:-4} wpp=${WPP:-17} > $OUT/run.log 2>&1
echo "rc=$?"; tail -3 $OUT/run.log
ls -la $OUT | head
f=$(ls $OUT/*pc_sampling*csv 2>/dev/null | head -1)
[ -n "$f" ] && head -3 $f && python3 $R/tools/pc_aggregate.py $f > $R/gpurun_out/pcs_summary.json && rm -f $f
