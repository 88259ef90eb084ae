#!/bin/bash
# GPU box experiment: do workers that share a CU slow each other, or is it the number of workers on the chip?  The same 256 (or 512) pool workers spread over all CUs,
# or packed three (two) to a CU on every third (second) CU through a stream with a CU mask (HENC_CU_MASK_EVERY, context.cpp).    tools/cu_sharing.sh [variant]
cd $GRAFT_REPO_ROOT
V=${1:-q3m}
cp homerhevc_amd/libhomer_gpu.so /tmp/libhomer_gpu.product.so
cp build/variants/$V/libhomer_gpu.so homerhevc_amd/libhomer_gpu.so
run() {
  python3 bench.py --sequences 128 --steps 4 --warmup 3 --no-cpu-baseline --no-single-thread-order 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'fps', d['value'], 'kernel ms', d['roofline']['ms_per_launch'], d['stream_matches_reference'])"
}
HENC_POOL_WORKERS=256 run "256 workers, all CUs (1 per CU)"
HENC_POOL_WORKERS=256 HENC_CU_MASK_EVERY=2 run "256 workers, every 2nd CU (2 per CU)"
HENC_POOL_WORKERS=256 HENC_CU_MASK_EVERY=3 run "256 workers, every 3rd CU (3 per CU)"
HENC_POOL_WORKERS=512 run "512 workers, all CUs (2 per CU)"
HENC_POOL_WORKERS=384 HENC_CU_MASK_EVERY=2 run "384 workers, every 2nd CU (3 per CU)"
HENC_POOL_WORKERS=384 run "384 workers, all CUs (1.5 per CU)"
run "768 workers, all CUs (3 per CU)"
cp /tmp/libhomer_gpu.product.so homerhevc_amd/libhomer_gpu.so
