cd $GRAFT_REPO_ROOT
cp homerhevc_amd/libhomer_gpu.so /tmp/prod.so; cp build/variants/${1:-fu}/libhomer_gpu.so homerhevc_amd/libhomer_gpu.so
for A in 1 0; do
if [ $A = 0 ]; then export HENC_NO_XCD_AFFINITY=1; else unset HENC_NO_XCD_AFFINITY; fi
python3 bench.py --sequences 256 --steps 5 --warmup 3 --no-cpu-baseline --no-single-thread-order 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('affinity $A cfg2 fps', d['value'], 'kernel ms', d['roofline']['ms_per_launch'], d['stream_matches_reference'], 'single', d.get('single_sequence',{}).get('value'))"
done
cp /tmp/prod.so homerhevc_amd/libhomer_gpu.so
