cd $GRAFT_REPO_ROOT
cp homerhevc_amd/libhomer_gpu.so /tmp/prod.so; cp build/variants/${1:-fu}/libhomer_gpu.so homerhevc_amd/libhomer_gpu.so
python3 bench.py --sequences 256 --steps 5 --warmup 3 --no-cpu-baseline --no-single-thread-order 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 fps', d['value'], 'kernel ms', d['roofline']['ms_per_launch'], d['stream_matches_reference'], 'single', d.get('single_sequence',{}).get('value'))"
python3 bench.py --workload cfg3-2160p-cbr --sequences 32 --steps 6 --warmup 2 --no-cpu-baseline --no-single-thread-order 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cbr fps', d['value'], 'kernel ms', d['roofline']['ms_per_launch'], d['stream_matches_reference'], 'single', d.get('single_sequence',{}).get('value'))"
cp /tmp/prod.so homerhevc_amd/libhomer_gpu.so
