cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bprof
export HOMER_GPU_LIB=build/variants/prof/libhomer_gpu.so
HENC_LDS_BYTES=100000 python3 tools/batch_profile.py --sequences 128 --frames 3 --out gpurun_out/bprof/w1.json > gpurun_out/bprof/w1.log 2>&1
HENC_LDS_BYTES=70000 python3 tools/batch_profile.py --sequences 128 --frames 3 --out gpurun_out/bprof/w2.json > gpurun_out/bprof/w2.log 2>&1
python3 tools/batch_profile.py --sequences 128 --frames 3 --out gpurun_out/bprof/w3.json > gpurun_out/bprof/w3.log 2>&1
tail -2 gpurun_out/bprof/w3.log | cut -c1-300
