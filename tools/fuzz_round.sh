#!/bin/bash
# GPU box: a longer run of tools/encoder_fuzz.py on the device encoder (frame by frame, in batches, through the chain call) than the test suite's; prints the differing cases.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=${1:-5100}
( python3 tools/encoder_fuzz.py --gpu --decode --cases 450 --seed $S --max-ctus 150 > gpurun_out/fuzz_single.log 2>&1
  python3 tools/encoder_fuzz.py --gpu --batch 8 --cases 320 --seed $((S+1)) --max-ctus 150 > gpurun_out/fuzz_batch.log 2>&1
  python3 tools/encoder_fuzz.py --gpu --engines-only --chain-sets 2 --cases 120 --seed $((S+2)) --max-ctus 150 --max-cols 16 > gpurun_out/fuzz_chain.log 2>&1
  python3 tools/encoder_fuzz.py --gpu --threads-only --extra-keys --cases 200 --seed $((S+3)) --max-ctus 200 --max-cols 20 --max-rows 12 > gpurun_out/fuzz_threads.log 2>&1 )
grep "decoder-side check" gpurun_out/fuzz_single.log
for f in single batch chain threads; do echo "$f: $(grep -c IDENTICAL gpurun_out/fuzz_$f.log) identical, $(grep -c REFUSED gpurun_out/fuzz_$f.log) refused, $(grep -c DIFFERENT gpurun_out/fuzz_$f.log) different"; grep DIFFERENT gpurun_out/fuzz_$f.log | head -5; done
