#!/bin/bash
# GPU box: does background memory / ALU activity change the duration of the single-sequence CTU kernel?
export GPU_MAX_HW_QUEUES=16
WPP=17 python tools/multi_stream.py 1 1 2>&1 | cut -c1-150
for mode in 0 1; do
  tools/ubench/burner 64 $mode 60 &
  B=$!
  sleep 1
  echo "--- with burner mode $mode"
  WPP=17 python tools/multi_stream.py 1 1 2>&1 | cut -c1-150
  wait $B
done
