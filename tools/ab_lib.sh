#!/bin/bash
# Same-box A/B of bench.py between the in-tree libadk.so and another build (ADK_LIB override in _lib.py), two rounds.
# usage: bash tools/ab_lib.sh /abs/or/relative/path/to/other/libadk.so     (prints audio-s/s, ms per call, conv TF/s)
OTHER=$(readlink -f "${1:?path of the other libadk.so}")
show='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_denoiser_call"], d["roofline"]["achieved"])'
for i in 1 2; do
  echo "in-tree:"; timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-calls 0 | python -c "$show" || exit 1
  echo "other:"; ADK_LIB=$OTHER timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-calls 0 | python -c "$show" || exit 1
done
