#!/bin/bash
# A/B of development builds of the library on the bfloat16 top-1 bench line:
#   bash tools/ab_bf16.sh build_dev/libA.so build_dev/libB.so ...   ("main" = the in-tree library)
set -e
for lib in "$@"; do
  for rep in 1 2; do
    if [ "$lib" = main ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$PWD/$lib; fi
    timeout -k 10 120 python bench.py --dtype bf16 --no-cpu-baseline --steps 30 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'])"
  done
done
