#!/bin/bash
# A/B of development builds on the float32 headline bench line:  bash tools/ab_f32.sh main build_dev/libX.so ...
set -e
for lib in "$@"; do
  for rep in 1 2; do
    if [ "$lib" = main ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$PWD/$lib; fi
    timeout -k 10 120 python bench.py --dtype f32 --no-cpu-baseline --no-extra --steps 10 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'])"
  done
done
