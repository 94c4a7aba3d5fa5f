#!/bin/bash
# A/B of development builds on the float32 filter beyond top-1 / 128 channels:
#   bash tools/ab_f32k.sh main build_dev/libX.so ...
set -e
for lib in "$@"; do
  if [ "$lib" = main ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$PWD/$lib; fi
  for cfg in "24 128 256 5" "128 64 128 5" "24 128 256 1" "128 64 64 5" "128 64 128 8"; do
    set -- $cfg
    timeout -k 10 120 python tools/ab_check.py $1 $2 $3 $4 f32 | cut -c1-60
    timeout -k 10 120 python bench.py --dtype f32 --pairs $1 --grid $2 --channels $3 --k $4 --no-cpu-baseline --no-extra --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', '$cfg', 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'TF', r['achieved'], 'frac', r['frac'])"
  done
done
