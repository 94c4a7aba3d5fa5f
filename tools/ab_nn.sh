#!/bin/bash
# A/B of development builds of the library on any dense-NN bench line:
#   bash tools/ab_nn.sh "<bench args>" main build_dev/libX.so ...
set -e
args=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    if [ "$lib" = main ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$PWD/$lib; fi
    timeout -k 10 120 python bench.py $args --no-cpu-baseline --no-extra --steps 20 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'])"
  done
done
