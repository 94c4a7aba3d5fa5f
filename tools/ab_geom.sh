#!/bin/bash
# A/B of development builds of the library on a geometry bench line:
#   bash tools/ab_geom.sh <workload> main build_dev/libX.so ...
set -e
wl=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    if [ "$lib" = main ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$PWD/$lib; fi
    timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --steps 20 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl $lib', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'of fill', r.get('frac_of_fill'))"
  done
done
