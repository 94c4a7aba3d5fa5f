#!/bin/bash
# Per-kernel A/B of development builds under the tracer:
#   bash tools/ab_kernel.sh <kernel-name-substring> "<bench args>" main build_dev/libX.so ...
# prints the traced average duration of the named kernel for every library.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
kern=$1; args=$2; shift 2
for lib in "$@"; do
  if [ "$lib" = main ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$PWD/$lib; fi
  tag=$(basename "$lib" .so)
  out=gpurun_out/abk_$tag
  rm -rf "$out"
  timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- \
      python3 bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-extra > "$out.log" 2>&1
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$kern" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']:
        print(sys.argv[3], r['Name'].split('(')[0][-40:], r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
done
