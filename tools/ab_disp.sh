#!/bin/bash
# Per-kernel times of the make_disparity bench line under rocprofv3 for the in-tree library ("base") and for
# development builds (tools/build_dev.sh): bash tools/ab_disp.sh base <tag> ...   (run on the GPU box, repo root)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for a in "$@"; do
  if [ $a = base ]; then unset ISPA_HIP_LIBRARY; else export ISPA_HIP_LIBRARY=$R/build_dev/lib$a.so; fi
  rm -rf $R/gpurun_out/abl_$a; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$a -- python3 $R/bench.py --workload disparity --no-cpu-baseline --no-extra > $R/gpurun_out/abl_$a.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/abl_$a -name "*kernel_stats.csv" | head -1)
  echo "== $a: $(grep -a "disparity_points\|disparity_maps" $f | awk -F, '{print "points " $(NF-4)}') $(grep -a disparity_bands $f | awk -F, '{print "bands " $(NF-4)}') $(grep -a fillBufferAligned $f | awk -F, '{print "fill " $(NF-4)}')"
done
