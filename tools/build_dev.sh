#!/bin/bash
# Development build of the library with extra compiler flags for ONE source file:
#   bash tools/build_dev.sh <tag> <source stem, e.g. nn> <flags...>   -> build_dev/lib<tag>.so
# (the other objects come from the in-tree build; select the library with ISPA_HIP_LIBRARY=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; stem=$2; shift 2
mkdir -p $R/build_dev
C=$R/pose_estimation_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
    -I$R/include -I$C "$@" -c $C/$stem.hip -o $R/build_dev/${stem}_$tag.o
objs=""
for o in $C/build/*.o; do
  [ "$(basename $o)" = "$stem.o" ] && continue
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build_dev/${stem}_$tag.o -o $R/build_dev/lib$tag.so
echo built build_dev/lib$tag.so
