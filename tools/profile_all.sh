#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ on the GPU box (run from the repo root):
#   bash tools/profile_all.sh            (ONLY="zbuffer corr" bash tools/profile_all.sh  re-runs just those tags)
# then, back in the build container:  python tools/profile_collect.py gpurun_out profiles r03
# One pass per counter set; the PMC passes never combine with API tracing.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
skip() { [ -n "$ONLY" ] && ! [[ " $ONLY " == *" $1 "* ]]; }
run() {   # tag, bench args...
    local tag=$1; shift
    skip $tag && return 0
    echo "== $tag: $*"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -- \
        python3 $R/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_$tag.json 2> $OUT/stats_$tag.err || return 1
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${tag}_$c -- \
            python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_${tag}_$c.err || return 1
    done
}
sq() {    # tag, bench args...
    local tag=$1; shift
    skip $tag && return 0
    timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_${tag}_SQ -- \
        python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_${tag}_SQ.err || return 1
}
sq2() {   # tag, bench args...: where the issue slots and the waiting go (second SQ counter set)
    local tag=$1; shift
    skip $tag && return 0
    timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD \
        SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_${tag}_SQ2 -- \
        python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_${tag}_SQ2.err || return 1
}
run nn___dtype_f32 --workload nn --dtype f32 --no-extra &&
run nn___dtype_bf16 --workload nn --dtype bf16 &&
run nn___dtype_f32___filter_split --workload nn --dtype f32 --filter split --no-extra &&
run project --workload project &&
run project_depth_per_pose --workload project --depth-per-pose &&
run disparity --workload disparity &&
run zbuffer --workload zbuffer &&
run zbuffer_unfused --workload zbuffer --unfused --poses 720 &&
run corr --workload corr &&
run nn___dtype_f32___matches --workload nn --dtype f32 --matches --no-extra &&
run nn___config5_bf16_k5 --workload nn --dtype bf16 --grid 128 --channels 256 --k 5 --pairs 24 --no-extra &&
sq nn___dtype_f32 --workload nn --dtype f32 &&
sq nn___dtype_bf16 --workload nn --dtype bf16 &&
sq nn___dtype_f32___filter_split --workload nn --dtype f32 --filter split &&
sq zbuffer --workload zbuffer &&
sq disparity --workload disparity &&
sq2 disparity --workload disparity &&
sq2 nn___dtype_bf16 --workload nn --dtype bf16 &&
sq2 nn___dtype_f32 --workload nn --dtype f32 &&
cd $R && { [ -n "$ONLY" ] || timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; } &&
echo "profile_all: done"
