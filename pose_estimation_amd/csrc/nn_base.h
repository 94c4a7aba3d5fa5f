// Dense nearest neighbour, shared by every pass: tile constants, development switches, the XCD-aware
// workgroup order, element loads and the float64 squared distance all float64 passes use (included by nn.hip).
#pragma once

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef ISPA_TOP1_32X32
#define ISPA_TOP1_32X32 0   // development switch: bfloat16 top-1 through the 32x32x16 kernel instead of the 16x16x32 one
#endif
#ifndef ISPA_ABLATE
#define ISPA_ABLATE 0   // development switch for timing-only builds (results are wrong when non-zero):
#endif                  // bf16 filter 1 = no top-list update, 2 = no chunk streaming; f32 filter 4, 5, 6 = both

constexpr int kTQ = 128;       // queries per workgroup tile
constexpr int kTR = 128;       // references per chunk
constexpr int kKC = 16;        // channels per LDS stage
constexpr int kGroups = kKC / 8; // float4 groups per thread per staged tile
constexpr int kThreads = 256;  // 4 waves: 2 (reference halves) x 2 (query halves)
constexpr int kSubStreams = 4; // partial result streams per (tile, split): wave-row x lane-half
constexpr int kMaxK = 8;
// Score of a masked (out-of-range) reference and of an empty list slot: a huge finite negative, so
// that it survives the mantissa tagging below; anything under kAbsent counts as "no entry".
constexpr float kNegInf = -3.0e38f;
constexpr float kAbsent = -1.0e38f;
// A distance entry that nn_finalize_kernel leaves for nn_refine_kernel: a NaN bit pattern no arithmetic produces.
constexpr uint32_t kRefineMark = 0x7fc5a5a5u;

// Blocks b and b+8 share an XCD (and its L2).  Renumber so that each XCD walks a contiguous range
// of logical ids: the query tiles of one image pair, which all stream the same reference map, then
// run on one L2.  Bijective for any grid size; a pure speed choice.
__device__ __forceinline__ int xcd_order(int bid, int nblocks) {
    const int q = nblocks >> 3, rem = nblocks & 7, x = bid & 7;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + (bid >> 3);
}

// What one workgroup of a filter kernel works on.  Grid = pairs x reference splits x query tiles, walked in the
// XCD-friendly order of xcd_order(): the query tiles of one (pair, split), which stream the same references,
// sit next to each other.
struct FilterWork {
    int split, b;            // reference split, image pair
    int64_t q0;              // first query column of the tile
    int64_t rbeg, rend;      // reference range of the split
    int nchunks;             // chunks of `chunk_refs` references in that range
};
__device__ __forceinline__ FilterWork filter_work(int n_qtiles, int nsplit, int64_t refs_per_split, int64_t nr,
                                                  int tile_queries, int chunk_refs) {
    int bid = xcd_order(blockIdx.x, gridDim.x);
    FilterWork w;
    w.q0 = (int64_t)(bid % n_qtiles) * tile_queries;
    bid /= n_qtiles;
    w.split = bid % nsplit;
    w.b = bid / nsplit;
    w.rbeg = (int64_t)w.split * refs_per_split;
    w.rend = min(w.rbeg + refs_per_split, nr);
    w.nchunks = (int)((w.rend - w.rbeg + chunk_refs - 1) / chunk_refs);
    return w;
}

// The partial list of one query stream goes to part_val / part_idx (pairs, streams, SLOTS, queries); finalize
// merges the streams.  `to_ref` turns a list code into a reference index.
template <int SLOTS, typename List, typename ToRef>
__device__ __forceinline__ void store_partial(const List &best, ToRef to_ref, float *__restrict__ part_val,
                                              int *__restrict__ part_idx, int b, int nparts, int part, int64_t nq,
                                              int64_t qcol) {
    if (qcol >= nq) return;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int code = best.code(s);
        const int64_t o = (((int64_t)b * nparts + part) * SLOTS + s) * nq + qcol;
        part_val[o] = best.value(s);
        part_idx[o] = code >= 0 ? to_ref(code) : -1;
    }
}

__device__ __forceinline__ float load_as_float(const float *p) { return *p; }
__device__ __forceinline__ float load_as_float(const uint16_t *p) {
    return __uint_as_float(((uint32_t)*p) << 16);
}

// Squared distance of one (query column, reference column) pair in the CPU oracle's arithmetic:
// (qq + rr) - 2 dot, dot accumulated over channels in increasing order in float64 (a product of two
// float32 values is exact in float64), qq and rr from the norms kernel, accumulated the same way.  Two
// identical columns give exactly 0.  32 channels of loads are in flight at a time (every one of them is a
// different 64-byte sector, a microsecond or more away when the maps do not fit into L2), then 16.
template <typename T>
__device__ __forceinline__ double exact_d2(const T *__restrict__ qcol, const T *__restrict__ rcol, int C, int64_t nq,
                                           int64_t nr, double qq, double rr) {
    double dot = 0.0;
    int c0 = 0;
    for (; c0 + 32 <= C; c0 += 32) {
        float qv[32], rv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            qv[u] = load_as_float(qcol + (int64_t)(c0 + u) * nq);
            rv[u] = load_as_float(rcol + (int64_t)(c0 + u) * nr);
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) dot = fma((double)qv[u], (double)rv[u], dot);
    }
    for (; c0 + 16 <= C; c0 += 16) {
        float qv[16], rv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            qv[u] = load_as_float(qcol + (int64_t)(c0 + u) * nq);
            rv[u] = load_as_float(rcol + (int64_t)(c0 + u) * nr);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) dot = fma((double)qv[u], (double)rv[u], dot);
    }
    for (; c0 < C; ++c0)
        dot = fma((double)load_as_float(qcol + (int64_t)c0 * nq), (double)load_as_float(rcol + (int64_t)c0 * nr), dot);
    return (qq + rr) - 2.0 * dot;
}

}  // namespace
