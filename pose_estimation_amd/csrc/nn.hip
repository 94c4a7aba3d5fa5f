// Dense nearest-neighbour correspondence between two feature maps (squared L2, top-k).
//
// Contract: the `KNN` layer of the Caffe-UCN fork as the reference uses it
// (ucn_code/nets/train.prototxt:2116-2141; consumer hard_negative_processing_layer.py:175-192).
//
// Structure (one call = 5 or 6 launches on the caller's stream):
//   1. norms      nh[j] = -0.5 * |x_j|^2 for queries and references (float64 sums kept for step 4),
//                 max reference norm per pair; the split variant also writes the bf16 hi / lo images
//   2. filter     MFMA contraction  s[j][i] = r_j . q_i - 0.5 |r_j|^2  (the accumulator starts at
//                 nh[j]), reduced in registers to the (k+1) largest s per query and partial
//                 stream; the Nq x Nr score matrix never leaves the register file.  Variants:
//                 2a  float32 MFMA, tiles staged in LDS (any k, any C, converting loader for odd shapes)
//                 2a' float32 MFMA, query fragments resident in registers (top-1, C <= 128)
//                 2b  bfloat16 32x32x16 MFMA (k > 1, C <= 256)
//                 2c  bfloat16 16x16x32 MFMA, tagged top-1 (C <= 256)
//                 2d  float32 inputs as a bfloat16 hi/lo split on the 16x16x32 MFMA (top-1, C <= 128)
//   3. finalize   merges the partial streams; a query whose consecutive score gaps all exceed
//                 the rounding bound of the filter is final; inside the bound the references that
//                 can still matter are re-scored in float64 when they are all known by index,
//                 otherwise the query is queued
//   4. exact      float64 re-scoring of every reference for the queued queries, in the channel
//                 order the CPU oracle uses, so the index order is exact under all inputs
//                 (a batched top-1 variant shares one sweep over a pair's reference map)
#include <type_traits>

#include "common.h"

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

// ---------------------------------------------------------------------------------------------
// 1. norms
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nn_norms_kernel(const T *__restrict__ x, int C, int64_t n,
                                                       float *__restrict__ nh, double *__restrict__ sq64,
                                                       int *__restrict__ maxbits) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (j < n) {
        const T *col = x + (int64_t)b * C * n + j;
        int c = 0;
        for (; c + 8 <= C; c += 8) {           // 8 loads in flight, then the ordered accumulation
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_as_float(col + (int64_t)(c + u) * n);
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fma((double)v[u], (double)v[u], s);
        }
        for (; c < C; ++c) {                   // channel order of the exact pass and of the oracle
            const double v = (double)load_as_float(col + (int64_t)c * n);
            s = fma(v, v, s);
        }
        nh[(int64_t)b * n + j] = (float)(-0.5 * s);
        sq64[(int64_t)b * n + j] = s;
    }
    if (maxbits) {
        // non-negative floats order like their bit patterns; round up so the bound stays a bound
        float m = __double2float_ru(s);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_int(m));
    }
}

// Small problems (a single image pair per call): one cell per thread in 64-thread workgroups, so that a
// few thousand cells still spread over a hundred CUs, and 32 channels of loads in flight per thread (the
// four-cells-per-thread kernels above would run as 8 workgroups waiting for 4 loads at a time).  Query and
// reference map in one launch; workgroup (0, 0) also clears the queue counters of the later passes, which
// saves the separate memset launch.  Same sums in the same channel order as nn_norms_kernel.
template <typename T>
__global__ __launch_bounds__(64) void nn_norms_small_kernel(const T *__restrict__ q, const T *__restrict__ r, int C,
                                                            int64_t nq, int64_t nr, int nbq,
                                                            float *__restrict__ qnh, double *__restrict__ q64,
                                                            float *__restrict__ rnh, double *__restrict__ r64,
                                                            int *__restrict__ zero, int nzero) {
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int t = threadIdx.x; t < nzero; t += 64) zero[t] = 0;
    const int b = blockIdx.y;
    const bool is_q = (int)blockIdx.x < nbq;
    const int64_t n = is_q ? nq : nr;
    const int64_t j = (int64_t)(is_q ? blockIdx.x : blockIdx.x - nbq) * 64 + threadIdx.x;
    if (j >= n) return;
    const T *col = (is_q ? q : r) + (int64_t)b * C * n + j;
    double s = 0.0;
    int c = 0;
    for (; c + 32 <= C; c += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = load_as_float(col + (int64_t)(c + u) * n);
#pragma unroll
        for (int u = 0; u < 32; ++u) s = fma((double)v[u], (double)v[u], s);
    }
    for (; c < C; ++c) {
        const double v = (double)load_as_float(col + (int64_t)c * n);
        s = fma(v, v, s);
    }
    (is_q ? qnh : rnh)[(int64_t)b * n + j] = (float)(-0.5 * s);
    (is_q ? q64 : r64)[(int64_t)b * n + j] = s;
}

// ISPA_NN_NORMALIZE: L2NormLayer.forward (ucn_code/lib/layers/l2norm_layer.py:18-29) fused into the norms
// pass.  y = x / sqrt(sum_c x^2) with the arithmetic of ispa_l2norm (float32 fmaf chain over channels,
// sqrtf, division), written to a copy that the later passes read, and in the same sweep the float64 norms
// of y that nn_norms_kernel would compute from that copy: the call returns what ispa_l2norm on both maps
// followed by ispa_nn_topk returns, bit for bit, with one pass over the inputs less.
__global__ __launch_bounds__(256) void nn_l2norm_norms_kernel(const float *__restrict__ x, int C, int64_t n,
                                                              float *__restrict__ y, float *__restrict__ nh,
                                                              double *__restrict__ sq64, int *__restrict__ maxbits) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (j < n) {
        const float *col = x + (int64_t)b * C * n + j;
        float *dst = y + (int64_t)b * C * n + j;
        float s32 = 0.f;
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(c + u) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s32 = fmaf(v[u], v[u], s32);
        }
        for (; c < C; ++c) s32 = fmaf(col[(int64_t)c * n], col[(int64_t)c * n], s32);
        const float norm = sqrtf(s32);
        for (c = 0; c + 8 <= C; c += 8) {          // second sweep: the column is in cache
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(c + u) * n] / norm;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dst[(int64_t)(c + u) * n] = v[u];
                s = fma((double)v[u], (double)v[u], s);
            }
        }
        for (; c < C; ++c) {
            const float v = col[(int64_t)c * n] / norm;
            dst[(int64_t)c * n] = v;
            s = fma((double)v, (double)v, s);
        }
        nh[(int64_t)b * n + j] = (float)(-0.5 * s);
        sq64[(int64_t)b * n + j] = s;
    }
    if (maxbits) {
        float m = __double2float_ru(s);             // NaN (an all-zero column: 0 / 0) is dropped by fmaxf
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_int(m));
    }
}

// Four consecutive cells per thread (16-byte loads for float, 8-byte for bfloat16): same sums in the
// same channel order, wider memory transactions.  Needs n % 4 == 0 and 16-byte aligned rows.
template <typename T>
__device__ __forceinline__ void norms4_block(const T *__restrict__ x, int C, int64_t n, float *__restrict__ nh,
                                             double *__restrict__ sq64, int *__restrict__ maxbits, int b,
                                             int block) {
    const int64_t j = ((int64_t)block * 256 + threadIdx.x) * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < n) {
        const T *col = x + (int64_t)b * C * n + j;
        auto load4 = [&](int c, float (&v)[4]) {
            if (sizeof(T) == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(col + (int64_t)c * n);
                v[0] = t.x;
                v[1] = t.y;
                v[2] = t.z;
                v[3] = t.w;
            } else {
                const uint2 t = *reinterpret_cast<const uint2 *>(col + (int64_t)c * n);
                v[0] = __uint_as_float(t.x << 16);
                v[1] = __uint_as_float(t.x & 0xffff0000u);
                v[2] = __uint_as_float(t.y << 16);
                v[3] = __uint_as_float(t.y & 0xffff0000u);
            }
        };
        int c = 0;
        for (; c + 4 <= C; c += 4) {           // 4 wide loads in flight, then the ordered accumulation
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) load4(c + u, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = fma((double)v[u][e], (double)v[u][e], s[e]);
        }
        for (; c < C; ++c) {
            float v[4];
            load4(c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = fma((double)v[e], (double)v[e], s[e]);
        }
        float4 out;
        out.x = (float)(-0.5 * s[0]);
        out.y = (float)(-0.5 * s[1]);
        out.z = (float)(-0.5 * s[2]);
        out.w = (float)(-0.5 * s[3]);
        *reinterpret_cast<float4 *>(nh + (int64_t)b * n + j) = out;
        double2 *o64 = reinterpret_cast<double2 *>(sq64 + (int64_t)b * n + j);
        o64[0] = make_double2(s[0], s[1]);
        o64[1] = make_double2(s[2], s[3]);
    }
    if (maxbits) {
        float m = __double2float_ru(fmax(fmax(s[0], s[1]), fmax(s[2], s[3])));
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_int(m));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nn_norms4_kernel(const T *__restrict__ x, int C, int64_t n,
                                                        float *__restrict__ nh, double *__restrict__ sq64,
                                                        int *__restrict__ maxbits) {
    norms4_block(x, C, n, nh, sq64, maxbits, blockIdx.y, blockIdx.x);
}

// query map and reference map of every pair in one launch: blocks [0, nbq) take the queries
template <typename T>
__global__ __launch_bounds__(256) void nn_norms4_both_kernel(const T *__restrict__ q, const T *__restrict__ r, int C,
                                                             int64_t nq, int64_t nr, int nbq,
                                                             float *__restrict__ qnh, double *__restrict__ q64,
                                                             float *__restrict__ rnh, double *__restrict__ r64,
                                                             int *__restrict__ rmax) {
    if ((int)blockIdx.x < nbq) norms4_block(q, C, nq, qnh, q64, (int *)nullptr, blockIdx.y, blockIdx.x);
    else norms4_block(r, C, nr, rnh, r64, rmax, blockIdx.y, blockIdx.x - nbq);
}

// ---------------------------------------------------------------------------------------------
// 2. filter pass, float32 MFMA (v_mfma_f32_32x32x2_f32: exact float32 fma chain over channels)
// ---------------------------------------------------------------------------------------------
// Running top list of one query stream.  Interface shared by both implementations:
//   begin_half(id)  start of a half-step (32 reference rows = 16 accumulator registers per lane)
//   push(x, reg)    score of accumulator register `reg` (compile-time after unrolling)
//   end_half()
//   value(s), code(s)  s-th best score and its code = half_step_id * 16 + reg (-1: none)
template <int K>
struct TopList {            // K+1 best scores, descending, each with its code
    float v[K + 1];
    int c[K + 1];
    int base;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            v[s] = kNegInf;
            c[s] = -1;
        }
        base = 0;
    }
    __device__ __forceinline__ void begin_half(int id) { base = id * 16; }
    __device__ __forceinline__ void end_half() {}
    template <typename Acc>
    __device__ __forceinline__ void end_half(const Acc &) {}
    __device__ __forceinline__ void push(float x, int reg) {
        const int code = base + reg;
        if (K == 1) {
            v[1] = __builtin_amdgcn_fmed3f(v[0], v[1], x);
            const bool better = x > v[0];
            c[0] = better ? code : c[0];
            v[0] = better ? x : v[0];
        } else {
            // Most scores beat nobody's list: skip the update when no lane of the wave has a
            // candidate (the chance falls like 64 (K+1) / t with the position t in the stream).
            if (__builtin_amdgcn_ballot_w64(x > v[K]) == 0) return;
            // sorted insert, branch-free: slot s takes its upper neighbour if x goes above that one,
            // x if x goes exactly here, else keeps its entry (x after equal scores: lower code first).
            // Scores: one v_med3 per slot; codes: two selects per slot on the K+1 comparisons.
            bool g[K + 1];
#pragma unroll
            for (int s = 0; s <= K; ++s) g[s] = x > v[s];
            const float xs = g[K] ? x : kNegInf;         // a score that stays out (or a NaN) changes nothing
#pragma unroll
            for (int s = K; s > 0; --s) {
                c[s] = g[s - 1] ? c[s - 1] : (g[s] ? code : c[s]);
                v[s] = __builtin_amdgcn_fmed3f(xs, v[s - 1], v[s]);
            }
            c[0] = g[0] ? code : c[0];
            v[0] = fmaxf(xs, v[0]);
        }
    }
    __device__ __forceinline__ void flush() {}
    __device__ __forceinline__ float value(int s) const { return v[s]; }
    __device__ __forceinline__ int code(int s) const { return v[s] > kAbsent ? c[s] : -1; }
};

// Top-1 (plus the runner-up's score) at three vector instructions per score: the accumulator
// register number replaces the 4 low mantissa bits of the score (a perturbation below 2^-19
// relative, accounted for in the margin test), so the best score carries its own position and
// v_max / v_med3 do all the work; the half-step that produced it is tracked once per 16 scores.
struct TagTop1 {
    float a, b, a_before, pinf;
    int half, cur;
    __device__ __forceinline__ void init() {
        a = b = a_before = kNegInf;
        half = -1;
        cur = -1;
        // +inf in a register the optimiser cannot see through: med3(a, t, +inf) then stays ONE
        // v_med3_f32 instead of a canonicalising v_max pair
        pinf = __builtin_inff();
        asm volatile("" : "+v"(pinf));
    }
    __device__ __forceinline__ void begin_half(int id) {
        cur = id;
        a_before = a;
    }
    __device__ __forceinline__ void push(float x, int reg) {
        const float t = __uint_as_float((__float_as_uint(x) & 0xfffffff0u) | (unsigned)reg);
        b = __builtin_amdgcn_fmed3f(a, b, t);
        // max(a, t) as ONE two-operand instruction (the compiler would add a canonicalising
        // v_max in front of fmaxf because t comes out of integer arithmetic)
        asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(t));
    }
    // Two scores at once in five instructions: with m = med3(a, t1, t2) the new best is
    // max3(a, t1, t2) and the new runner-up max(b, m) (b <= a, so whichever of a, t1, t2 is the
    // middle one is the second largest of the three, and only b can still beat it).  Split in two
    // halves so that a caller can place them between different MFMAs.
    float t1, t2, mid;
    __device__ __forceinline__ void push2_begin(float x1, int reg1, float x2, int reg2) {
        t1 = __uint_as_float((__float_as_uint(x1) & 0xfffffff0u) | (unsigned)reg1);
        t2 = __uint_as_float((__float_as_uint(x2) & 0xfffffff0u) | (unsigned)reg2);
        asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid) : "v"(a), "v"(t1), "v"(t2));
    }
    __device__ __forceinline__ void push2_end() {
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(t1), "v"(t2));
        asm("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(mid));
    }
    // Four scores in nine instructions, in four slices a caller can place between different MFMAs:
    // with m1 = med3(a, t1, t2), a' = max3(a, t1, t2), m2 = med3(a', t3, t4), a'' = max3(a', t3, t4) the
    // new runner-up is max3(b, m1, m2).
    float mid2;
    __device__ __forceinline__ float tagged(float x, int tag) const {
        return __uint_as_float((__float_as_uint(x) & 0xfffffff0u) | (unsigned)tag);
    }
    template <typename V>
    __device__ __forceinline__ void push4_step(int step, const V &x, int tag0, int off = 0) {
        if (step == 0) {
            t1 = tagged(x[off], tag0);
            t2 = tagged(x[off + 1], tag0 + 1);
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid) : "v"(a), "v"(t1), "v"(t2));
        } else if (step == 1) {
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(t1), "v"(t2));
            t1 = tagged(x[off + 2], tag0 + 2);
        } else if (step == 2) {
            t2 = tagged(x[off + 3], tag0 + 3);
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid2) : "v"(a), "v"(t1), "v"(t2));
        } else {
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(t1), "v"(t2));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(mid), "v"(mid2));
        }
    }
    // x[off .. off+3] with tags tag0 .. tag0+3
    template <typename V>
    __device__ __forceinline__ void push4(const V &x, int tag0, int off = 0) {
#pragma unroll
        for (int step = 0; step < 4; ++step) push4_step(step, x, tag0, off);
    }
    __device__ __forceinline__ void end_half() { half = a != a_before ? cur : half; }
    template <typename Acc>
    __device__ __forceinline__ void end_half(const Acc &) { end_half(); }
    __device__ __forceinline__ void flush() {}
    __device__ __forceinline__ float value(int s) const { return s == 0 ? a : b; }
    __device__ __forceinline__ int code(int s) const {
        return (s == 0 && a > kAbsent) ? half * 16 + (int)(__float_as_uint(a) & 15u) : -1;
    }
};

// Top-k for the bfloat16 filter without a branch per score.  TopList above gates its 24-operation insert
// with a wave ballot; along a stream of a few thousand scores some lane of the wave has a candidate at every
// second score, so the wave pays for the insert most of the time and the branch keeps the vector work
// from overlapping with the MFMAs.  Here the low kTagBits mantissa bits of a score are replaced by its
// position inside an epoch of kEpoch half-steps (half-step within the epoch, accumulator register), so a
// value carries its own code and lists of VALUES only (one v_med3 per slot) do the work:
//   per score    the best three of the current half-step's 16 scores (g1 >= g2 >= g3): 4 operations;
//   per half-step  if no lane's g3 beats what its lists already hold as (K+1)-th best, no lane can have more
//                than two scores of this half-step in its final list, and g1, g2 are inserted into the epoch
//                list (6 operations each); otherwise (the first few hundred scores of a stream, rarely later)
//                all 16 scores are inserted;
//   per epoch    the entries of the epoch list that still matter are merged, with their codes, into the
//                stream's running list (the gated insert, at most K + 1 times), and the epoch list starts
//                empty again.
// The perturbation (below 2^-(23 - kTagBits) relative) is part of the margin the finalize pass tests.
constexpr int kEpoch = 8, kTagBits = 7;                   // 8 half-steps x 16 registers = 128 positions
#ifndef ISPA_TOPK_GROUP
#define ISPA_TOPK_GROUP 1
#endif
template <int K>
struct TagTopK {
    float v[K + 1];                                       // this epoch's best, tagged, descending
    float g1, g2, g3;                                     // this half-step's best three, tagged
    TopList<K> run;                                       // best of the finished epochs, with codes
    int tag_base, cur;
    unsigned keep;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int s = 0; s <= K; ++s) v[s] = kNegInf;
    }
    __device__ __forceinline__ void init() {
        clear();
        run.init();
        tag_base = 0;
        cur = 0;
        g1 = g2 = g3 = kNegInf;
        keep = ~((1u << kTagBits) - 1u);
        // the mask lives in a vector register: v_and_or_b32 may read one scalar operand, and that is the tag
        asm volatile("" : "+v"(keep));
    }
    // merges the epoch whose first half-step is `first` into the running list
    __device__ __forceinline__ void commit(int first) {
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            if (__builtin_amdgcn_ballot_w64(v[s] > run.v[K]) == 0) break;   // sorted: the rest is smaller still
            const int tag = (int)(__float_as_uint(v[s]) & ((1u << kTagBits) - 1u));
            run.begin_half(first + (tag >> 4));
            run.push(v[s], tag & 15);
        }
        clear();
    }
    __device__ __forceinline__ void begin_half(int id) {
        if ((id & (kEpoch - 1)) == 0) commit(id - kEpoch);
        cur = id;
        tag_base = (id & (kEpoch - 1)) << 4;
        g1 = g2 = g3 = kNegInf;
    }
    __device__ __forceinline__ float tagged(float x, int reg) const {
        return __uint_as_float((__float_as_uint(x) & keep) | (unsigned)(tag_base + reg));
    }
    __device__ __forceinline__ void insert(float t) {
#pragma unroll
        for (int s = K; s > 0; --s) asm("v_med3_f32 %0, %1, %2, %3" : "=v"(v[s]) : "v"(t), "v"(v[s - 1]), "v"(v[s]));
        asm("v_max_f32 %0, %1, %2" : "=v"(v[0]) : "v"(t), "v"(v[0]));
    }
    __device__ __forceinline__ void push(float x, int reg) {
        const float t = tagged(x, reg);
        if (ISPA_TOPK_GROUP) {
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(g3) : "v"(t), "v"(g2), "v"(g3));
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(g2) : "v"(t), "v"(g1), "v"(g2));
            asm("v_max_f32 %0, %1, %2" : "=v"(g1) : "v"(t), "v"(g1));
        } else {
            insert(t);
        }
    }
    // `scores`: the 16 accumulator registers pushed since begin_half
    template <typename Acc>
    __device__ __forceinline__ void end_half(const Acc &scores) {
        if (!ISPA_TOPK_GROUP) return;
        const float have = fmaxf(v[K], run.v[K]);         // nothing at or below it can end up in the list
        if (__builtin_amdgcn_ballot_w64(g3 > have) == 0) {
            insert(g1);
            insert(g2);
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) insert(tagged(scores[reg], reg));
        }
    }
    __device__ __forceinline__ void flush() { commit(cur & ~(kEpoch - 1)); }
    __device__ __forceinline__ float value(int s) const { return run.value(s); }
    __device__ __forceinline__ int code(int s) const { return run.code(s); }
};

template <int K, bool TAGGED>
struct ListFor {
    typedef TopList<K> type;
};
template <int K>
struct ListFor<K, true> {           // top-k: the tagged value lists (defined above for the bfloat16 filter)
    typedef TagTopK<K> type;
};
template <>
struct ListFor<1, true> {
    typedef TagTop1 type;
};
// list of the bfloat16 filter: tagged for every k
template <int K, int KS>
struct BfListFor {
    typedef TagTopK<K> type;
};
template <int KS>
struct BfListFor<1, KS> {
    typedef TagTop1 type;
};

// Guarded tile loader: rows = channels [c0, c0+kKC), columns = [n0, n0+128) of x (C, n), converted
// to float; kKC rows x 128 columns in groups of 4, kGroups groups per thread.  load_tile only issues
// the global loads (into registers), store_tile writes them to LDS a stage later.
template <typename T>
__device__ __forceinline__ void load_tile(const T *__restrict__ x, int C, int64_t n, int c0, int64_t n0,
                                          float4 (&regs)[kGroups], bool vec_ok) {
#pragma unroll
    for (int e = 0; e < kGroups; ++e) {
        const int g = threadIdx.x + kThreads * e;
        const int row = g >> 5;
        const int col = (g & 31) * 4;
        const int c = c0 + row;
        const int64_t nn = n0 + col;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            const T *src = x + (int64_t)c * n + nn;
            if (vec_ok && nn + 3 < n) {
                if (sizeof(T) == 4) {
                    val = *reinterpret_cast<const float4 *>(src);
                } else {
                    const uint2 raw = *reinterpret_cast<const uint2 *>(src);
                    val.x = __uint_as_float(raw.x << 16);
                    val.y = __uint_as_float(raw.x & 0xffff0000u);
                    val.z = __uint_as_float(raw.y << 16);
                    val.w = __uint_as_float(raw.y & 0xffff0000u);
                }
            } else {
                if (nn < n) val.x = load_as_float(src);
                if (nn + 1 < n) val.y = load_as_float(src + 1);
                if (nn + 2 < n) val.z = load_as_float(src + 2);
                if (nn + 3 < n) val.w = load_as_float(src + 3);
            }
        }
        regs[e] = val;
    }
}

__device__ __forceinline__ void store_tile(const float4 (&regs)[kGroups], float (*__restrict__ dst)[kTQ]) {
#pragma unroll
    for (int e = 0; e < kGroups; ++e) {
        const int g = threadIdx.x + kThreads * e;
        *reinterpret_cast<float4 *>(&dst[g >> 5][(g & 31) * 4]) = regs[e];
    }
}

// DIRECT: tiles go global -> LDS with LDS-direct loads (no staging registers, no ds_write); needs
// float input, 16-byte aligned rows and C a multiple of kKC.  Otherwise tiles pass through
// registers with per-element guards and conversion.
template <typename T, int K, bool DIRECT>
__global__ __launch_bounds__(kThreads, K == 1 ? 4 : 2) void nn_filter_f32_kernel(
    const T *__restrict__ q, const T *__restrict__ r, const float *__restrict__ rnh, int C, int64_t nq,
    int64_t nr, int n_qtiles, int nsplit, int64_t refs_per_split, float *__restrict__ part_val,
    int *__restrict__ part_idx, int vec_ok) {
    // two LDS stages of (kKC channels x 128 queries, kKC channels x 128 references): while the MFMAs
    // consume one, the global loads of the next are in flight into registers
    __shared__ __attribute__((aligned(16))) float Qs[2][kKC][kTQ];
    __shared__ __attribute__((aligned(16))) float Rs[2][kKC][kTR];
    __shared__ float Nh[2][kTR];

    // block -> (pair, split, query tile)
    const FilterWork fw = filter_work(n_qtiles, nsplit, refs_per_split, nr, kTQ, kTR);
    const int split = fw.split, b = fw.b;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const T *qb = q + (int64_t)b * C * nq;
    const T *rb = r + (int64_t)b * C * nr;
    const float *nhb = rnh + (int64_t)b * nr;
    const int64_t q0 = fw.q0, rbeg = fw.rbeg, rend = fw.rend;
    const int nchunks = fw.nchunks;
    const int nk = (C + kKC - 1) / kKC;
    const int total = nchunks * nk;

    typename ListFor<K, true>::type best[2];
    best[0].init();
    best[1].init();

    float4 qreg[kGroups], rreg[kGroups];
    float nh_reg = kNegInf;
    // LDS-direct path: one wave instruction moves 1 KiB = 2 tile rows; this wave owns kKC/4
    // consecutive rows of the query tile and of the reference tile.  Columns past the end of the map
    // re-read column 0 (finite values; masked references / unwritten queries).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int d_row = lane >> 5, d_col = (lane & 31) * 4;
    // the row part of the address is uniform (scalar base), the lane part a 32-bit offset
    auto direct_tile = [&](const T *x, int64_t n, int c0, int64_t n0, float (*dst)[kTQ]) {
        const int64_t col = n0 + d_col < n ? n0 + d_col : 0;
        const unsigned voff = (unsigned)(((int64_t)d_row * n + col) * 4);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(&dst[0][0]);
#pragma unroll
        for (int e = 0; e < kKC / 8; ++e) {
            const int row = (kKC / 4) * wave_u + 2 * e;
            const uint64_t sbase = (uint64_t)x + (uint64_t)(c0 + row) * (uint64_t)n * 4u;
            const unsigned ldsaddr = __builtin_amdgcn_readfirstlane(lds0 + row * kTQ * 4);
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(voff), "s"(ldsaddr), "s"(sbase)
                : "memory");
        }
    };
    // fetch(chunk, ks, buf): start moving stage (chunk, ks) towards LDS buffer `buf`
    auto fetch = [&](int chunk, int ks, int buf) {
        if (DIRECT) {
            if (ISPA_ABLATE != 7 || chunk == 0) direct_tile(qb, nq, ks * kKC, q0, Qs[buf]);
            direct_tile(rb, nr, ks * kKC, rbeg + (int64_t)chunk * kTR, Rs[buf]);
        } else {
            load_tile(qb, C, nq, ks * kKC, q0, qreg, vec_ok);
            load_tile(rb, C, nr, ks * kKC, rbeg + (int64_t)chunk * kTR, rreg, vec_ok);
        }
        if (ks == 0 && threadIdx.x < kTR) {
            const int64_t j = rbeg + (int64_t)chunk * kTR + threadIdx.x;
            nh_reg = j < rend ? nhb[j] : kNegInf;
        }
    };
    // park(buf, chunk, ks): make that stage complete in LDS (followed by the block barrier)
    auto park = [&](int buf, int chunk, int ks) {
        if (DIRECT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            store_tile(qreg, Qs[buf]);
            store_tile(rreg, Rs[buf]);
        }
        if (ks == 0 && threadIdx.x < kTR) Nh[chunk & 1][threadIdx.x] = nh_reg;
    };
    if (total > 0) {
        fetch(0, 0, 0);
        park(0, 0, 0);
    }
    __syncthreads();

    int st = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        f32x16 acc[2][2];
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float init = Nh[chunk & 1][wr * 64 + ri * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h];
                acc[ri][0][reg] = init;
                acc[ri][1][reg] = init;
            }
        for (int ks = 0; ks < nk; ++ks, ++st) {
            const int buf = st & 1;
            const bool more = st + 1 < total;
            const int nks = ks + 1 < nk ? ks + 1 : 0;
            const int nchunk = ks + 1 < nk ? chunk : chunk + 1;
            if (more && ISPA_ABLATE != 5 && ISPA_ABLATE != 6) fetch(nchunk, nks, buf ^ 1);
            // LDS operand reads run one k-step ahead of the MFMAs that use them
            float a0 = Rs[buf][h][wr * 64 + l31], a1 = Rs[buf][h][wr * 64 + 32 + l31];
            float b0 = Qs[buf][h][wq * 64 + l31], b1 = Qs[buf][h][wq * 64 + 32 + l31];
#pragma unroll
            for (int kk = 0; kk < kKC; kk += 2) {
                float a0n = a0, a1n = a1, b0n = b0, b1n = b1;
                if (kk + 2 < kKC) {
                    a0n = Rs[buf][kk + 2 + h][wr * 64 + l31];
                    a1n = Rs[buf][kk + 2 + h][wr * 64 + 32 + l31];
                    b0n = Qs[buf][kk + 2 + h][wq * 64 + l31];
                    b1n = Qs[buf][kk + 2 + h][wq * 64 + 32 + l31];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                a0 = a0n;
                a1 = a1n;
                b0 = b0n;
                b1 = b1n;
            }
            if (more && ISPA_ABLATE != 5 && ISPA_ABLATE != 6) park(buf ^ 1, nchunk, nks);
            __syncthreads();
        }
        // this lane owns query column (wq*64 + qi*32 + l31) and 16 reference rows per tile
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                best[qi].begin_half(chunk * 2 + ri);
#pragma unroll
                for (int reg = 0; reg < 16; reg += 4) {
                    if (ISPA_ABLATE == 4 || ISPA_ABLATE == 6) {
                        asm volatile("" ::"v"(acc[ri][qi][reg]), "v"(acc[ri][qi][reg + 1]), "v"(acc[ri][qi][reg + 2]),
                                     "v"(acc[ri][qi][reg + 3]));
                    } else if constexpr (K == 1) {
                        best[qi].push4(acc[ri][qi], reg, reg);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) best[qi].push(acc[ri][qi][reg + e], reg + e);
                    }
                }
                best[qi].end_half(acc[ri][qi]);
            }
    }
    best[0].flush();
    best[1].flush();

    // one partial stream per (split, wr, h); finalize merges them
    const int nparts = nsplit * kSubStreams;
    const int part = split * kSubStreams + wr * 2 + h;
    // code = (chunk * 2 + row tile) * 16 + accumulator register
    auto to_ref = [&](int code) {
        const int reg = code & 15, ri = (code >> 4) & 1;
        return (int)(rbeg + (int64_t)(code >> 5) * kTR + wr * 64 + ri * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
    };
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
        store_partial<K + 1>(best[qi], to_ref, part_val, part_idx, b, nparts, part, nq, q0 + wq * 64 + qi * 32 + l31);
}

// ---------------------------------------------------------------------------------------------
// 2a'. float32 filter with the query fragments resident in registers (C = 16 * NKS <= 128, float
// input, aligned rows).  Same tile (128 queries x 128-reference chunks, 2 x 2 waves of 64 x 64) and
// the same MFMA chain as nn_filter_f32_kernel, but a wave loads its 64 queries once, straight into
// MFMA operand layout (C/2 registers per 32-query tile), so only reference stages pass through LDS:
// a ring of four 16-channel stages filled by LDS-direct loads three stages ahead.  Two blocks per CU.
// ---------------------------------------------------------------------------------------------
constexpr int kRing = 4;      // reference stages in LDS; loads run kRing - 1 stages ahead

// One 16-channel x 128-reference stage -> LDS ring slot, two LDS-direct instructions per wave (1 KiB =
// 2 rows each; this wave owns rows 4 wave .. 4 wave + 3).  The row part of the address is uniform
// and goes into the scalar base, the per-lane part (row parity, column) is the 32-bit offset `voff`,
// so a stage costs no vector address arithmetic.
__device__ __forceinline__ unsigned stage_voff(int64_t nr, int64_t n0, int lane) {
    const int d_row = lane >> 5, d_col = (lane & 31) * 4;
    const int64_t col = n0 + d_col < nr ? n0 + d_col : 0;   // past the end: column 0 (masked by the norms)
    return (unsigned)(((int64_t)d_row * nr + col) * 4);
}
__device__ __forceinline__ void stage_fetch(const float *rb, int64_t nr, int ks, int wave_u, unsigned voff,
                                            unsigned lds_slot) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 4 * wave_u + 2 * e;
        const uint64_t sbase = (uint64_t)rb + (uint64_t)(ks * kKC + row) * (uint64_t)nr * 4u;
        const unsigned ldsaddr = __builtin_amdgcn_readfirstlane(lds_slot + row * kTR * 4);
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(ldsaddr), "s"(sbase)
            : "memory");
    }
}

template <int NKS, int K>
__global__ __launch_bounds__(kThreads, 2) void nn_filter_f32_qreg_kernel(
    const float *__restrict__ q, const float *__restrict__ r, const float *__restrict__ rnh, int64_t nq, int64_t nr,
    int n_qtiles, int nsplit, int64_t refs_per_split, float *__restrict__ part_val, int *__restrict__ part_idx) {
    constexpr int C = NKS * kKC;
    __shared__ __attribute__((aligned(16))) float Rs[kRing][kKC][kTR];
    __shared__ __attribute__((aligned(16))) float Nh[2][kTR];

    const FilterWork fw = filter_work(n_qtiles, nsplit, refs_per_split, nr, kTQ, kTR);
    const int split = fw.split, b = fw.b;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const float *qb = q + (int64_t)b * C * nq;
    const float *rb = r + (int64_t)b * C * nr;
    const float *nhb = rnh + (int64_t)b * nr;
    const int64_t q0 = fw.q0, rbeg = fw.rbeg, rend = fw.rend;
    const int nchunks = fw.nchunks;
    const int total = nchunks * NKS;

    // reference stage `st` (chunk st / NKS, channels 16 (st % NKS) ..) -> ring slot; one wave
    // instruction moves 1 KiB = 2 rows, this wave owns rows 4 wave .. 4 wave + 3.  Stages past the
    // end re-fetch the last one (every wave issues the same number of loads, which the vmcnt
    // arithmetic below relies on); columns past the end of the range re-read column 0.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(&Rs[0][0][0]);
    // stage (chunk, ks) -> ring slot; chunks past the end re-fetch the last one into a slot nobody
    // reads (every wave issues the same number of loads, which the vmcnt arithmetic relies on)
    auto chunk_voff = [&](int chunk) {
        const int c = chunk < nchunks ? chunk : nchunks - 1;
        return stage_voff(nr, rbeg + (int64_t)c * kTR, lane);
    };
    auto fetch = [&](unsigned voff, int ks, int slot) {
        stage_fetch(rb, nr, ks, wave_u, voff, lds0 + slot * (kKC * kTR * 4));
    };
    auto load_nh = [&](int chunk) {
        const int64_t j = rbeg + (int64_t)chunk * kTR + (threadIdx.x & (kTR - 1));
        return (chunk < nchunks && j < rend) ? nhb[j] : kNegInf;
    };

    // query fragments: lane (l31, h) of k-step kk holds channel 2 kk + h of query column l31
    float bq[2][C / 2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int64_t col = q0 + wq * 64 + qi * 32 + l31;
        const float *src = qb + (col < nq ? col : 0);
#pragma unroll
        for (int kk = 0; kk < C / 2; ++kk) bq[qi][kk] = src[(int64_t)(2 * kk + h) * nq];
    }
    if (total > 0) Nh[0][threadIdx.x & (kTR - 1)] = load_nh(0);
    float nh_next = load_nh(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // plain loads done before the counted ones start
    if (total > 0) {
#pragma unroll
        for (int e = 0; e < kRing - 1; ++e) fetch(chunk_voff(e / NKS), e % NKS, e);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kRing - 2)) : "memory");   // stage 0 landed
    }
    __syncthreads();

    typename ListFor<K, true>::type best[2];
    best[0].init();
    best[1].init();

    int st = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        f32x16 acc[2][2];
        // register reg of a tile is reference row (reg & 3) + 8 (reg >> 2) + 4 h: four consecutive
        // floats per reg >> 2, one 16-byte LDS read each
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t = *reinterpret_cast<const float4 *>(&Nh[chunk & 1][wr * 64 + ri * 32 + 8 * g4 + 4 * h]);
                acc[ri][0][4 * g4 + 0] = acc[ri][1][4 * g4 + 0] = t.x;
                acc[ri][0][4 * g4 + 1] = acc[ri][1][4 * g4 + 1] = t.y;
                acc[ri][0][4 * g4 + 2] = acc[ri][1][4 * g4 + 2] = t.z;
                acc[ri][0][4 * g4 + 3] = acc[ri][1][4 * g4 + 3] = t.w;
            }
        unsigned voff[(NKS + kRing - 2) / NKS + 1];
#pragma unroll
        for (int d = 0; d <= (NKS + kRing - 2) / NKS; ++d) voff[d] = chunk_voff(chunk + d);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks, ++st) {
            const int slot = st & (kRing - 1);
            // the slot of stage st - 1 is free: every wave passed the barrier that ended it
            fetch(voff[(ks + kRing - 1) / NKS], (ks + kRing - 1) % NKS, (st + kRing - 1) & (kRing - 1));
            float a0 = Rs[slot][h][wr * 64 + l31], a1 = Rs[slot][h][wr * 64 + 32 + l31];
#pragma unroll
            for (int kk = 0; kk < kKC; kk += 2) {
                float a0n = a0, a1n = a1;
                if (kk + 2 < kKC) {
                    a0n = Rs[slot][kk + 2 + h][wr * 64 + l31];
                    a1n = Rs[slot][kk + 2 + h][wr * 64 + 32 + l31];
                }
                const float b0 = bq[0][ks * (kKC / 2) + kk / 2], b1 = bq[1][ks * (kKC / 2) + kk / 2];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                a0 = a0n;
                a1 = a1n;
            }
            if (ks == NKS - 1) {                         // norms of the next chunk, read a chunk ago
                Nh[(chunk + 1) & 1][threadIdx.x & (kTR - 1)] = nh_next;
                nh_next = load_nh(chunk + 2);
            }
            // stage st + 1 landed: at most the kRing - 2 youngest stages (and the norm load) outstanding
            if (ks == NKS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kRing - 2) + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kRing - 2)) : "memory");
            __syncthreads();
        }
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                best[qi].begin_half(chunk * 2 + ri);
#pragma unroll
                for (int reg = 0; reg < 16; reg += 4) {
                    if constexpr (K == 1) {
                        best[qi].push4(acc[ri][qi], reg, reg);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) best[qi].push(acc[ri][qi][reg + e], reg + e);
                    }
                }
                best[qi].end_half(acc[ri][qi]);
            }
    }
    best[0].flush();
    best[1].flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may still be landing in LDS at exit

    const int nparts = nsplit * kSubStreams;
    const int part = split * kSubStreams + wr * 2 + h;
    // code = (chunk * 2 + row tile) * 16 + accumulator register
    auto to_ref = [&](int code) {
        const int reg = code & 15, ri = (code >> 4) & 1;
        return (int)(rbeg + (int64_t)(code >> 5) * kTR + wr * 64 + ri * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
    };
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
        store_partial<K + 1>(best[qi], to_ref, part_val, part_idx, b, nparts, part, nq, q0 + wq * 64 + qi * 32 + l31);
}

// ---------------------------------------------------------------------------------------------
// 2b. filter pass, bfloat16 MFMA (v_mfma_f32_32x32x16_bf16, float32 accumulation)
//
// One 256-thread block = 256 queries x (a split of) all references; wave w owns queries
// [64w, 64w+64) and every wave sees every reference.  Two such blocks fit on a CU, and because they
// run out of phase one block's chunk boundary (barrier, LDS refill) is covered by the other's MFMAs.
// Both operands are channel-major in memory (C rows of cells), the TRANSPOSE of what the MFMA
// fragments want (8 consecutive channels per lane), so tiles are copied to LDS as they are and read
// back with ds_read_b64_tr_b16.  A wave's query fragments are read once and stay in registers
// (C/2 VGPRs) for the whole kernel; reference chunks of 64 cells stream through a ring of three
// LDS images, fetched into registers two chunks ahead and written to LDS one chunk ahead, with the
// block barrier in the MIDDLE of a chunk so that no wave ever waits on an LDS read right after it.  LDS rows are padded so that the four
// rows a transposed read touches fall on four different bank quarters.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One LDS-direct instruction (64 lanes x 16 bytes -> 1 KiB of LDS at `lds_dst`) with the uniform part
// of the address in a scalar base and the lane part in a 32-bit byte offset: no vector address
// arithmetic per load.  Issued from inline asm so that the compiler neither counts it nor orders LDS
// reads behind it; callers wait with an explicit vmcnt.
__device__ __forceinline__ void glds_b128(unsigned voff, uint64_t sbase, unsigned lds_dst) {
    unsigned keep;
    // wave-uniform by construction; said explicitly so that the operands stay scalar registers
    // whatever the optimiser concludes about the control flow around the call
    // (the builtin returns int: widen through uint32_t, not by sign extension)
    const uint32_t base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sbase >> 32));
    const uint32_t base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sbase);
    sbase = ((uint64_t)base_hi << 32) | (uint64_t)base_lo;
    lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_dst), "s"(sbase)
        : "memory");
}
// The same instruction for a loop that issues many of them: the LDS destination is `lds_base + IMM`, added
// straight into m0 (one scalar instruction), and m0 is NOT restored.  Nothing else in these kernels uses m0
// (gfx9 LDS instructions do not read it), so the save / restore pair of glds_b128 is two scalar issue slots
// per load for nothing; the callers keep `sbase` and `lds_base` in scalar registers themselves.
template <int IMM>
__device__ __forceinline__ void glds_b128_lean(unsigned voff, uint64_t sbase, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM)
                 : "memory");
}
template <int E, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (E < N) {
        f(std::integral_constant<int, E>{});
        static_for<E + 1, N>(f);
    }
}
// 16-byte segment (of the 8 in a 64-reference chunk row) that lane `lane` fetches when the 32-byte
// slots of a row are permuted by (row bit 1) | (row bit 3) << 1; row bit 3 is the parity `par` of the
// instruction index
__device__ __forceinline__ int perm_seg(int lane, int par) {
    const int perm = ((lane >> 4) & 1) | (par << 1);
    return ((((lane & 7) >> 1) ^ perm) << 1) | (lane & 1);
}

constexpr int kBQ = 256;             // queries per block
constexpr int kBR = 64;              // references per chunk
constexpr int kBSubStreams = 2;      // partial result streams per (tile, split): the two lane halves
constexpr int kRStride = 128;        // bytes per LDS row of 64 bf16 reference cells (swizzled, not padded)
constexpr int kQStride = 512 + 64;   // bytes per LDS row of 256 bf16 query cells

__device__ __forceinline__ bf16x8 tr_frag(const char *lds_row0_col, int stride) {
    // rows k..k+3 then k+4..k+7 of the 16-column block this lane group addresses
    typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds_row0_col));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds_row0_col + 4 * stride));
    return u.v;
}

template <int KS>
constexpr int bf16_filter_lds_bytes() {
    constexpr int stream = 3 * 16 * KS * kRStride + 3 * kBR * (int)sizeof(float);
    constexpr int qimage = 16 * KS * kQStride;
    return stream > qimage ? stream : qimage;
}

// QT = 32-query tiles per wave.  QT = 2: 4 waves of 64 queries (two workgroups per CU up to 128 channels, one
// above).  QT = 1: 8 waves of 32 queries for more than 128 channels, where the LDS images leave room for one
// workgroup only: two waves per SIMD, so that one wave's LDS reads and list updates run under the other's MFMAs
// (with one wave per SIMD nothing hides them).
template <int KS, int K, int QT>
__global__ __launch_bounds__(512 / QT, (KS > 8 || QT == 1) ? 1 : 2) void nn_filter_bf16_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ r, const float *__restrict__ rnh, int64_t nq,
    int64_t nr, int n_qtiles, int nsplit, int64_t refs_per_split, float *__restrict__ part_val,
    int *__restrict__ part_idx) {
    constexpr int C = 16 * KS;
    constexpr int kRBuf = C * kRStride;                 // one reference chunk image
    constexpr int kNT = 512 / QT;                       // threads
    constexpr int kSeg = KS * QT / 4;                   // 1 KiB pieces of a chunk image per wave
    static_assert(QT == 2 || KS % 4 == 0, "8 waves share 2 KS pieces evenly only if KS is a multiple of 4");
    // up to 128 channels two blocks share a CU; above that the query fragments alone take 128
    // registers and the LDS images 144 KiB, so one block owns the CU
    static_assert(KS % 2 == 0 && KS <= 16, "C must be a multiple of 32, at most 256");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *rbuf = lds;                                   // ring of 3 chunk images
    float *nhbuf = reinterpret_cast<float *>(lds + 3 * kRBuf);   // ring of 3 x kBR floats

    const FilterWork fw = filter_work(n_qtiles, nsplit, refs_per_split, nr, kBQ, kBR);
    const int split = fw.split, b = fw.b;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wq = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // transposed-read addressing: 16-lane group g covers columns 16*(g&1).., channels 8*(g>>1)..
    const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int tr_row = 8 * h + ((lane & 15) >> 2);

    const uint16_t *qb = q + (int64_t)b * C * nq;
    const uint16_t *rb = r + (int64_t)b * C * nr;
    const float *nhb = rnh + (int64_t)b * nr;
    const int64_t q0 = fw.q0, rbeg = fw.rbeg, rend = fw.rend;
    const int nchunks = fw.nchunks;

    // ---- query tile -> LDS -> per-wave fragments in registers --------------------------------
#pragma unroll
    for (int e = 0; e < KS * QT; ++e) {                  // C rows x 32 segments of 8 queries
        const int seg = tid + kNT * e;
        const int row = seg >> 5, col = (seg & 31) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q0 + col < nq) v = *reinterpret_cast<const uint4 *>(qb + (int64_t)row * nq + q0 + col);
        *reinterpret_cast<uint4 *>(lds + row * kQStride + col * 2) = v;
    }
    // Reference chunk -> LDS slot, 1 KiB (8 rows x 128 B) per wave instruction, kSeg instructions
    // per wave.  Lane l lands on row l>>3, 16-byte segment l&7 of its 8 rows; it fetches the
    // segment whose index has bit 2 flipped when the row index has bit 1 set (the half swap).
    // Columns past the end of this block's reference range re-read the row start: real, finite
    // feature values whose scores the masked norm entry pushes out of contention.
    // The loads are issued from inline asm: the compiler neither counts them nor orders LDS reads
    // behind them (it would wait for every LDS-direct load before the next ds_read); the wait is the
    // explicit vmcnt(0) in front of the mid-chunk barrier.
    const int st_seg = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    const int wq_u = __builtin_amdgcn_readfirstlane(wq);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)rbuf;
    const unsigned row_off = (unsigned)((int64_t)(lane >> 3) * nr * 2);        // bytes; nr < 2^27 on this path
    const unsigned lane_off = row_off + st_seg * 16;
    auto stage = [&](int64_t r0, int slot) {
        // uniform: only a last chunk can be ragged.  Steady state: the chunk's column offset goes into the lane
        // offset (one vector add), the scalar bases of the kSeg pieces are loop invariants, m0 = slot base + piece
        if (__builtin_expect((int)(rend - r0) >= kBR, 1)) {
            const unsigned col2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)r0 * 2u));   // < 2^28
            const unsigned v0 = lane_off + col2;
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + slot * kRBuf + wq_u * kSeg * 1024));
            static_for<0, kSeg>([&](auto ic) {
                constexpr int e = decltype(ic)::value;
                const int id = wq_u * kSeg + e;          // rows [8 id, 8 id + 8)
                uint64_t sbase = (uint64_t)rb + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sbase >> 32));
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sbase);
                sbase = ((uint64_t)hi << 32) | lo;
                glds_b128_lean<e * 1024>(v0, sbase, sb);
            });
        } else {
            const unsigned voff = row_off + (r0 + st_seg * 8 < rend ? (unsigned)(r0 + st_seg * 8) * 2 : 0);
#pragma unroll
            for (int e = 0; e < kSeg; ++e) {
                const int id = wq_u * kSeg + e;
                const uint64_t sbase = (uint64_t)rb + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                glds_b128(voff, sbase, __builtin_amdgcn_readfirstlane(lds_base + slot * kRBuf + id * 1024));
            }
        }
    };
    // norm entry of the chunk after next, parked in a register; the masking is applied when it is
    // written to LDS, so that nothing waits for the load (and, in order, for the LDS-direct loads
    // issued before it) early
    float pre_nh = 0.f;
    bool pre_ok = false;
    auto fetch_nh = [&](int64_t r0) {
        pre_ok = r0 + (tid & (kBR - 1)) < rend;
        pre_nh = nhb[pre_ok ? r0 + (tid & (kBR - 1)) : 0];
    };
    auto clamp_chunk = [&](int c) { return rbeg + (int64_t)(c < nchunks ? c : nchunks - 1) * kBR; };
    __syncthreads();
    bf16x8 bq[QT][KS];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bq[qi][ks] = tr_frag(lds + (ks * 16 + tr_row) * kQStride + (wq * 32 * QT + qi * 32 + tr_col) * 2, kQStride);
    __syncthreads();                                     // query image no longer needed
    if (nchunks > 0) {
        stage(clamp_chunk(0), 0);
        stage(clamp_chunk(1), 1);
        if (wq_u == 0) fetch_nh(clamp_chunk(0));
        if (wq_u == 0) nhbuf[tid & (kBR - 1)] = pre_ok ? pre_nh : kNegInf;
        if (wq_u == 0) fetch_nh(clamp_chunk(1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    typename BfListFor<K, KS>::type best[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) best[qi].init();

    // Each chunk is two half-steps (its reference rows 0-31 and 32-63).  The MFMAs of one half-step
    // are interleaved, in a pinned program order, with the top-list update of the previous
    // half-step's accumulators, so the matrix pipe and the vector ALU work at the same time; the
    // transposed LDS reads run two k-steps ahead of their MFMAs.
    f32x16 acc_a[QT], acc_b[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc_b[qi][reg] = kNegInf;               // nothing pending yet
        // opaque to the optimiser: otherwise it peels the first iteration to fold these constants and
        // the pinned MFMA / vector interleave of the steady-state loop is lost
        asm volatile("" : "+v"(acc_b[qi]));
    }

    auto load_init = [&](const float *nh, int ri) {
        f32x16 v;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4 *>(nh + ri * 32 + 8 * g + 4 * h);
            v[4 * g + 0] = t.x;
            v[4 * g + 1] = t.y;
            v[4 * g + 2] = t.z;
            v[4 * g + 3] = t.w;
        }
        return v;
    };
    const int rd_swap = (lane >> 3) & 1;                 // bit 1 of the channel rows this lane addresses
    auto frag = [&](const char *rimg, int ri, int ks) {
        return tr_frag(rimg + (ks * 16 + tr_row) * kRStride + ((ri ^ rd_swap) * 32 + tr_col) * 2, kRStride);
    };
    // MFMAs of half-step (`rimg`, `ri`) into `now`, top-list update of `prev` (half-step id
    // prev_half).  a0 / a1 hold the fragments of k-steps 0 and 1 on entry and those of the next
    // half-step (`next_img`, `next_ri`) on exit.
    auto half_step = [&](const char *rimg, int ri, const char *next_img, int next_ri, bf16x8 &a0, bf16x8 &a1,
                         const f32x16 &init, f32x16 (&now)[QT], const f32x16 (&prev)[QT], int prev_half) {
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) best[qi].begin_half(prev_half);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a2 = ks + 2 < KS ? frag(rimg, ri, ks + 2) : frag(next_img, next_ri, ks + 2 - KS);
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                now[qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[qi][ks], ks ? now[qi] : init, 0, 0, 0);
#pragma unroll
                for (int reg = ks * 16 / KS; reg < (ks + 1) * 16 / KS; ++reg) {
                    if (ISPA_ABLATE == 1) asm volatile("" ::"v"(prev[qi][reg]));
                    else if constexpr (K == 1) best[qi].push4_step(reg & 3, prev[qi], reg & ~3, reg & ~3);
                    else best[qi].push(prev[qi][reg], reg);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1;
            a1 = a2;
        }
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) best[qi].end_half(prev[qi]);
    };

    // chunk c lives in ring slot c % 3.  Mid-chunk c: the norm entries of chunk c+1 go to LDS, all
    // LDS-direct loads of chunk c+1 (issued a chunk ago) are waited for, barrier; then the loads of
    // chunk c+2 are issued into the slot whose last readers worked on chunk c-1 and have all passed
    // this barrier.  Everything read right after the barrier was complete before it and is not
    // needed for another half-step.
    bf16x8 a0, a1;
    f32x16 init_a;
    if (nchunks > 0) {
        a0 = frag(rbuf, 0, 0);
        a1 = frag(rbuf, 0, 1);
        init_a = load_init(nhbuf, 0);
    }
    int slot = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int next_slot = slot == 2 ? 0 : slot + 1;
        const int after_slot = next_slot == 2 ? 0 : next_slot + 1;
        const char *rimg = rbuf + slot * kRBuf;
        const char *next_img = rbuf + next_slot * kRBuf;
        const f32x16 init_b = load_init(nhbuf + slot * kBR, 1);
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 0, rimg, 1, a0, a1, init_a, acc_a, acc_b, chunk * 2 - 1);
        if (wq_u == 0) nhbuf[next_slot * kBR + (tid & (kBR - 1))] = pre_ok ? pre_nh : kNegInf;   // one wave stages the norm entries
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ISPA_ABLATE != 2) stage(clamp_chunk(chunk + 2), after_slot);
        if (wq_u == 0) fetch_nh(clamp_chunk(chunk + 2));
        init_a = load_init(nhbuf + next_slot * kBR, 0);
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 1, next_img, 0, a0, a1, init_b, acc_b, acc_a, chunk * 2);
        slot = next_slot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may still be landing in LDS at exit
    if (nchunks > 0) {
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            best[qi].begin_half(nchunks * 2 - 1);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) best[qi].push(acc_b[qi][reg], reg);
            best[qi].end_half(acc_b[qi]);
            best[qi].flush();
        }
    }

    // one partial stream per (split, lane half); finalize merges them
    const int nparts = nsplit * kBSubStreams;
    const int part = split * kBSubStreams + h;
    // code = half-step * 16 + accumulator register
    auto to_ref = [&](int code) {
        const int reg = code & 15;
        return (int)(rbeg + (int64_t)(code >> 4) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
    };
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
        store_partial<K + 1>(best[qi], to_ref, part_val, part_idx, b, nparts, part, nq,
                             q0 + wq * 32 * QT + qi * 32 + l31);
}

// ---------------------------------------------------------------------------------------------
// 2c. top-1 filter on v_mfma_f32_16x16x32_bf16
//
// Same block structure as 2b (256 queries, 4 waves of 64 queries, 64-reference chunks in a
// 3-slot LDS ring filled by LDS-direct loads, mid-chunk barrier), on the 16x16x32 MFMA shape,
// which the chip sustains at a higher clock than 32x32x16 on random data.  A wave's 64x32
// half-step is 2 reference tiles x 4 query tiles of 16x16; a lane owns query column lane&15 of
// each query tile (4 query streams) and, per tile, the 4 reference rows 4*(lane>>4)+reg.  The
// tag in the low mantissa bits is tile*4 + reg (3 bits).  LDS reference image: 128-byte rows whose
// four 32-byte slots are permuted by (row bit 1) | (row bit 3) << 1, which puts the 8 channel rows
// a 32-lane half reads at once on 8 different 32-byte bank groups.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kT16SubStreams = 4;    // partial result streams per (tile, split): lane >> 4

// NQ = 16-query tiles per wave.  NQ = 4 (what runs): 4 waves of 64 queries, two workgroups per CU up to 128
// channels (two waves per SIMD, 192 registers each).  NQ = 2 (development switch ISPA_T16_NQ): 8 waves of 32
// queries for the same 256-query block (C a multiple of 64, at most 128), 120 registers, four waves per SIMD.
// Measured slower (1.88 against 1.77 ms per 512 pairs of 64x64x128): the loop is bound by instruction issue,
// not by latency that more resident waves could hide, and a wave with half the query tiles reads the same
// reference fragments and does the same bookkeeping for half the MFMAs.
template <int KS32, int NQ>
__global__ __launch_bounds__(1024 / NQ, KS32 > 4 ? 1 : 8 / NQ) void nn_filter_bf16_t16_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ r, const float *__restrict__ rnh, int64_t nq,
    int64_t nr, int n_qtiles, int nsplit, int64_t refs_per_split, float *__restrict__ part_val,
    int *__restrict__ part_idx) {
    constexpr int C = 32 * KS32;
    constexpr int KS = 2 * KS32;                        // 16-channel units, as in bf16_filter_lds_bytes
    constexpr int kRBuf = C * kRStride;
    constexpr int kNT = 1024 / NQ;                      // threads
    constexpr int kSeg = KS * NQ / 8;                   // LDS-direct instructions per wave per chunk
    static_assert(NQ == 4 || (NQ == 2 && KS32 % 2 == 0), "8 waves share the chunk pieces evenly only if C % 64 == 0");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *rbuf = lds;
    float *nhbuf = reinterpret_cast<float *>(lds + 3 * kRBuf);

    const FilterWork fw = filter_work(n_qtiles, nsplit, refs_per_split, nr, kBQ, kBR);
    const int split = fw.split, b = fw.b;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wq = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // transposed-read addressing: every 16-lane group covers the same 16 columns, group g the
    // channels 8g .. 8g+7 of a 32-channel k-step
    const int tr_col = 4 * (lane & 3);
    const int tr_row = 8 * g + ((lane & 15) >> 2);

    const uint16_t *qb = q + (int64_t)b * C * nq;
    const uint16_t *rb = r + (int64_t)b * C * nr;
    const float *nhb = rnh + (int64_t)b * nr;
    const int64_t q0 = fw.q0, rbeg = fw.rbeg, rend = fw.rend;
    const int nchunks = fw.nchunks;

    // ---- query tile -> LDS -> per-wave fragments in registers --------------------------------
#pragma unroll
    for (int e = 0; e < KS * NQ / 2; ++e) {
        const int seg = tid + kNT * e;
        const int row = seg >> 5, col = (seg & 31) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q0 + col < nq) v = *reinterpret_cast<const uint4 *>(qb + (int64_t)row * nq + q0 + col);
        *reinterpret_cast<uint4 *>(lds + row * kQStride + col * 2) = v;
    }
    // reference chunk -> LDS slot: lane l lands on row l>>3, physical 16-byte segment l&7 of the
    // 8 rows its instruction covers and fetches the segment whose 32-byte slot index is XORed
    // with (row bit 1) | (row bit 3) << 1; row bit 3 is the parity of the instruction index
    const int wq_u = __builtin_amdgcn_readfirstlane(wq);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)rbuf;
    const unsigned row_off = (unsigned)((int64_t)(lane >> 3) * nr * 2);        // bytes; nr < 2^27 on this path
    const unsigned lane_off[2] = {row_off + perm_seg(lane, 0) * 16, row_off + perm_seg(lane, 1) * 16};
    auto stage = [&](int64_t r0, int slot) {
        // uniform: only a last chunk can be ragged.  Two code paths (the loads are volatile asm, so the
        // compiler keeps the branch): the steady state pays no vector arithmetic for the ragged case
        if (__builtin_expect((int)(rend - r0) >= kBR, 1)) {      // 32-bit scalar compare (a split is < 2^27 long)
            // the chunk's column offset goes into the lane offsets (two vector adds per chunk), so that the
            // scalar bases of the kSeg pieces are loop invariants; m0 = slot base + piece offset
            const unsigned col2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)r0 * 2u));   // < 2^28
            const unsigned v0 = lane_off[0] + col2, v1 = lane_off[1] + col2;
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + slot * kRBuf + wq_u * kSeg * 1024));
            static_for<0, kSeg>([&](auto ic) {
                constexpr int e = decltype(ic)::value;
                const int id = wq_u * kSeg + e;          // rows [8 id, 8 id + 8)
                uint64_t sbase = (uint64_t)rb + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sbase >> 32));
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sbase);
                sbase = ((uint64_t)hi << 32) | lo;
                glds_b128_lean<e * 1024>((id & 1) ? v1 : v0, sbase, sb);
            });
        } else {
#pragma unroll
            for (int e = 0; e < kSeg; ++e) {
                const int id = wq_u * kSeg + e;
                const int64_t col = r0 + perm_seg(lane, id & 1) * 8;
                const unsigned voff = row_off + (col < rend ? (unsigned)col * 2 : 0);
                const uint64_t sbase = (uint64_t)rb + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                glds_b128(voff, sbase, __builtin_amdgcn_readfirstlane(lds_base + slot * kRBuf + id * 1024));
            }
        }
    };
    float pre_nh = 0.f;
    bool pre_ok = false;
    const int nh_lane = tid & (kBR - 1);
    auto fetch_nh = [&](int64_t r0) {
        // 32-bit lane arithmetic on a uniform base (rend - r0 fits an int: a split is at most nr < 2^27 long)
        const int left = (int)(rend - r0);
        pre_ok = nh_lane < left;
        pre_nh = (nhb + r0)[pre_ok ? nh_lane : 0];
    };
    auto clamp_chunk = [&](int c) { return rbeg + (int64_t)(c < nchunks ? c : nchunks - 1) * kBR; };
    __syncthreads();
    bf16x8 bq[NQ][KS32];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks)
            bq[qt][ks] = tr_frag(lds + (ks * 32 + tr_row) * kQStride + (wq * 16 * NQ + qt * 16 + tr_col) * 2, kQStride);
    __syncthreads();
    if (nchunks > 0) {
        stage(clamp_chunk(0), 0);
        stage(clamp_chunk(1), 1);
        if (wq_u == 0) fetch_nh(clamp_chunk(0));       // one wave stages the norm entries
        if (wq_u == 0) nhbuf[tid & (kBR - 1)] = pre_ok ? pre_nh : kNegInf;
        if (wq_u == 0) fetch_nh(clamp_chunk(1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    TagTop1 best[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) best[qt].init();

    // fragment of reference tile `rt` (16 cells) of half-step `ri`, k-step `ks` of image `rimg`
    const int rd_perm = ((lane >> 3) & 1) | ((g & 1) << 1);   // (row bit 1) | (row bit 3) << 1 of this lane's rows
    auto frag = [&](const char *rimg, int ri, int rt, int ks) {
        const int slot = (ri * 2 + rt) ^ rd_perm;               // 32-byte slot = 16 cells
        return tr_frag(rimg + (ks * 32 + tr_row) * kRStride + slot * 32 + tr_col * 2, kRStride);
    };
    auto load_init = [&](const float *nh, int ri, int rt) {
        const float4 t = *reinterpret_cast<const float4 *>(nh + ri * 32 + rt * 16 + 4 * g);
        f32x4 v = {t.x, t.y, t.z, t.w};
        return v;
    };

    f32x4 acc_a[2][NQ], acc_b[2][NQ];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) {
            acc_b[rt][qt] = f32x4{kNegInf, kNegInf, kNegInf, kNegInf};
            asm volatile("" : "+v"(acc_b[rt][qt]));     // opaque: keeps the loop from being peeled
        }

    // MFMAs of half-step (`rimg`, `ri`) into `now`; top-list update of `prev` (half-step prev_ri of
    // chunk prev_chunk) interleaved one score per MFMA, four scores of one query tile at a time.  The
    // tag is the position within the CHUNK (prev_ri * 8 + tile * 4 + register), so which chunk gave
    // the best score is tracked once per chunk.  a[0..1] hold the two reference-tile fragments of
    // k-step 0 on entry and those of the next half-step on exit.
    auto half_step = [&](const char *rimg, int ri, const char *next_img, int next_ri, bf16x8 (&a)[2],
                         const f32x4 (&init)[2], f32x4 (&now)[2][NQ], const f32x4 (&prev)[2][NQ], int prev_chunk,
                         int prev_ri) {
        if (prev_ri == 0) {
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) best[qt].begin_half(prev_chunk);
        }
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            bf16x8 an[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (ISPA_ABLATE == 3 || ISPA_ABLATE == 13 || ISPA_ABLATE == 123) {                  // no LDS operand reads
                    an[rt] = a[rt];
                    asm volatile("" : "+v"(an[rt]));
                } else {
                    an[rt] = ks + 1 < KS32 ? frag(rimg, ri, rt, ks + 1) : frag(next_img, next_ri, rt, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) {
                    now[rt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], bq[qt][ks],
                                                                          ks ? now[rt][qt] : init[rt], 0, 0, 0);
                    // 8 NQ scores per half-step spread over its 2 NQ * KS32 MFMAs
                    constexpr int per = 4 / KS32 > 0 ? 4 / KS32 : 1;
                    const int m = (ks * 2 * NQ + rt * NQ + qt) * per;     // scores [m, m+per) of the previous half-step
#pragma unroll
                    for (int e = 0; e < per; ++e) {
                        const int sc = m + e;
                        if (sc < 8 * NQ) {
                            const int prt = sc / (4 * NQ), pqt = (sc >> 2) % NQ, reg = sc & 3;
                            if (ISPA_ABLATE == 1 || ISPA_ABLATE == 13 || ISPA_ABLATE == 123)
                                asm volatile("" ::"v"(prev[prt][pqt][reg]));
                            else
                                best[pqt].push4_step(reg, prev[prt][pqt], prev_ri * 8 + prt * 4);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one MFMA, then its share of the vector work
                }
            a[0] = an[0];
            a[1] = an[1];
        }
        if (KS32 == 3) {                                 // 24 MFMAs covered 24 scores: the rest here (NQ = 4)
#pragma unroll
            for (int sc = 24; sc < 32; sc += 4)
                best[(sc >> 2) & 3].push4(prev[sc >> 4][(sc >> 2) & 3], prev_ri * 8 + (sc >> 4) * 4);
        }
        if (prev_ri == 1) {
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) best[qt].end_half();
        }
    };

    bf16x8 a[2];
    f32x4 init_a[2];
    if (nchunks > 0) {
        a[0] = frag(rbuf, 0, 0, 0);
        a[1] = frag(rbuf, 0, 1, 0);
        init_a[0] = load_init(nhbuf, 0, 0);
        init_a[1] = load_init(nhbuf, 0, 1);
    }
    // One chunk, in ring slot `slot` (a compile-time constant: the loop below is unrolled over the three slots,
    // so every LDS address of the body is a loop-invariant lane register plus an immediate offset, and the
    // staging's m0 values are constants - no per-chunk address arithmetic).
    auto chunk_body = [&](int chunk, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        constexpr int next_slot = (slot + 1) % 3, after_slot = (slot + 2) % 3;
        const char *rimg = rbuf + slot * kRBuf;
        const char *next_img = rbuf + next_slot * kRBuf;
        const f32x4 init_b[2] = {load_init(nhbuf + slot * kBR, 1, 0), load_init(nhbuf + slot * kBR, 1, 1)};
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 0, rimg, 1, a, init_a, acc_a, acc_b, chunk - 1, 1);
        if (wq_u == 0) nhbuf[next_slot * kBR + (tid & (kBR - 1))] = pre_ok ? pre_nh : kNegInf;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ISPA_ABLATE != 2 && ISPA_ABLATE != 123) stage(clamp_chunk(chunk + 2), after_slot);
        if (wq_u == 0) fetch_nh(clamp_chunk(chunk + 2));
        init_a[0] = load_init(nhbuf + next_slot * kBR, 0, 0);
        init_a[1] = load_init(nhbuf + next_slot * kBR, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 1, next_img, 0, a, init_b, acc_b, acc_a, chunk, 0);
    };
    int chunk = 0;
    for (; chunk + 3 <= nchunks; chunk += 3) {
        chunk_body(chunk, std::integral_constant<int, 0>{});
        chunk_body(chunk + 1, std::integral_constant<int, 1>{});
        chunk_body(chunk + 2, std::integral_constant<int, 2>{});
    }
    if (chunk < nchunks) {
        chunk_body(chunk, std::integral_constant<int, 0>{});
        if (chunk + 1 < nchunks) chunk_body(chunk + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may still be landing in LDS at exit
    if (nchunks > 0) {                                   // second half of the last chunk
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) best[qt].push4(acc_b[rt][qt], 8 + rt * 4);
            best[qt].end_half();
        }
    }

    // one partial stream per (split, lane >> 4); finalize merges them
    const int nparts = nsplit * kT16SubStreams;
    const int part = split * kT16SubStreams + g;
    // code = chunk * 16 + (half-step * 8 + tile * 4 + accumulator register)
    auto to_ref = [&](int code) {
        const int tag = code & 15;
        return (int)(rbeg + (int64_t)(code >> 4) * kBR + (tag >> 2) * 16 + 4 * g + (tag & 3));
    };
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
        store_partial<2>(best[qt], to_ref, part_val, part_idx, b, nparts, part, nq, q0 + wq * 16 * NQ + qt * 16 + l15);
}

// ---------------------------------------------------------------------------------------------
// 2d. float32 inputs through a split-bfloat16 filter (top-1)
//
// x = hi + lo + e with hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-16 |x|.  The filter score is
// r_hi.q_hi + r_lo.q_hi + r_hi.q_lo - |r|^2/2 on the 16x16x32 bf16 MFMA: three MFMAs per tile and
// k-step at the bf16 rate instead of float32 MFMAs at 1/16 of it.  What is dropped (r_lo.q_lo and
// the e terms) is at most 3 * 2^-16 |q||r|, which goes into the margin test; queries inside the
// margin are re-scored in float64 on the ORIGINAL float32 inputs like everywhere else, so the
// results are the exact ones for the float32 inputs.
//
// nn_split_norms_kernel writes hi / lo (and the norms) in one pass over the float32 map.
// nn_filter_split_kernel: 256-thread block = 128 queries (wave w: 32 queries = 2 tiles of 16),
// 64-reference chunks whose hi and lo images sit in a 2-slot LDS ring filled by LDS-direct loads;
// the same 32-byte-slot permutation as in 2c keeps the transposed reads conflict-free.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t bf16_rne(float x) {
    const uint32_t u = __float_as_uint(x);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__global__ __launch_bounds__(256) void nn_split_norms_kernel(const float *__restrict__ x, int C, int64_t n,
                                                             uint16_t *__restrict__ hi, uint16_t *__restrict__ lo,
                                                             float *__restrict__ nh, double *__restrict__ sq64,
                                                             int *__restrict__ maxbits) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (j < n) {
        const int64_t base = (int64_t)b * C * n + j;
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[base + (int64_t)(c + u) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s = fma((double)v[u], (double)v[u], s);
                const uint16_t h = bf16_rne(v[u]);
                hi[base + (int64_t)(c + u) * n] = h;
                lo[base + (int64_t)(c + u) * n] = bf16_rne(v[u] - __uint_as_float((uint32_t)h << 16));
            }
        }
        for (; c < C; ++c) {
            const float v = x[base + (int64_t)c * n];
            s = fma((double)v, (double)v, s);
            const uint16_t h = bf16_rne(v);
            hi[base + (int64_t)c * n] = h;
            lo[base + (int64_t)c * n] = bf16_rne(v - __uint_as_float((uint32_t)h << 16));
        }
        nh[(int64_t)b * n + j] = (float)(-0.5 * s);
        sq64[(int64_t)b * n + j] = s;
    }
    if (maxbits) {
        float m = __double2float_ru(s);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_int(m));
    }
}

constexpr int kSQ = 128;             // queries per block of the split filter
constexpr int kSQStride = 256 + 64;  // bytes per LDS row of 128 bf16 query cells

template <int KS32>
constexpr int split_filter_lds_bytes() {
    constexpr int ring = 2 * 2 * 32 * KS32 * kRStride + 2 * kBR * (int)sizeof(float);   // 2 slots x (hi, lo)
    constexpr int qimage = 32 * KS32 * kSQStride;
    return ring > qimage ? ring : qimage;
}

template <int KS32>
__global__ __launch_bounds__(256, KS32 > 4 ? 1 : 2) void nn_filter_split_kernel(
    const uint16_t *__restrict__ qhi, const uint16_t *__restrict__ qlo, const uint16_t *__restrict__ rhi,
    const uint16_t *__restrict__ rlo, const float *__restrict__ rnh, int64_t nq, int64_t nr, int n_qtiles,
    int nsplit, int64_t refs_per_split, float *__restrict__ part_val, int *__restrict__ part_idx) {
    constexpr int C = 32 * KS32;
    constexpr int kImg = C * kRStride;                  // one image (hi or lo) of a chunk
    constexpr int kSlot = 2 * kImg;
    constexpr int kInstr = C / 8;                       // LDS-direct instructions per image
    constexpr int kPerWave = kInstr / 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *ring = lds;
    float *nhbuf = reinterpret_cast<float *>(lds + 2 * kSlot);

    const FilterWork fw = filter_work(n_qtiles, nsplit, refs_per_split, nr, kSQ, kBR);
    const int split = fw.split, b = fw.b;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wq = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tr_col = 4 * (lane & 3);
    const int tr_row = 8 * g + ((lane & 15) >> 2);

    const int64_t qoff = (int64_t)b * C * nq, roff = (int64_t)b * C * nr;
    const float *nhb = rnh + (int64_t)b * nr;
    const int64_t q0 = fw.q0, rbeg = fw.rbeg, rend = fw.rend;
    const int nchunks = fw.nchunks;

    // ---- query fragments: stage the hi image, read, then the lo image, read ----------------------
    bf16x8 bq[2][2][KS32];                              // [hi/lo][query tile][k-step]
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const uint16_t *src = (part ? qlo : qhi) + qoff;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2 * KS32; ++e) {            // C rows x 16 segments of 8 queries
            const int seg = tid + 256 * e;
            const int row = seg >> 4, col = (seg & 15) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q0 + col < nq) v = *reinterpret_cast<const uint4 *>(src + (int64_t)row * nq + q0 + col);
            *reinterpret_cast<uint4 *>(lds + row * kSQStride + col * 2) = v;
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
                bq[part][qt][ks] =
                    tr_frag(lds + (ks * 32 + tr_row) * kSQStride + (wq * 32 + qt * 16 + tr_col) * 2, kSQStride);
    }
    __syncthreads();

    // ---- reference chunks: LDS-direct loads, 8 rows x 128 B per instruction -----------------------
    const int wq_u = __builtin_amdgcn_readfirstlane(wq);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)ring;
    const unsigned row_off = (unsigned)((int64_t)(lane >> 3) * nr * 2);        // bytes; nr < 2^27 on this path
    const unsigned lane_off[2] = {row_off + perm_seg(lane, 0) * 16, row_off + perm_seg(lane, 1) * 16};
    auto stage = [&](int64_t r0, int slot) {
        // uniform: only a last chunk can be ragged.  Steady state: the chunk's column offset goes into the lane
        // offsets, the scalar bases of the pieces are loop invariants, m0 = slot base + piece offset
        if (__builtin_expect((int)(rend - r0) >= kBR, 1)) {
            const unsigned col2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)r0 * 2u));   // < 2^28
            const unsigned v0 = lane_off[0] + col2, v1 = lane_off[1] + col2;
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + slot * kSlot + wq_u * kPerWave * 1024));
            static_for<0, 2 * kPerWave>([&](auto ic) {
                constexpr int part = decltype(ic)::value / kPerWave, e = decltype(ic)::value % kPerWave;
                const int id = wq_u * kPerWave + e;                        // rows [8 id, 8 id + 8)
                uint64_t sbase = (uint64_t)((part ? rlo : rhi) + roff) + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sbase >> 32));
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sbase);
                sbase = ((uint64_t)hi << 32) | lo;
                glds_b128_lean<part * kImg + e * 1024>((id & 1) ? v1 : v0, sbase, sb);
            });
        } else {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const uint16_t *img = (part ? rlo : rhi) + roff;
#pragma unroll
                for (int e = 0; e < kPerWave; ++e) {
                    const int id = wq_u * kPerWave + e;
                    const int64_t col = r0 + perm_seg(lane, id & 1) * 8;
                    const unsigned voff = row_off + (col < rend ? (unsigned)col * 2 : 0);
                    const uint64_t sbase = (uint64_t)img + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                    glds_b128(voff, sbase, __builtin_amdgcn_readfirstlane(lds_base + slot * kSlot + part * kImg + id * 1024));
                }
            }
        }
    };
    float pre_nh = 0.f;
    bool pre_ok = false;
    const int nh_lane = tid & (kBR - 1);
    auto fetch_nh = [&](int64_t r0) {
        const int left = (int)(rend - r0);               // 32-bit lane arithmetic on a uniform base
        pre_ok = nh_lane < left;
        pre_nh = (nhb + r0)[pre_ok ? nh_lane : 0];
    };
    auto clamp_chunk = [&](int c) { return rbeg + (int64_t)(c < nchunks ? c : nchunks - 1) * kBR; };
    if (nchunks > 0) {
        stage(clamp_chunk(0), 0);
        stage(clamp_chunk(1), 1);
        if (wq_u == 0) fetch_nh(clamp_chunk(0));        // one wave stages the norm entries
        if (wq_u == 0) nhbuf[tid & (kBR - 1)] = pre_ok ? pre_nh : kNegInf;
        if (wq_u == 0) fetch_nh(clamp_chunk(1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    TagTop1 best[2];
    best[0].init();
    best[1].init();

    const int rd_perm = ((lane >> 3) & 1) | ((g & 1) << 1);
    auto frag = [&](const char *img, int ri, int rt, int ks) {
        const int slot32 = (ri * 2 + rt) ^ rd_perm;
        return tr_frag(img + (ks * 32 + tr_row) * kRStride + slot32 * 32 + tr_col * 2, kRStride);
    };
    auto load_init = [&](const float *nh, int ri, int rt) {
        const float4 t = *reinterpret_cast<const float4 *>(nh + ri * 32 + rt * 16 + 4 * g);
        f32x4 v = {t.x, t.y, t.z, t.w};
        return v;
    };

    f32x4 acc_a[2][2], acc_b[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            acc_b[rt][qt] = f32x4{kNegInf, kNegInf, kNegInf, kNegInf};
            asm volatile("" : "+v"(acc_b[rt][qt]));     // opaque: keeps the loop from being peeled
        }

    // MFMAs of half-step `ri` of the slot at `img` into `now` (3 per tile and k-step); the top-list
    // update of `prev` (half-step prev_half) is spread over them, one score per three MFMAs.
    auto half_step = [&](const char *img, int ri, const f32x4 (&init)[2], f32x4 (&now)[2][2],
                         const f32x4 (&prev)[2][2], int prev_chunk, int prev_ri) {
        if (prev_ri == 0) {                              // the best score's chunk is tracked once per chunk
            best[0].begin_half(prev_chunk);
            best[1].begin_half(prev_chunk);
        }
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            ah[rt] = frag(img, ri, rt, 0);
            al[rt] = frag(img + kImg, ri, rt, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            bf16x8 nh_[2], nl_[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                nh_[rt] = ks + 1 < KS32 ? frag(img, ri, rt, ks + 1) : ah[rt];
                nl_[rt] = ks + 1 < KS32 ? frag(img + kImg, ri, rt, ks + 1) : al[rt];
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 t = ks ? now[rt][qt] : init[rt];
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], bq[0][qt][ks], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], bq[0][qt][ks], t, 0, 0, 0);
                    now[rt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], bq[1][qt][ks], t, 0, 0, 0);
                    // 16 scores of the previous half-step over 4 * KS32 tile visits
                    constexpr int per = 4 / KS32 > 0 ? 4 / KS32 : 1;
                    const int m = (ks * 4 + rt * 2 + qt) * per;
#pragma unroll
                    for (int e = 0; e < per; ++e) {
                        const int sc = m + e;
                        if (sc < 16)                     // four scores of one query tile at a time
                            best[(sc >> 2) & 1].push4_step(sc & 3, prev[sc >> 3][(sc >> 2) & 1], prev_ri * 8 + (sc >> 3) * 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                ah[rt] = nh_[rt];
                al[rt] = nl_[rt];
            }
        }
        if (KS32 == 3) {
#pragma unroll
            for (int sc = 12; sc < 16; sc += 4)
                best[(sc >> 2) & 1].push4(prev[sc >> 3][(sc >> 2) & 1], prev_ri * 8 + (sc >> 3) * 4);
        }
        if (prev_ri == 1) {
            best[0].end_half();
            best[1].end_half();
        }
    };

    // one chunk in ring slot `slot`, a compile-time constant (the loop is unrolled over the two slots: LDS
    // addresses are loop-invariant lane registers plus immediates, the staging's m0 values constants)
    auto chunk_body = [&](int chunk, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const char *img = ring + slot * kSlot;
        const float *nh = nhbuf + slot * kBR;
        const f32x4 init_a[2] = {load_init(nh, 0, 0), load_init(nh, 0, 1)};
        const f32x4 init_b[2] = {load_init(nh, 1, 0), load_init(nh, 1, 1)};
        half_step(img, 0, init_a, acc_a, acc_b, chunk - 1, 1);
        half_step(img, 1, init_b, acc_b, acc_a, chunk, 0);
        // chunk end: the next chunk's images (requested a chunk ago) must have landed, and this slot
        // must be free of readers before it is refilled
        if (wq_u == 0) nhbuf[(slot ^ 1) * kBR + (tid & (kBR - 1))] = pre_ok ? pre_nh : kNegInf;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage(clamp_chunk(chunk + 2), slot);
        if (wq_u == 0) fetch_nh(clamp_chunk(chunk + 2));
    };
    int chunk = 0;
    for (; chunk + 2 <= nchunks; chunk += 2) {
        chunk_body(chunk, std::integral_constant<int, 0>{});
        chunk_body(chunk + 1, std::integral_constant<int, 1>{});
    }
    if (chunk < nchunks) chunk_body(chunk, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may still be landing in LDS at exit
    if (nchunks > 0) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {               // second half of the last chunk
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) best[qt].push4(acc_b[rt][qt], 8 + rt * 4);
            best[qt].end_half();
        }
    }

    const int nparts = nsplit * kT16SubStreams;
    const int part = split * kT16SubStreams + g;
    // code = chunk * 16 + (half-step * 8 + tile * 4 + accumulator register)
    auto to_ref = [&](int code) {
        const int tag = code & 15;
        return (int)(rbeg + (int64_t)(code >> 4) * kBR + (tag >> 2) * 16 + 4 * g + (tag & 3));
    };
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
        store_partial<2>(best[qt], to_ref, part_val, part_idx, b, nparts, part, nq, q0 + wq * 32 + qt * 16 + l15);
}

// ---------------------------------------------------------------------------------------------
// 3. finalize
// ---------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(256) void nn_finalize_kernel(const float *__restrict__ part_val,
                                                          const int *__restrict__ part_idx, int nparts,
                                                          const float *__restrict__ qnh,
                                                          const int *__restrict__ rmaxbits, int C, int64_t nq,
                                                          int64_t nr, int k_out, int index_base,
                                                          float err_rel, float tag_rel,
                                                          const T *__restrict__ q,
                                                          const T *__restrict__ r,
                                                          const double *__restrict__ qq64,
                                                          const double *__restrict__ rr64,
                                                          float *__restrict__ dist,
                                                          int *__restrict__ idx, int *__restrict__ amb_count,
                                                          int *__restrict__ amb_list,
                                                          int *__restrict__ pair_count,
                                                          int *__restrict__ pair_list,
                                                          int *__restrict__ light_count,
                                                          int *__restrict__ light_list) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    // merged list of the K+1 best (score, reference) pairs, descending; an entry is present iff its
    // score is > -inf.  A stream's last entry may carry no reference (code -1): only its score is
    // needed, to measure the margin below the k-th neighbour.
    float v[K + 1];
    int c[K + 1];
#pragma unroll
    for (int s = 0; s <= K; ++s) {
        v[s] = kNegInf;
        c[s] = -1;
    }
    // the lists of several streams are loaded together (a thread that waited for each entry before
    // asking for the next would spend ~1 us per entry: with many streams that is the kernel's duration)
    constexpr int PB = K == 1 ? 8 : (K <= 5 ? 2 : 1);
    for (int p0 = 0; p0 < nparts; p0 += PB) {
        float xs[PB][K + 1];
        int cs[PB][K + 1];
#pragma unroll
        for (int e = 0; e < PB; ++e)
#pragma unroll
            for (int s = 0; s <= K; ++s) {
                const int pp = p0 + e < nparts ? p0 + e : nparts - 1;
                const int64_t o = (((int64_t)b * nparts + pp) * (K + 1) + s) * nq + i;
                xs[e][s] = part_val[o];
                cs[e][s] = part_idx[o];
            }
#pragma unroll
        for (int e = 0; e < PB; ++e) {
            if (p0 + e >= nparts) break;
#pragma unroll
            for (int s = 0; s <= K; ++s) {
                const float x = xs[e][s];
                const unsigned code = (unsigned)cs[e][s];   // -1 sorts after every real index
                if (!(x > kAbsent) && x == x) break;        // absent (NaN falls through)
                if (x > v[K] || (x == v[K] && code < (unsigned)c[K]) || x != x) {
                    v[K] = x;
                    c[K] = (int)code;
#pragma unroll
                    for (int t = K; t > 0; --t) {
                        const bool sw = v[t] > v[t - 1] ||
                                        (v[t] == v[t - 1] && (unsigned)c[t] < (unsigned)c[t - 1]) || v[t] != v[t];
                        if (sw) {
                            const float tv = v[t];
                            v[t] = v[t - 1];
                            v[t - 1] = tv;
                            const int tc = c[t];
                            c[t] = c[t - 1];
                            c[t - 1] = tc;
                        }
                    }
                }
            }
        }
    }
    // two filter scores are ordered for certain when they differ by more than
    // err_rel * (|q| max|r| + max|r|^2 / 2); err_rel comes from the filter kernel's rounding model
    const float qq = -2.f * qnh[(int64_t)b * nq + i];
    const float rmax2 = __int_as_float(rmaxbits[b]);
    const float scale = sqrtf(qq) * sqrtf(rmax2) + 0.5f * rmax2;
    // ... plus what the position tag in the low mantissa bits hides: a share tag_rel of the score's own
    // magnitude, which for the scores that matter is well below `scale`.  delta_of(x) bounds the error of
    // the difference of a score x and any score in [x - delta_of(x), x] (and, the bound growing much
    // slower than the score falls, nothing further below x can overtake it either).
    auto delta_of = [&](float x) { return err_rel * scale + tag_rel * fabsf(x); };
    bool sure = true;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (s >= k_out) break;
        if (!(v[s] > kAbsent)) {                   // fewer references than k (or NaN: exact pass)
            if (v[s] != v[s]) sure = false;
            break;
        }
        if (c[s] < 0) sure = false;                // cannot happen for distinct scores; be safe
        if (v[s + 1] != v[s + 1]) sure = false;
        if (v[s + 1] > kAbsent && !(v[s] - v[s + 1] > delta_of(fmaxf(fabsf(v[s]), fabsf(v[s + 1]))))) sure = false;
    }
    if (sure) {
        // Distance of a kept neighbour.  The filter score s carries an error of at most delta / 2 (delta
        // bounds the difference of two scores), so qq - 2 s is within err = delta + 2^-22 qq of the exact
        // squared distance d (the second term: float32 roundings of qq and of the subtraction).  Where
        // that is provably under 1e-4 d, i.e. qq - 2 s >= 10002 err, it is returned; otherwise (close
        // matches: the cancellation in |q|^2 + |r|^2 - 2 q.r eats the digits) the entry is marked and
        // nn_refine_kernel recomputes the distance as the float64 pass computes it.
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (s >= k_out) break;
            const int64_t o = ((int64_t)b * k_out + s) * nq + i;
            const bool have = v[s] > kAbsent && c[s] >= 0;      // k_out <= K
            float d = __builtin_inff();
            int j = -1;
            if (have) {
                j = c[s];
                d = qq - 2.f * v[s];
                const float err = delta_of(v[s]) + 2.4e-7f * qq;
                if (!(d >= 10002.f * err)) d = __uint_as_float(kRefineMark);   // nn_refine_kernel recomputes it
            }
            idx[o] = have ? j + index_base : -1;
            dist[o] = d;
        }
        return;
    }
    if (K == 1 && v[0] > kAbsent) {
        // Top-1, margin too small.  Every reference that can still be the exact nearest one scores
        // at least v[0] - delta.  If in every stream only the stream's best can do that, the
        // contenders are known by index and re-scoring just those in float64 settles the query.  A
        // stream whose runner-up is also in range may hide contenders, and then the query goes to
        // the full float64 pass.
        const float lim = v[0] - delta_of(v[0]);
        int cand[8];
        int nc = 0;
        bool hidden = false;
#pragma unroll 8
        for (int p = 0; p < nparts; ++p) {
            const int64_t o = (((int64_t)b * nparts + p) * 2) * nq + i;
            const float a_s = part_val[o], b_s = part_val[o + nq];
            if (a_s >= lim) {
                if (nc < 8) cand[nc++] = part_idx[o];
                else hidden = true;
            }
            if (b_s > kAbsent && b_s >= lim) hidden = true;
        }
        for (int e = 0; e < nc; ++e) hidden |= cand[e] < 0;
        if (!hidden && nc >= 1) {
            // known contenders: nn_rescore_top1_kernel settles the query, one wave per query (doing it
            // here would make this one thread's dependent loads the duration of the whole kernel)
            const int slot = atomicAdd(light_count, 1);
            light_list[2 * slot] = (int)((int64_t)b * nq + i);
            light_list[2 * slot + 1] = __float_as_int(lim);
            return;
        }
    }
    if (K > 1 && k_out <= K && v[k_out - 1] > kAbsent) {
        // Top-k, some margin too small.  Every reference that can be among the exact k nearest scores
        // at least lim = v[k-1] - delta.  A stream can hide an unkept reference above lim only if its
        // own last kept score reaches lim.  If no stream does, the contenders are the kept entries
        // >= lim of all streams, known by index: they are re-scored in float64 (the oracle's channel
        // order) and the k smallest (distance, index) kept.
        const float lim = v[k_out - 1] - delta_of(v[k_out - 1]);
        bool hidden = false;
#pragma unroll
        for (int s = 0; s < K; ++s)
            if (v[s] != v[s]) hidden = true;
        for (int p = 0; p < nparts && !hidden; ++p) {
            const float last = part_val[(((int64_t)b * nparts + p) * (K + 1) + K) * nq + i];
            if (last != last || (last > kAbsent && last >= lim)) hidden = true;
        }
        if (!hidden) {
            // known contenders: nn_rescore_topk_kernel settles the query, one wave per query (doing
            // it here would make this one thread's dependent loads the duration of the whole kernel)
            const int slot = atomicAdd(light_count, 1);
            light_list[2 * slot] = (int)((int64_t)b * nq + i);
            light_list[2 * slot + 1] = __float_as_int(lim);
            return;
        }
    }
    // two queues: a global one (balanced one-query-per-block pass) and one per pair (the batched
    // pass lets the queries of one pair share one sweep over the pair's reference map)
    const int slot = atomicAdd(amb_count, 1);
    amb_list[slot] = (int)((int64_t)b * nq + i);   // b*nq+i < 2^31 is checked on the host
    const int pslot = atomicAdd(&pair_count[b], 1);
    pair_list[(int64_t)b * nq + pslot] = (int)i;
}

// ---------------------------------------------------------------------------------------------
// 3b. top-1 queries whose contenders are known: one wave per query.  Lane p looks at stream p; if that
// stream's best score reaches the query's limit, the lane re-scores that reference in float64 (the
// oracle's channel order, 16 channels of loads in flight at a time); the wave keeps the smallest
// (distance, index).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nn_rescore_top1_kernel(const float *__restrict__ part_val,
                                                              const int *__restrict__ part_idx, int nparts,
                                                              const T *__restrict__ q, const T *__restrict__ r,
                                                              const double *__restrict__ qq64,
                                                              const double *__restrict__ rr64, int C, int64_t nq,
                                                              int64_t nr, int index_base,
                                                              const int *__restrict__ light_count,
                                                              const int *__restrict__ light_list,
                                                              float *__restrict__ dist, int *__restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const int count = *light_count;
    for (int w = wave; w < count; w += nwaves) {
        const int64_t flat = light_list[2 * w];
        const float lim = __int_as_float(light_list[2 * w + 1]);
        const int b = (int)(flat / nq);
        const int64_t i = flat % nq;
        const T *qcol = q + (int64_t)b * C * nq + i;
        double best_d = __builtin_inf();
        int best_j = 0x7fffffff;
        for (int p0 = 0; p0 < nparts; p0 += 64) {
            const int p = p0 + lane;
            int j = -1;
            if (p < nparts) {
                const int64_t o = (((int64_t)b * nparts + p) * 2) * nq + i;
                if (part_val[o] >= lim) j = part_idx[o];
            }
            if (j < 0) continue;
            const double d = exact_d2(qcol, r + (int64_t)b * C * nr + j, C, nq, nr, qq64[flat], rr64[(int64_t)b * nr + j]);
            if (d < best_d || (d == best_d && j < best_j)) {
                best_d = d;
                best_j = j;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_xor(best_d, off);
            const int oj = __shfl_xor(best_j, off);
            if (od < best_d || (od == best_d && oj < best_j)) {
                best_d = od;
                best_j = oj;
            }
        }
        if (lane == 0) {                                 // k_out == 1
            idx[flat] = best_j + index_base;
            dist[flat] = (float)best_d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3c. top-k queries whose contenders are known: one wave per query.  Lane t looks at kept entry
// t % K of stream t / K; entries that reach the query's limit are re-scored in float64 as above, and
// the wave extracts the k smallest (distance, index) with k rounds of a wave-wide minimum.
// ---------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(256) void nn_rescore_topk_kernel(const float *__restrict__ part_val,
                                                              const int *__restrict__ part_idx, int nparts,
                                                              const T *__restrict__ q, const T *__restrict__ r,
                                                              const double *__restrict__ qq64,
                                                              const double *__restrict__ rr64, int C, int64_t nq,
                                                              int64_t nr, int k_out, int index_base,
                                                              const int *__restrict__ light_count,
                                                              const int *__restrict__ light_list,
                                                              float *__restrict__ dist, int *__restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const int count = *light_count;
    const int total = nparts * K;
    for (int w = wave; w < count; w += nwaves) {
        const int64_t flat = light_list[2 * w];
        const float lim = __int_as_float(light_list[2 * w + 1]);
        const int b = (int)(flat / nq);
        const int64_t i = flat % nq;
        const T *qcol = q + (int64_t)b * C * nq + i;
        double cd[K];                                    // the wave's k smallest so far, same in every lane
        int cj[K];
#pragma unroll
        for (int e = 0; e < K; ++e) {
            cd[e] = __builtin_inf();
            cj[e] = 0x7fffffff;
        }
        for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            int j = -1;
            if (t < total) {
                const int64_t o = (((int64_t)b * nparts + t / K) * (K + 1) + t % K) * nq + i;
                const float x = part_val[o];
                if (x > kAbsent && x >= lim) j = part_idx[o];
            }
            double d = __builtin_inf();
            int dj = 0x7fffffff;
            if (j >= 0) {
                d = exact_d2(qcol, r + (int64_t)b * C * nr + j, C, nq, nr, qq64[flat], rr64[(int64_t)b * nr + j]);
                dj = j;
            }
            const int n_valid = __popcll(__builtin_amdgcn_ballot_w64(j >= 0));
            const int rounds = n_valid < k_out ? n_valid : k_out;
            for (int e = 0; e < rounds; ++e) {
                double md = d;
                int mj = dj;
                for (int off = 32; off > 0; off >>= 1) {
                    const double od = __shfl_xor(md, off);
                    const int oj = __shfl_xor(mj, off);
                    if (od < md || (od == md && oj < mj)) {
                        md = od;
                        mj = oj;
                    }
                }
                if (dj == mj) {                          // the owner retires it (a reference sits in one stream only)
                    d = __builtin_inf();
                    dj = 0x7fffffff;
                }
                // insert into the ascending (distance, index) list; what falls off the end is dropped
                double xd = md;
                int xj = mj;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const bool lt = xd < cd[s] || (xd == cd[s] && xj < cj[s]);
                    const double td = lt ? cd[s] : xd;
                    const int tj = lt ? cj[s] : xj;
                    cd[s] = lt ? xd : cd[s];
                    cj[s] = lt ? xj : cj[s];
                    xd = td;
                    xj = tj;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < K; ++s)
                if (s < k_out) {
                    const int64_t o = ((int64_t)b * k_out + s) * nq + i;
                    idx[o] = cj[s] + index_base;
                    dist[o] = (float)cd[s];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3d. distances of close matches.  nn_finalize_kernel marks the entries whose filter score cannot give
// the distance to 1e-4 relative; here they are recomputed in float64 (exact_d2).  One 512-thread block
// per (slice of kRefineEntries result entries, pair); entry e = s * nq + i is neighbour s of query i.
//   few marked entries in the slice: each is re-scored straight from global memory (C scattered loads);
//   many (good matches everywhere, the realistic case for corresponding images): the pair's reference map
//   passes through LDS a group of channels at a time, every thread keeps float64 dot products for its
//   entries and gathers the reference values from LDS: the map is read once per slice instead of one
//   64-byte sector per (entry, channel).
// Channel order is increasing either way, so both routes return the float64 pass's bits.
// ---------------------------------------------------------------------------------------------
// Two workgroups share a CU (64 KB of LDS each), so one's staging overlaps the other's gathers.
constexpr int kRefineThreads = 512;
constexpr int kRefinePer = 8;                                   // entries per thread
constexpr int kRefineEntries = kRefineThreads * kRefinePer;
constexpr int kRefineLds = 64 * 1024;

template <typename T>
__global__ __launch_bounds__(kRefineThreads) void nn_refine_kernel(const T *__restrict__ q, const T *__restrict__ r,
                                                                   const double *__restrict__ qq64,
                                                                   const double *__restrict__ rr64, int C, int64_t nq,
                                                                   int64_t nr, int k_out, int index_base, int vec_ok,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ dist) {
    extern __shared__ __align__(16) unsigned char refine_lds[];
    __shared__ int n_marked;
    const int b = blockIdx.y;
    const int64_t total = (int64_t)k_out * nq, e0 = (int64_t)blockIdx.x * kRefineEntries;
    const T *qb = q + (int64_t)b * C * nq, *rb = r + (int64_t)b * C * nr;
    if (threadIdx.x == 0) n_marked = 0;
    __syncthreads();
    int64_t qi[kRefinePer];
    int rj[kRefinePer];
    int mine = 0;
#pragma unroll
    for (int u = 0; u < kRefinePer; ++u) {
        const int64_t e = e0 + threadIdx.x + (int64_t)u * kRefineThreads;
        rj[u] = -1;
        qi[u] = 0;
        if (e < total) {
            const int64_t o = (int64_t)b * total + e;
            if (__float_as_uint(dist[o]) == kRefineMark) {
                const int j = idx[o] - index_base;
                if (j >= 0 && j < nr) {                         // always true for an entry finalize marked
                    rj[u] = j;
                    qi[u] = e % nq;
                    ++mine;
                }
            }
        }
    }
    if (mine) atomicAdd(&n_marked, mine);
    __syncthreads();
    const int marked = n_marked;
    if (marked == 0) return;
    // LDS route: worth it once the scattered loads (a 64-byte sector each) would move more than the map
    const int64_t row = (nr + 7) / 8 * 8;                       // elements per staged channel row
    const int group = (int)(kRefineLds / (row * (int64_t)sizeof(T)) < C ? kRefineLds / (row * (int64_t)sizeof(T)) : C);
    const bool sweep = group >= 1 && (int64_t)marked * 64 > nr * (int64_t)sizeof(T);
    double dot[kRefinePer];
#pragma unroll
    for (int u = 0; u < kRefinePer; ++u) dot[u] = 0.0;
    if (!sweep) {
#pragma unroll
        for (int u = 0; u < kRefinePer; ++u)
            if (rj[u] >= 0) {
                const int64_t o = (int64_t)b * total + e0 + threadIdx.x + (int64_t)u * kRefineThreads;
                dist[o] = (float)exact_d2(qb + qi[u], rb + rj[u], C, nq, nr, qq64[(int64_t)b * nq + qi[u]],
                                          rr64[(int64_t)b * nr + rj[u]]);
            }
        return;
    }
    T *tile = reinterpret_cast<T *>(refine_lds);
    constexpr int kVec = 16 / (int)sizeof(T);
    int rc[kRefinePer];
#pragma unroll
    for (int u = 0; u < kRefinePer; ++u) rc[u] = rj[u] < 0 ? 0 : rj[u];
    for (int c0 = 0; c0 < C; c0 += group) {
        const int nc = C - c0 < group ? C - c0 : group;
        __syncthreads();                                        // the previous group has been consumed
        if (vec_ok) {                                           // rows of nr elements, 16-byte aligned
            const int64_t vrow = nr / kVec;
            for (int cc = 0; cc < nc; ++cc)
                for (int64_t x = threadIdx.x; x < vrow; x += kRefineThreads)
                    *reinterpret_cast<uint4 *>(tile + cc * row + x * kVec) =
                        *reinterpret_cast<const uint4 *>(rb + (int64_t)(c0 + cc) * nr + x * kVec);
        } else {
            for (int cc = 0; cc < nc; ++cc)
                for (int64_t x = threadIdx.x; x < nr; x += kRefineThreads)
                    tile[cc * row + x] = rb[(int64_t)(c0 + cc) * nr + x];
        }
        __syncthreads();
        // Unmarked entries gather cell 0 (and query 0) and are dropped at the end: without branches in
        // the loop two channels' loads are in flight before the first product.
        auto channels = [&](int cc, auto count) {
            constexpr int kN = decltype(count)::value;
            float qv[kN][kRefinePer], rv[kN][kRefinePer];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < kN; ++e)
#pragma unroll
                for (int u = 0; u < kRefinePer; ++u) {
                    qv[e][u] = load_as_float(qb + (int64_t)(c0 + cc + e) * nq + qi[u]);
                    rv[e][u] = load_as_float(tile + (cc + e) * row + rc[u]);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < kN; ++e)
#pragma unroll
                for (int u = 0; u < kRefinePer; ++u) dot[u] = fma((double)qv[e][u], (double)rv[e][u], dot[u]);
        };
        int cc = 0;
        for (; cc + 2 <= nc; cc += 2) channels(cc, std::integral_constant<int, 2>{});
        for (; cc < nc; ++cc) channels(cc, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int u = 0; u < kRefinePer; ++u)
        if (rj[u] >= 0) {
            const int64_t o = (int64_t)b * total + e0 + threadIdx.x + (int64_t)u * kRefineThreads;
            dist[o] = (float)((qq64[(int64_t)b * nq + qi[u]] + rr64[(int64_t)b * nr + rj[u]]) - 2.0 * dot[u]);
        }
}

// ---------------------------------------------------------------------------------------------
// 3e. small problems, top-1: finalize, re-scoring of known contenders and distance refinement in one
// launch.  nn_finalize_kernel gives a query to one thread, which walks all partial streams (two passes of
// dependent loads when a margin is small); with a single image pair that is 4096 threads on the whole chip
// and the kernel lasts as long as the filter pass.  Here a 1024-thread workgroup owns 64 queries: wave w
// holds the lists of streams w, w + 16, ... in registers (at most kFsParts each), the 16 waves merge through
// LDS, wave 0 decides like nn_finalize_kernel (same comparisons, same margins), and queries inside the margin
// get their contenders from the registers the waves still hold; known contenders are re-scored by 16
// threads per query (exact_d2, as nn_rescore_top1_kernel does), hidden ones go to the float64 pass's queue.
// Marked distances (close matches) are recomputed on the spot (as nn_refine_kernel does).  The largest
// reference norm of the pair is reduced here from the float64 norms, so nothing needs a zeroed cell
// before the norms pass.
// ---------------------------------------------------------------------------------------------
constexpr int kFsWaves = 16, kFsParts = 8, kFsCand = 8;
constexpr int64_t kSmallQueries = 16384, kSmallRefs = 16384;

__device__ __forceinline__ void top2_insert(float x, int code, float &v0, int &c0, float &v1, int &c1) {
    if (!(x > kAbsent) && x == x) return;                      // absent (NaN falls through)
    if (x > v1 || (x == v1 && (unsigned)code < (unsigned)c1) || x != x) {
        v1 = x;
        c1 = code;
        if (v1 > v0 || (v1 == v0 && (unsigned)c1 < (unsigned)c0) || v1 != v1) {
            const float tv = v1;
            v1 = v0;
            v0 = tv;
            const int tc = c1;
            c1 = c0;
            c0 = tc;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64 * kFsWaves) void nn_finalize_small_kernel(
    const float *__restrict__ part_val, const int *__restrict__ part_idx, int nparts, const float *__restrict__ qnh,
    int C, int64_t nq, int64_t nr, int index_base, float err_rel, float tag_rel, const T *__restrict__ q,
    const T *__restrict__ r,
    const double *__restrict__ qq64, const double *__restrict__ rr64, float *__restrict__ dist,
    int *__restrict__ idx, int *__restrict__ amb_count, int *__restrict__ amb_list, int *__restrict__ pair_count,
    int *__restrict__ pair_list, int *__restrict__ light_count) {
    __shared__ float s_v[2][kFsWaves][64];
    __shared__ int s_c[2][kFsWaves][64];
    __shared__ float s_rmax[kFsWaves];
    __shared__ float s_lim[64];
    __shared__ int s_state[64];                  // 0 settled, 1 inside the margin, 2 contenders known, 3 queued
    __shared__ int s_ncand[64], s_hidden[64];
    __shared__ int s_cand[64][kFsCand];
    __shared__ double s_cd[64][kFsCand];
    __shared__ int s_any;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool live = i < nq;
    const int64_t ii = live ? i : nq - 1;
    if (tid < 64) {
        s_state[tid] = 0;
        s_ncand[tid] = 0;
        s_hidden[tid] = 0;
    }
    if (tid == 0) s_any = 0;
    // this wave's streams, all loads in flight together
    float xs[kFsParts][2];
    int cs[kFsParts][2];
#pragma unroll
    for (int e = 0; e < kFsParts; ++e) {
        const int p = wave + kFsWaves * e;
        const int pp = p < nparts ? p : nparts - 1;
        const int64_t o = (((int64_t)b * nparts + pp) * 2) * nq + ii;
        xs[e][0] = part_val[o];
        xs[e][1] = part_val[o + nq];
        cs[e][0] = part_idx[o];
        cs[e][1] = part_idx[o + nq];
    }
    // largest squared reference norm of the pair, rounded up (what the norms pass gives the large-batch path)
    double m = 0.0;
    for (int64_t j = tid; j < nr; j += 64 * kFsWaves) m = fmax(m, rr64[(int64_t)b * nr + j]);
    float mf = __double2float_ru(m);
    for (int off = 32; off > 0; off >>= 1) mf = fmaxf(mf, __shfl_xor(mf, off));
    if (lane == 0) s_rmax[wave] = mf;
    float v0 = kNegInf, v1 = kNegInf;
    int c0 = -1, c1 = -1;
#pragma unroll
    for (int e = 0; e < kFsParts; ++e) {
        if (wave + kFsWaves * e >= nparts) break;
        top2_insert(xs[e][0], cs[e][0], v0, c0, v1, c1);
        top2_insert(xs[e][1], cs[e][1], v0, c0, v1, c1);
    }
    s_v[0][wave][lane] = v0;
    s_v[1][wave][lane] = v1;
    s_c[0][wave][lane] = c0;
    s_c[1][wave][lane] = c1;
    __syncthreads();
    const int64_t flat = (int64_t)b * nq + i;
    double qq_exact = 0.0;
    if (wave == 0 && live) {
        for (int w = 1; w < kFsWaves; ++w) {
            top2_insert(s_v[0][w][lane], s_c[0][w][lane], v0, c0, v1, c1);
            top2_insert(s_v[1][w][lane], s_c[1][w][lane], v0, c0, v1, c1);
        }
        float rmax2 = s_rmax[0];
        for (int w = 1; w < kFsWaves; ++w) rmax2 = fmaxf(rmax2, s_rmax[w]);
        // margins and decisions of nn_finalize_kernel, K = k_out = 1
        const float qq = -2.f * qnh[flat];
        const float scale = sqrtf(qq) * sqrtf(rmax2) + 0.5f * rmax2;
        auto delta_of = [&](float x) { return err_rel * scale + tag_rel * fabsf(x); };
        bool sure = true;
        if (!(v0 > kAbsent)) {
            if (v0 != v0) sure = false;
        } else {
            if (c0 < 0) sure = false;
            if (v1 != v1) sure = false;
            if (v1 > kAbsent && !(v0 - v1 > delta_of(fmaxf(fabsf(v0), fabsf(v1))))) sure = false;
        }
        qq_exact = qq64[flat];
        if (sure) {
            const float err = delta_of(v0) + 2.4e-7f * qq;
            const bool have = v0 > kAbsent && c0 >= 0;
            float d = __builtin_inff();
            if (have) {
                d = qq - 2.f * v0;
                if (!(d >= 10002.f * err))              // a close match: the filter score cannot give 1e-4 relative
                    d = (float)exact_d2(q + (int64_t)b * C * nq + i, r + (int64_t)b * C * nr + c0, C, nq, nr, qq_exact,
                                        rr64[(int64_t)b * nr + c0]);
            }
            idx[flat] = have ? c0 + index_base : -1;
            dist[flat] = d;
        } else if (v0 > kAbsent) {
            s_lim[lane] = v0 - delta_of(v0);
            s_state[lane] = 1;
            s_any = 1;
        } else {
            s_state[lane] = 3;
            s_any = 1;
        }
    }
    __syncthreads();
    if (!s_any) return;                                         // uniform over the workgroup
    // contenders of the queries inside the margin, from the registers: every stream's best that reaches the
    // limit; a stream whose runner-up reaches it too may hide more
    if (s_state[lane] == 1) {
        const float lim = s_lim[lane];
#pragma unroll
        for (int e = 0; e < kFsParts; ++e) {
            if (wave + kFsWaves * e >= nparts) break;
            if (xs[e][0] >= lim) {
                const int slot = atomicAdd(&s_ncand[lane], 1);
                if (slot < kFsCand) s_cand[lane][slot] = cs[e][0];
                else s_hidden[lane] = 1;
            }
            if (xs[e][1] > kAbsent && xs[e][1] >= lim) s_hidden[lane] = 1;
        }
    }
    __syncthreads();
    if (wave == 0 && s_state[lane] == 1) {
        const int nc = s_ncand[lane] < kFsCand ? s_ncand[lane] : kFsCand;
        bool hidden = s_hidden[lane] != 0 || nc < 1;
        for (int e = 0; e < nc; ++e) hidden |= s_cand[lane][e] < 0;
        s_state[lane] = hidden ? 3 : 2;
    }
    __syncthreads();
    {
        const int ql = tid >> 4, slot = tid & 15;               // 16 threads per query, one contender each
        const int64_t qi = (int64_t)blockIdx.x * 64 + ql;
        if (s_state[ql] == 2 && slot < kFsCand && slot < s_ncand[ql]) {
            const int j = s_cand[ql][slot];
            s_cd[ql][slot] = exact_d2(q + (int64_t)b * C * nq + qi, r + (int64_t)b * C * nr + j, C, nq, nr,
                                      qq64[(int64_t)b * nq + qi], rr64[(int64_t)b * nr + j]);
        }
    }
    __syncthreads();
    if (wave == 0 && live) {
        if (s_state[lane] == 2) {
            double best_d = __builtin_inf();
            int best_j = 0x7fffffff;
            for (int e = 0; e < s_ncand[lane]; ++e) {
                const double d = s_cd[lane][e];
                const int j = s_cand[lane][e];
                if (d < best_d || (d == best_d && j < best_j)) {
                    best_d = d;
                    best_j = j;
                }
            }
            idx[flat] = best_j + index_base;
            dist[flat] = (float)best_d;
            atomicAdd(light_count, 1);                          // statistics only
        } else if (s_state[lane] == 3) {
            const int slot = atomicAdd(amb_count, 1);
            amb_list[slot] = (int)flat;
            const int pslot = atomicAdd(&pair_count[b], 1);
            pair_list[(int64_t)b * nq + pslot] = (int)i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 4. exact float64 pass for the queued queries
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool key_less(double d0, int j0, double d1, int j1) {
    return d0 < d1 || (d0 == d1 && j0 < j1);
}

// One block (NT threads: 1024 for long reference maps, where a single queued query would otherwise set the
// duration of the launch) per queued query.  d2(j) = (qq + rr_j) - 2 dot_j with dot accumulated over channels in
// increasing order in float64 (every product of two float32 values is exact in float64); qq and
// rr come from the norms kernel, accumulated the same way.  Four references per thread are in
// flight at a time so that the dependent fma chains overlap.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void nn_exact_kernel(const T *__restrict__ q, const T *__restrict__ r,
                                                       const double *__restrict__ qq64,
                                                       const double *__restrict__ rr64, int C, int64_t nq,
                                                       int64_t nr, int k_out, int index_base, int vec_ok,
                                                       const int *__restrict__ amb_count,
                                                       const int *__restrict__ amb_list, int skip_from,
                                                       int n_pair, const int *__restrict__ pair_count,
                                                       float *__restrict__ dist, int *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *qs = reinterpret_cast<double *>(smem);                 // C doubles
    double *sd = qs + C;                                           // NT doubles
    int *sj = reinterpret_cast<int *>(sd + NT);                    // NT ints
    int *sown = sj + NT;                                           // NT ints
    const int count = *amb_count;
    // pairs with long queues belong to nn_exact_top1_kernel, but only when queues are long on
    // average: sharing reference reads pays off in bandwidth, not in latency
    const bool some_batched = count >= (int64_t)skip_from * n_pair;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int64_t flat = amb_list[w];
        const int b = (int)(flat / nq);
        const int64_t i = flat % nq;
        if (some_batched && pair_count[b] >= skip_from) continue;   // block-uniform
        const T *qcol = q + (int64_t)b * C * nq + i;
        const T *rb = r + (int64_t)b * C * nr;
        const double *rrb = rr64 + (int64_t)b * nr;
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += NT) qs[c] = (double)load_as_float(qcol + (int64_t)c * nq);
        __syncthreads();
        const double qq = qq64[flat];
        double bd[kMaxK];
        int bj[kMaxK];
#pragma unroll
        for (int s = 0; s < kMaxK; ++s) {
            bd[s] = __builtin_inf();
            bj[s] = 0x7fffffff;
        }
        auto consider = [&](double dot, int j) {
            const double d = (qq + rrb[j]) - 2.0 * dot;
            // insert into the thread-local ascending list (static indexing keeps it in registers)
            double cd = d;
            int cj = j;
#pragma unroll
            for (int s = 0; s < kMaxK; ++s) {
                const bool lt = key_less(cd, cj, bd[s], bj[s]);
                const double td = lt ? bd[s] : cd;
                const int tj = lt ? bj[s] : cj;
                bd[s] = lt ? cd : bd[s];
                bj[s] = lt ? cj : bj[s];
                cd = td;
                cj = tj;
            }
        };
        constexpr int V = 16 / (int)sizeof(T);          // cells per 16-byte load: 4 float, 8 bfloat16
        if (vec_ok) {
            // V consecutive references per thread and load; every chain keeps the channel order
            for (int64_t j0 = (int64_t)threadIdx.x * V; j0 < nr; j0 += NT * V) {
                double dot[V];
#pragma unroll
                for (int u = 0; u < V; ++u) dot[u] = 0.0;
                const T *rp = rb + j0;
                // kDepth channel rows are requested before the first of them is used (the scheduling barriers pin
                // that order: left alone, the compiler issues one load, waits for it, uses it, and a thread's sweep
                // becomes a chain of C memory latencies, about 1400 cycles each from the Infinity Cache / HBM)
                constexpr int kDepth = 8;
                auto use = [&](const uint4 &raw, double qv) {
                    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int u = 0; u < V; ++u) {
                        float rv;
                        if (sizeof(T) == 4)
                            rv = __uint_as_float(w[u & 3]);
                        else
                            rv = __uint_as_float((u & 1) ? (w[u >> 1] & 0xffff0000u) : (w[u >> 1] << 16));
                        dot[u] = fma(qv, (double)rv, dot[u]);
                    }
                };
                int c0 = 0;
                for (; c0 + kDepth <= C; c0 += kDepth) {
                    uint4 raw[kDepth];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < kDepth; ++e)
                        raw[e] = *reinterpret_cast<const uint4 *>(rp + (int64_t)(c0 + e) * nr);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < kDepth; ++e) use(raw[e], qs[c0 + e]);
                }
                for (; c0 < C; ++c0) use(*reinterpret_cast<const uint4 *>(rp + (int64_t)c0 * nr), qs[c0]);
#pragma unroll
                for (int u = 0; u < V; ++u) consider(dot[u], (int)(j0 + u));
            }
        } else {
            for (int64_t j0 = threadIdx.x; j0 < nr; j0 += 4 * NT) {
                double dot[4] = {0.0, 0.0, 0.0, 0.0};
                const T *rp[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    in[u] = j0 + NT * u < nr;
                    rp[u] = rb + (in[u] ? j0 + NT * u : j0);
                }
                for (int c = 0; c < C; ++c) {
                    const double qv = qs[c];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        dot[u] = fma(qv, (double)load_as_float(rp[u] + (int64_t)c * nr), dot[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (in[u]) consider(dot[u], (int)(j0 + NT * u));
            }
        }
        // k_out rounds of block-wide lexicographic minimum over the thread-local heads
        for (int s = 0; s < k_out; ++s) {
            __syncthreads();
            sd[threadIdx.x] = bd[0];
            sj[threadIdx.x] = bj[0];
            sown[threadIdx.x] = threadIdx.x;
            __syncthreads();
            for (int off = NT / 2; off > 0; off >>= 1) {
                if (threadIdx.x < off) {
                    const int o = threadIdx.x + off;
                    if (key_less(sd[o], sj[o], sd[threadIdx.x], sj[threadIdx.x])) {
                        sd[threadIdx.x] = sd[o];
                        sj[threadIdx.x] = sj[o];
                        sown[threadIdx.x] = sown[o];
                    }
                }
                __syncthreads();
            }
            if ((int)threadIdx.x == sown[0]) {      // pop the winner from its owner's list
#pragma unroll
                for (int t = 0; t + 1 < kMaxK; ++t) {
                    bd[t] = bd[t + 1];
                    bj[t] = bj[t + 1];
                }
                bd[kMaxK - 1] = __builtin_inf();
                bj[kMaxK - 1] = 0x7fffffff;
            }
            if (threadIdx.x == 0) {
                const int64_t o = ((int64_t)b * k_out + s) * nq + i;
                const bool have = sj[0] != 0x7fffffff;
                idx[o] = have ? sj[0] + index_base : -1;
                dist[o] = have ? (float)sd[0] : __builtin_inff();
            }
        }
    }
}

// Top-1 variant that shares the sweep over a pair's reference map among up to kQB queued queries
// of that pair (the pass is bound by re-reading the map, not by arithmetic).  Same float64 chain per
// (query, reference) as above: channels in increasing order, d2 = (qq + rr) - 2 dot, ties to the
// lowest index.  Needs nr % 4 == 0 and 16-byte aligned rows (the generic kernel covers the rest).
constexpr int kQB = 8;
constexpr int kBatchMin = 4;   // shorter queues are cheaper one query per block (nn_exact_kernel)

template <typename T>
__global__ __launch_bounds__(256) void nn_exact_top1_kernel(const T *__restrict__ q, const T *__restrict__ r,
                                                            const double *__restrict__ qq64,
                                                            const double *__restrict__ rr64, int C, int64_t nq,
                                                            int64_t nr, int index_base,
                                                            const int *__restrict__ total_count,
                                                            const int *__restrict__ pair_count,
                                                            const int *__restrict__ pair_list,
                                                            float *__restrict__ dist, int *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *qs = reinterpret_cast<double *>(smem);                 // [kQB][C]
    double *sd = qs + kQB * C;                                     // 256
    int *sj = reinterpret_cast<int *>(sd + 256);                   // 256
    // The pair index runs fastest over the grid: workgroups go to the eight XCDs in turn by linear index, and
    // with the batch slot fastest the busy ones (a pair's first slots) would land on every second XCD only.
    const int b = blockIdx.x;
    const int count = pair_count[b];
    if (count < kBatchMin || *total_count < kBatchMin * (int)gridDim.x) return;
    const T *rb = r + (int64_t)b * C * nr;
    const double *rrb = rr64 + (int64_t)b * nr;
    for (int base = blockIdx.y * kQB; base < count; base += gridDim.y * kQB) {
        const int nqry = min(kQB, count - base);
        __syncthreads();
        for (int t = threadIdx.x; t < kQB * C; t += 256) {
            const int qi = t / C, c = t % C;
            const int64_t i = pair_list[(int64_t)b * nq + base + (qi < nqry ? qi : 0)];
            qs[t] = (double)load_as_float(q + ((int64_t)b * C + c) * nq + i);
        }
        __syncthreads();
        double bd[kQB];
        int bj[kQB];
#pragma unroll
        for (int qi = 0; qi < kQB; ++qi) {
            bd[qi] = __builtin_inf();
            bj[qi] = 0x7fffffff;
        }
        for (int64_t j0 = (int64_t)threadIdx.x * 4; j0 < nr; j0 += 1024) {
            double dot[kQB][4];
#pragma unroll
            for (int qi = 0; qi < kQB; ++qi)
#pragma unroll
                for (int u = 0; u < 4; ++u) dot[qi][u] = 0.0;
            const T *rp = rb + j0;
            using Raw = typename std::conditional<sizeof(T) == 4, float4, uint2>::type;
            auto use = [&](const Raw &raw, int c) {
                float rv[4];
                if constexpr (sizeof(T) == 4) {
                    rv[0] = raw.x;
                    rv[1] = raw.y;
                    rv[2] = raw.z;
                    rv[3] = raw.w;
                } else {
                    rv[0] = __uint_as_float(raw.x << 16);
                    rv[1] = __uint_as_float(raw.x & 0xffff0000u);
                    rv[2] = __uint_as_float(raw.y << 16);
                    rv[3] = __uint_as_float(raw.y & 0xffff0000u);
                }
#pragma unroll
                for (int qi = 0; qi < kQB; ++qi) {
                    const double qv = qs[qi * C + c];
#pragma unroll
                    for (int u = 0; u < 4; ++u) dot[qi][u] = fma(qv, (double)rv[u], dot[qi][u]);
                }
            };
            // Batches of kDepth channel rows, two buffers: the rows of the next batch are requested before the
            // products of the current one start, and the scheduling barriers keep the compiler from folding
            // that back into load - wait - use.  No control flow inside a batch; a batch past the end re-reads
            // the last row and is never used.
            constexpr int kDepth = sizeof(T) == 4 ? 4 : 8;      // 64 bytes per buffer: two waves per SIMD still fit
            auto load = [&](Raw(&dst)[kDepth], int cb) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < kDepth; ++e) {
                    const int c = cb + e < C ? cb + e : C - 1;
                    dst[e] = *reinterpret_cast<const Raw *>(rp + (int64_t)c * nr);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto use_batch = [&](const Raw(&src)[kDepth], int cb) {
#pragma unroll
                for (int e = 0; e < kDepth; ++e) use(src[e], cb + e);
            };
            const int nb = C / kDepth;
            int c0 = 0;
            if (nb > 0) {
                Raw even[kDepth], odd[kDepth];
                load(even, 0);
                int i = 0;
                for (; i + 2 <= nb; i += 2) {
                    load(odd, (i + 1) * kDepth);
                    use_batch(even, i * kDepth);
                    load(even, (i + 2) * kDepth);
                    use_batch(odd, (i + 1) * kDepth);
                }
                if (i < nb) use_batch(even, i * kDepth);
                c0 = nb * kDepth;
            }
            for (; c0 < C; ++c0) use(*reinterpret_cast<const Raw *>(rp + (int64_t)c0 * nr), c0);
#pragma unroll
            for (int qi = 0; qi < kQB; ++qi) {
                const int64_t i = pair_list[(int64_t)b * nq + base + (qi < nqry ? qi : 0)];
                const double qq = qq64[(int64_t)b * nq + i];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double d = (qq + rrb[j0 + u]) - 2.0 * dot[qi][u];
                    if (key_less(d, (int)(j0 + u), bd[qi], bj[qi])) {
                        bd[qi] = d;
                        bj[qi] = (int)(j0 + u);
                    }
                }
            }
        }
        // block-wide lexicographic minimum per query
        for (int qi = 0; qi < nqry; ++qi) {
            __syncthreads();
            double md = bd[0];
            int mj = bj[0];
#pragma unroll
            for (int t = 1; t < kQB; ++t)
                if (t == qi) {
                    md = bd[t];
                    mj = bj[t];
                }
            sd[threadIdx.x] = md;
            sj[threadIdx.x] = mj;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (threadIdx.x < off) {
                    const int o = threadIdx.x + off;
                    if (key_less(sd[o], sj[o], sd[threadIdx.x], sj[threadIdx.x])) {
                        sd[threadIdx.x] = sd[o];
                        sj[threadIdx.x] = sj[o];
                    }
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                const int64_t i = pair_list[(int64_t)b * nq + base + qi];
                const int64_t o = (int64_t)b * nq + i;                 // k_out == 1
                idx[o] = sj[0] + index_base;
                dist[o] = (float)sd[0];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Plan {
    bool bf16_mfma;          // native bf16 MFMA filter (else the float32 MFMA filter)
    bool t16;                // ... on the 16x16x32 shape (top-1 only)
    bool split;              // float32 inputs through the split-bfloat16 filter (top-1 only)
    size_t off_qhi, off_qlo, off_rhi, off_rlo;
    int tq;                  // queries per block of the chosen filter
    int n_qtiles, nsplit, nparts, ktemplate;
    int64_t refs_per_split;
    size_t off_qnh, off_rnh, off_q64, off_r64, off_rmax, off_count, off_pcount, off_list, off_plist, off_llist, off_pval, off_pidx, total;
    bool normalize;          // ISPA_NN_NORMALIZE: float32 inputs are L2-normalised into copies first
    size_t off_yq, off_yr;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

Plan make_plan(int dtype_and_flags, int B, int C, int64_t nq, int64_t nr, int k) {
    Plan p;
    const int dtype = dtype_and_flags & ~ISPA_NN_NORMALIZE;
    p.normalize = (dtype_and_flags & ISPA_NN_NORMALIZE) != 0 && dtype != ISPA_BF16;
    const bool fast_shape = C % 32 == 0 && nq % 8 == 0 && nr % 8 == 0 && nr < (1 << 27);   // 32-bit lane offsets
    p.bf16_mfma = dtype == ISPA_BF16 && fast_shape && C <= 256;
    p.split = dtype == ISPA_F32_SPLIT && fast_shape && C <= 256 && k == 1;
    p.tq = p.split ? kSQ : (p.bf16_mfma ? kBQ : kTQ);
    const int chunk_refs = (p.bf16_mfma || p.split) ? kBR : kTR;
    p.t16 = p.bf16_mfma && k == 1 && !ISPA_TOP1_32X32;          // top-1: the 16x16x32 kernel
    const int substreams =
        (p.t16 || p.split) ? kT16SubStreams : (p.bf16_mfma ? kBSubStreams : kSubStreams);
    p.n_qtiles = (int)((nq + p.tq - 1) / p.tq);
    p.ktemplate = k == 1 ? 1 : (k <= 5 ? 5 : kMaxK);
    const int64_t chunks = (nr + chunk_refs - 1) / chunk_refs;
    // enough workgroups to fill 256 CUs a few times over, without going below 2 chunks per split
    const int64_t want = 1024;
    int64_t nsplit = 1;
    const int64_t base = (int64_t)B * p.n_qtiles;
    // The LDS-staged float32 filter (four workgroups per CU) keeps the plain rule: about `want` workgroups.
    const bool staged = !p.bf16_mfma && !p.split && !(p.ktemplate == 1 && C % 32 == 0 && C <= 128);
    if (!staged) {
        // The other filters hold two workgroups per CU up to 128 channels and one above, and every
        // split stream pays a fixed price again: its query fragments and, for top-k, a sorted insert
        // for nearly every score of its first few hundred (measured: what ~800 further references
        // would cost; ~256 for top-1).  So: rounds of workgroups over the chip x (references per
        // split + that price), smallest wins.
        const int64_t slots = 256 * (C > 128 ? 1 : 2);
        const int64_t warm = p.ktemplate > 1 ? 800 : 256;
        int64_t best_cost = -1;
        for (int64_t n = 1; n == 1 || chunks / n >= 2; n *= 2) {
            const int64_t cost = ((base * n + slots - 1) / slots) * ((chunks + n - 1) / n * chunk_refs + warm);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                nsplit = n;
            }
        }
    } else {
        while (base * nsplit < want && chunks / (nsplit * 2) >= 2) nsplit *= 2;
    }
    const int64_t cps = (chunks + nsplit - 1) / nsplit;
    p.refs_per_split = cps * chunk_refs;
    p.nsplit = (int)((chunks + cps - 1) / cps);
    p.nparts = p.nsplit * substreams;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    p.off_qnh = take(sizeof(float) * B * nq);
    p.off_rnh = take(sizeof(float) * B * nr);
    p.off_q64 = take(sizeof(double) * B * nq);
    p.off_r64 = take(sizeof(double) * B * nr);
    p.off_rmax = take(sizeof(int) * B);
    p.off_count = take(16);
    p.off_pcount = take(sizeof(int) * B);
    p.off_list = take(sizeof(int) * B * nq);
    p.off_plist = take(sizeof(int) * B * nq);
    p.off_llist = take(sizeof(int) * 2 * B * nq);
    p.off_pval = take(sizeof(float) * (size_t)B * p.nparts * (p.ktemplate + 1) * nq);
    p.off_pidx = take(sizeof(int) * (size_t)B * p.nparts * (p.ktemplate + 1) * nq);
    p.off_qhi = p.off_qlo = p.off_rhi = p.off_rlo = 0;
    if (p.split) {
        p.off_qhi = take(sizeof(uint16_t) * (size_t)B * C * nq);
        p.off_qlo = take(sizeof(uint16_t) * (size_t)B * C * nq);
        p.off_rhi = take(sizeof(uint16_t) * (size_t)B * C * nr);
        p.off_rlo = take(sizeof(uint16_t) * (size_t)B * C * nr);
    }
    p.off_yq = p.off_yr = 0;
    if (p.normalize) {
        p.off_yq = take(sizeof(float) * (size_t)B * C * nq);
        p.off_yr = take(sizeof(float) * (size_t)B * C * nr);
    }
    p.total = off;
    return p;
}

template <int KS, int KT>
int launch_bf16(dim3 grid, hipStream_t s, const uint16_t *q, const uint16_t *r, const float *rnh, int64_t nq,
                int64_t nr, const Plan &p, float *pval, int *pidx) {
    constexpr int lds_bytes = bf16_filter_lds_bytes<KS>();
    // more than 128 channels (one workgroup per CU): 8 waves of 32 queries where the chunk pieces divide evenly
    constexpr int QT = (KS > 8 && KS % 4 == 0) ? 1 : 2;
    static unsigned long long configured = 0;       // per instantiation, one bit per device
    if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_filter_bf16_kernel<KS, KT, QT>), lds_bytes,
                                         &configured))
        return rc;
    hipLaunchKernelGGL((nn_filter_bf16_kernel<KS, KT, QT>), grid, dim3(512 / QT), lds_bytes, s, q, r, rnh, nq, nr,
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx);
    return ISPA_OK;
}

#ifndef ISPA_T16_NQ
#define ISPA_T16_NQ 4        // development switch: 2 = the 8-wave form of the top-1 kernel for 64 / 128 channels
#endif
template <int KS32>
int launch_t16(dim3 grid, hipStream_t s, const uint16_t *q, const uint16_t *r, const float *rnh, int64_t nq,
               int64_t nr, const Plan &p, float *pval, int *pidx) {
    constexpr int lds_bytes = bf16_filter_lds_bytes<2 * KS32>();
    // 4 waves of 64 queries; the 8-wave form (64 or 128 channels) is slower: 1.88 against 1.77 ms per 512 pairs
    constexpr int NQ = (KS32 % 2 == 0 && KS32 <= 4) ? ISPA_T16_NQ : 4;
    static unsigned long long configured = 0;       // per instantiation, one bit per device
    if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_filter_bf16_t16_kernel<KS32, NQ>), lds_bytes,
                                         &configured))
        return rc;
    hipLaunchKernelGGL((nn_filter_bf16_t16_kernel<KS32, NQ>), grid, dim3(1024 / NQ), lds_bytes, s, q, r, rnh, nq, nr,
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx);
    return ISPA_OK;
}

template <int KS32>
int launch_split(dim3 grid, hipStream_t s, const uint16_t *qhi, const uint16_t *qlo, const uint16_t *rhi,
                 const uint16_t *rlo, const float *rnh, int64_t nq, int64_t nr, const Plan &p, float *pval,
                 int *pidx) {
    constexpr int lds_bytes = split_filter_lds_bytes<KS32>();
    static unsigned long long configured = 0;
    if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_filter_split_kernel<KS32>), lds_bytes,
                                         &configured))
        return rc;
    hipLaunchKernelGGL((nn_filter_split_kernel<KS32>), grid, dim3(256), lds_bytes, s, qhi, qlo, rhi, rlo, rnh, nq,
                       nr, p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx);
    return ISPA_OK;
}

template <int KT>
int launch_bf16_ks(int ks, dim3 grid, hipStream_t s, const uint16_t *q, const uint16_t *r, const float *rnh,
                   int64_t nq, int64_t nr, const Plan &p, float *pval, int *pidx) {
    switch (ks) {
        case 2: return launch_bf16<2, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        case 4: return launch_bf16<4, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        case 6: return launch_bf16<6, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        case 8: return launch_bf16<8, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        case 10: return launch_bf16<10, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        case 12: return launch_bf16<12, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        case 14: return launch_bf16<14, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
        default: return launch_bf16<16, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx);
    }
}

template <typename T>
int run(const T *q, const T *r, int B, int C, int64_t nq, int64_t nr, int k, int index_base, float *dist,
        int32_t *idx, char *ws, const Plan &p, int32_t *stats, hipStream_t s) {
    float *qnh = (float *)(ws + p.off_qnh);
    float *rnh = (float *)(ws + p.off_rnh);
    double *q64 = (double *)(ws + p.off_q64);
    double *r64 = (double *)(ws + p.off_r64);
    int *rmax = (int *)(ws + p.off_rmax);
    int *count = (int *)(ws + p.off_count);
    int *list = (int *)(ws + p.off_list);
    int *pcount = (int *)(ws + p.off_pcount);
    int *plist = (int *)(ws + p.off_plist);
    int *llist = (int *)(ws + p.off_llist);
    float *pval = (float *)(ws + p.off_pval);
    int *pidx = (int *)(ws + p.off_pidx);

    // A call on a pair or two is a chain of short dependent launches: it takes the variants with more threads
    // per query and fewer launches (no memset, no separate re-scoring and refinement passes).
    const bool small = p.ktemplate == 1 && !p.split && !p.normalize && (int64_t)B * nq <= kSmallQueries &&
                       nr <= kSmallRefs && p.nparts <= kFsWaves * kFsParts;
    // rmax and count are adjacent blocks; zero both (sizes are multiples of 256 bytes)
    if (!small) ISPA_HIP(hipMemsetAsync(ws + p.off_rmax, 0, p.off_list - p.off_rmax, s));
    uint16_t *qhi = (uint16_t *)(ws + p.off_qhi), *qlo = (uint16_t *)(ws + p.off_qlo);
    uint16_t *rhi = (uint16_t *)(ws + p.off_rhi), *rlo = (uint16_t *)(ws + p.off_rlo);
    bool norms_done = false;
    if (p.normalize && sizeof(T) == 4) {
        // L2-normalised copies of both maps + their norms in one sweep; everything below reads the copies
        float *yq = (float *)(ws + p.off_yq), *yr = (float *)(ws + p.off_yr);
        hipLaunchKernelGGL(nn_l2norm_norms_kernel, dim3(ispa::blocks_for(nq, 256), B), dim3(256), 0, s,
                           reinterpret_cast<const float *>(q), C, nq, yq, qnh, q64, (int *)nullptr);
        hipLaunchKernelGGL(nn_l2norm_norms_kernel, dim3(ispa::blocks_for(nr, 256), B), dim3(256), 0, s,
                           reinterpret_cast<const float *>(r), C, nr, yr, rnh, r64, rmax);
        q = reinterpret_cast<const T *>(yq);
        r = reinterpret_cast<const T *>(yr);
        norms_done = !p.split;                       // the split filter still needs its hi / lo images
    }
    if (norms_done) {
    } else if (small) {
        const unsigned nbq = ispa::blocks_for(nq, 64), nbr = ispa::blocks_for(nr, 64);
        hipLaunchKernelGGL((nn_norms_small_kernel<T>), dim3(nbq + nbr, B), dim3(64), 0, s, q, r, C, nq, nr, (int)nbq,
                           qnh, q64, rnh, r64, rmax, (int)((p.off_list - p.off_rmax) / sizeof(int)));
    } else if (p.split) {
        const float *qf = reinterpret_cast<const float *>(q), *rf = reinterpret_cast<const float *>(r);
        hipLaunchKernelGGL(nn_split_norms_kernel, dim3(ispa::blocks_for(nq, 256), B), dim3(256), 0, s, qf, C, nq, qhi,
                           qlo, qnh, q64, (int *)nullptr);
        hipLaunchKernelGGL(nn_split_norms_kernel, dim3(ispa::blocks_for(nr, 256), B), dim3(256), 0, s, rf, C, nr, rhi,
                           rlo, rnh, r64, rmax);
    } else {
        // workspace blocks are 256-byte aligned, so nh / sq64 rows are aligned whenever n % 4 == 0
        const bool q4 = nq % 4 == 0 && (uintptr_t)q % 16 == 0, r4 = nr % 4 == 0 && (uintptr_t)r % 16 == 0;
        if (q4 && r4) {
            const unsigned nbq = ispa::blocks_for(nq / 4, 256), nbr = ispa::blocks_for(nr / 4, 256);
            hipLaunchKernelGGL((nn_norms4_both_kernel<T>), dim3(nbq + nbr, B), dim3(256), 0, s, q, r, C, nq, nr,
                               (int)nbq, qnh, q64, rnh, r64, rmax);
        } else {
            if (q4)
                hipLaunchKernelGGL((nn_norms4_kernel<T>), dim3(ispa::blocks_for(nq / 4, 256), B), dim3(256), 0, s, q, C,
                                   nq, qnh, q64, (int *)nullptr);
            else
                hipLaunchKernelGGL((nn_norms_kernel<T>), dim3(ispa::blocks_for(nq, 256), B), dim3(256), 0, s, q, C, nq,
                                   qnh, q64, (int *)nullptr);
            if (r4)
                hipLaunchKernelGGL((nn_norms4_kernel<T>), dim3(ispa::blocks_for(nr / 4, 256), B), dim3(256), 0, s, r, C,
                                   nr, rnh, r64, rmax);
            else
                hipLaunchKernelGGL((nn_norms_kernel<T>), dim3(ispa::blocks_for(nr, 256), B), dim3(256), 0, s, r, C, nr,
                                   rnh, r64, rmax);
        }
    }
    if (int rc = ispa::check_launch("nn_norms_kernel")) return rc;

    const int64_t nblocks = (int64_t)B * p.nsplit * p.n_qtiles;
    ISPA_REQUIRE(nblocks < (1ll << 31), "problem too large for one launch");
    const dim3 grid((unsigned)nblocks);
    const ispa::TimingSlot timed = ispa::timing_begin(s);
    // Rounding model handed to the finalize pass: two scores a, b are ordered for certain when they differ by
    // more than err_rel * (|q| max|r| + max|r|^2 / 2) + tag_rel * max(|a|, |b|).  tag_rel covers the mantissa
    // bits a filter replaces by a position tag (two scores, 1.25 safety, and 0.1 % so that the bound also
    // holds for every score within the bound below a).
    float err_rel, tag_rel = 0.f;
    const float u = 5.9604645e-8f;   // 2^-24
    const float kTag4 = 2.f * 1.25f * 1.001f * 1.9073486e-6f;     // 4 tagged bits: < 2^-19 of the score
    const float kTag7 = 2.f * 1.25f * 1.001f * 1.5258789e-5f;     // kTagBits = 7: < 2^-16
    if (p.split) {
        // per score: three MFMAs per 32 channels at 2 ulp each (same model as below), what the split
        // drops (at most 3.1 * 2^-16 |q||r|), the 4 tagged mantissa bits; two scores, 1.25 safety
        err_rel = 2.f * 1.25f * (2.f * (float)(3 * (C / 32) + 1) * 2.f * u + 3.1f * 1.5258789e-5f + 1.9073486e-6f);
        int rc;
        switch (C / 32) {
            case 1: rc = launch_split<1>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            case 2: rc = launch_split<2>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            case 3: rc = launch_split<3>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            case 4: rc = launch_split<4>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            case 5: rc = launch_split<5>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            case 6: rc = launch_split<6>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            case 7: rc = launch_split<7>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
            default: rc = launch_split<8>(grid, s, qhi, qlo, rhi, rlo, rnh, nq, nr, p, pval, pidx); break;
        }
        if (rc) return rc;
    } else if (p.bf16_mfma) {
        // rounding model of v_mfma_f32_32x32x16_bf16: products of two bf16 are exact in float32; each
        // instruction adds 16 of them to the accumulator with a total error of at most 2 units in
        // the last place of the largest magnitude involved.  (C/16 + 1) such steps, two scores
        // compared, 1.25 safety.  tests/test_gpu_parity.py measures the real error against this.
        err_rel = 2.f * 1.25f * 2.f * (float)(C / 16 + 1) * 2.f * u;
        tag_rel = p.ktemplate == 1 ? kTag4 : kTag7;
        const uint16_t *qh = reinterpret_cast<const uint16_t *>(q), *rh = reinterpret_cast<const uint16_t *>(r);
        int rc;
        if (p.t16) {
            switch (C / 32) {
                case 1: rc = launch_t16<1>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                case 2: rc = launch_t16<2>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                case 3: rc = launch_t16<3>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                case 4: rc = launch_t16<4>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                case 5: rc = launch_t16<5>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                case 6: rc = launch_t16<6>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                case 7: rc = launch_t16<7>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
                default: rc = launch_t16<8>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx); break;
            }
        }
#if ISPA_TOP1_32X32
        else if (p.ktemplate == 1) rc = launch_bf16_ks<1>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx);
#endif
        else if (p.ktemplate == 5) rc = launch_bf16_ks<5>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx);
        else rc = launch_bf16_ks<kMaxK>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx);
        if (rc) return rc;
    } else {
        // v_mfma_f32_32x32x2_f32 is a k-ordered float32 fma chain: C+1 roundings of at most
        // u * (|q||r| + |r|^2/2) each; two scores compared, 1.25 safety.
        err_rel = 2.f * 1.25f * (float)(C + 2) * u;
        tag_rel = p.ktemplate == 1 ? kTag4 : kTag7;      // both list types of the float32 filters are tagged
        const int elem = (int)sizeof(T);
        const int vec_ok = (nq % 4 == 0) && (nr % 4 == 0) && ((uintptr_t)q % (4 * elem) == 0) &&
                           ((uintptr_t)r % (4 * elem) == 0);
        const bool direct = sizeof(T) == 4 && vec_ok && C % kKC == 0 && (uintptr_t)q % 16 == 0 &&
                            (uintptr_t)r % 16 == 0 && nq < (1 << 28) && nr < (1 << 28);   // 32-bit lane offsets
#define ISPA_F32_FILTER(KT, D)                                                                                 \
    hipLaunchKernelGGL((nn_filter_f32_kernel<T, KT, D>), grid, dim3(kThreads), 0, s, q, r, rnh, C, nq, nr,    \
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx, vec_ok)
        // top-1 with at most 128 channels: the variant that keeps the query fragments in registers
        const bool qreg = direct && C % 32 == 0 && C <= 128 && p.ktemplate == 1;
        if (qreg) {
            const float *qf = reinterpret_cast<const float *>(q), *rf = reinterpret_cast<const float *>(r);
#define ISPA_QREG(NKS)                                                                                        \
    hipLaunchKernelGGL((nn_filter_f32_qreg_kernel<NKS, 1>), grid, dim3(kThreads), 0, s, qf, rf, rnh, nq, nr,   \
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx)
            switch (C / 32) {
                case 1: ISPA_QREG(2); break;
                case 2: ISPA_QREG(4); break;
                case 3: ISPA_QREG(6); break;
                default: ISPA_QREG(8); break;
            }
#undef ISPA_QREG
        } else if (p.ktemplate == 1) {
            if (direct) ISPA_F32_FILTER(1, true); else ISPA_F32_FILTER(1, false);
        } else if (p.ktemplate == 5) {
            if (direct) ISPA_F32_FILTER(5, true); else ISPA_F32_FILTER(5, false);
        } else {
            if (direct) ISPA_F32_FILTER(kMaxK, true); else ISPA_F32_FILTER(kMaxK, false);
        }
#undef ISPA_F32_FILTER
    }
    ispa::timing_end(timed, s);
    if (int rc = ispa::check_launch("nn_filter")) return rc;

    if (small) {
        hipLaunchKernelGGL((nn_finalize_small_kernel<T>), dim3(ispa::blocks_for(nq, 64), B), dim3(64 * kFsWaves), 0, s,
                           pval, pidx, p.nparts, qnh, C, nq, nr, index_base, err_rel, tag_rel, q, r, q64, r64, dist, idx, count,
                           list, pcount, plist, count + 1);
        if (int rc = ispa::check_launch("nn_finalize_small_kernel")) return rc;
    }
    const dim3 fgrid(ispa::blocks_for(nq, 256), B);
#define ISPA_FINALIZE(KT)                                                                                     \
    hipLaunchKernelGGL((nn_finalize_kernel<T, KT>), fgrid, dim3(256), 0, s, pval, pidx, p.nparts, qnh, rmax, C, nq, \
                       nr, k, index_base, err_rel, tag_rel, q, r, q64, r64, dist, idx, count, list, pcount, plist, count + 1, llist)
    if (small) {
    } else if (p.ktemplate == 1) ISPA_FINALIZE(1);
    else if (p.ktemplate == 5) ISPA_FINALIZE(5);
    else ISPA_FINALIZE(kMaxK);
#undef ISPA_FINALIZE
    if (int rc = ispa::check_launch("nn_finalize_kernel")) return rc;
    if (small) {
    } else if (p.ktemplate == 1) {
        // a wave per queued query, as many as the chip holds at once (256 CUs x 32 waves) for large batches
        const int64_t waves = (int64_t)B * nq < 8192 ? (int64_t)B * nq : 8192;
        hipLaunchKernelGGL((nn_rescore_top1_kernel<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, pval, pidx,
                           p.nparts, q, r, q64, r64, C, nq, nr, index_base, count + 1, llist, dist, idx);
        if (int rc = ispa::check_launch("nn_rescore_top1_kernel")) return rc;
    } else {
        const int64_t waves = (int64_t)B * nq < 8192 ? (int64_t)B * nq : 8192;
#define ISPA_RESCORE_TOPK(KT)                                                                                  \
    hipLaunchKernelGGL((nn_rescore_topk_kernel<T, KT>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, pval,  \
                       pidx, p.nparts, q, r, q64, r64, C, nq, nr, k, index_base, count + 1, llist, dist, idx)
        if (p.ktemplate == 5) ISPA_RESCORE_TOPK(5);
        else ISPA_RESCORE_TOPK(kMaxK);
#undef ISPA_RESCORE_TOPK
        if (int rc = ispa::check_launch("nn_rescore_topk_kernel")) return rc;
    }

    // float64 pass: pairs with long queues (when queues are long on average) share one sweep over
    // their reference map among kQB queries; everything else goes one query per block
    // (only the split filter's wider bound makes queues long enough: the other paths skip the launch)
    const bool batched = p.split && k == 1 && nr % 4 == 0 && (uintptr_t)r % 16 == 0 &&
                         (size_t)kQB * C * sizeof(double) <= 48 * 1024;
    if (batched) {
        int per_pair = (int)(2048 / B);
        per_pair = per_pair < 4 ? 4 : (per_pair > 64 ? 64 : per_pair);
        const size_t esmem = (size_t)kQB * C * sizeof(double) + 256 * (sizeof(double) + sizeof(int));
        hipLaunchKernelGGL((nn_exact_top1_kernel<T>), dim3(B, per_pair), dim3(256), esmem, s, q, r, q64, r64, C, nq,
                           nr, index_base, count, pcount, plist, dist, idx);
        if (int rc = ispa::check_launch("nn_exact_top1_kernel")) return rc;
    }
    const int64_t total_q = (int64_t)B * nq;
    const unsigned eblocks = (unsigned)(total_q < 4096 ? total_q : 4096) / (small ? 4 : 1) + (small ? 1 : 0);
    // (the split filter queues thousands of queries: throughput, not latency, so it keeps the small blocks)
    const int ethreads = nr >= 8192 && !p.split ? 1024 : 256;
    const size_t esmem = (size_t)C * sizeof(double) + ethreads * (sizeof(double) + 2 * sizeof(int));
    ISPA_REQUIRE(esmem <= 64 * 1024, "too many channels (%d) for the exact pass", C);
    // 16-byte loads of consecutive references need aligned rows
    const int evec = 16 / (int)sizeof(T);
    const int exact_vec_ok = nr % evec == 0 && (uintptr_t)r % 16 == 0;
    if (ethreads == 1024)
        hipLaunchKernelGGL((nn_exact_kernel<T, 1024>), dim3(eblocks), dim3(1024), esmem, s, q, r, q64, r64, C, nq, nr,
                           k, index_base, exact_vec_ok, count, list, batched ? kBatchMin : 0x7fffffff, B, pcount,
                           dist, idx);
    else
        hipLaunchKernelGGL((nn_exact_kernel<T, 256>), dim3(eblocks), dim3(256), esmem, s, q, r, q64, r64, C, nq, nr,
                           k, index_base, exact_vec_ok, count, list, batched ? kBatchMin : 0x7fffffff, B, pcount,
                           dist, idx);
    if (int rc = ispa::check_launch("nn_exact_kernel")) return rc;
    if (!small) {
        static unsigned long long configured = 0;       // per instantiation, one bit per device
        if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_refine_kernel<T>), kRefineLds, &configured))
            return rc;
        const int64_t slices = ((int64_t)k * nq + kRefineEntries - 1) / kRefineEntries;
        hipLaunchKernelGGL((nn_refine_kernel<T>), dim3((unsigned)slices, B), dim3(kRefineThreads), kRefineLds, s, q, r,
                           q64, r64, C, nq, nr, k, index_base, exact_vec_ok, idx, dist);
        if (int rc = ispa::check_launch("nn_refine_kernel")) return rc;
    }
    if (stats) ISPA_HIP(hipMemcpyAsync(stats, count, 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
    return ISPA_OK;
}

}  // namespace

extern "C" size_t ispa_nn_workspace_bytes_for(int dtype, int B, int C, int64_t nq, int64_t nr, int k) {
    if (B <= 0 || C <= 0 || nq <= 0 || nr <= 0 || k < 1 || k > kMaxK) return 0;
    const int base = dtype & ~ISPA_NN_NORMALIZE, flag = dtype & ISPA_NN_NORMALIZE;
    if (base != ISPA_F32 && base != ISPA_BF16 && base != ISPA_F32_SPLIT) return 0;
    if (flag && base == ISPA_BF16) return 0;
    const size_t a = make_plan(dtype, B, C, nq, nr, k).total;
    const size_t b = make_plan(ISPA_F32 | flag, B, C, nq, nr, k).total;   // the plan a call may fall back to
    return a > b ? a : b;
}

extern "C" size_t ispa_nn_workspace_bytes(int B, int C, int64_t nq, int64_t nr, int k) {
    if (B <= 0 || C <= 0 || nq <= 0 || nr <= 0 || k < 1 || k > kMaxK) return 0;
    // the larger of the two dtype plans, so that one workspace serves both
    const size_t a = make_plan(ISPA_F32, B, C, nq, nr, k).total;
    const size_t b = make_plan(ISPA_BF16, B, C, nq, nr, k).total;
    return a > b ? a : b;
}

namespace {
int nn_topk_checked(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr, int k,
                    int index_base, float *dist, int32_t *idx, void *workspace, size_t workspace_bytes,
                    int32_t *stats, void *stream, const char *workspace_name);

// Workspace of ispa_nn_topk_noalloc: one array per GPU inside the loaded code object.  In host code the
// name is only a handle for hipGetSymbolAddress; the device address it yields differs per GPU.
static __device__ __align__(256) unsigned char scratch[ISPA_NN_NOALLOC_SCRATCH_BYTES];
constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];      // device address of `scratch` per GPU, looked up on first use

int scratch_for_current_device(void **out) {
    int dev = 0;
    ISPA_HIP(hipGetDevice(&dev));
    ISPA_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal %d outside [0, %d)", dev, kMaxDevices);
    void *p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        ISPA_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));    // with device `dev` current: its copy
        ISPA_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address for the scratch array");
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE); // racing host threads store the same value
    }
    *out = p;
    return ISPA_OK;
}
}  // namespace

extern "C" size_t ispa_nn_noalloc_scratch_bytes(void) { return (size_t)ISPA_NN_NOALLOC_SCRATCH_BYTES; }

extern "C" int ispa_nn_topk_noalloc(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr,
                                    int k, int index_base, float *dist, int32_t *idx, int32_t *stats,
                                    void *stream) {
    ISPA_REQUIRE(B >= 0 && nq >= 0, "negative size");
    if (B == 0 || nq == 0) return ISPA_OK;
    ISPA_REQUIRE(k >= 1 && k <= kMaxK, "k = %d outside [1, %d]", k, kMaxK);
    ISPA_REQUIRE(C >= 1 && nr >= 1, "need at least one channel and one reference (C=%d, nr=%lld)", C, (long long)nr);
    const size_t need = ispa_nn_workspace_bytes_for(dtype, B, C, nq, nr, k);
    if (need == 0 || need > (size_t)ISPA_NN_NOALLOC_SCRATCH_BYTES)
        return ispa::fail(ISPA_EWORKSPACE,
                          "problem too large for ispa_nn_topk_noalloc: it needs %zu bytes of scratch, the static "
                          "buffer holds %zu (use ispa_nn_topk with a caller workspace)",
                          need, (size_t)ISPA_NN_NOALLOC_SCRATCH_BYTES);
    void *ws = nullptr;
    if (int rc = scratch_for_current_device(&ws)) return rc;
    return nn_topk_checked(q, r, dtype, B, C, nq, nr, k, index_base, dist, idx, ws,
                           (size_t)ISPA_NN_NOALLOC_SCRATCH_BYTES, stats, stream, "scratch");
}

extern "C" int ispa_nn_topk(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr,
                            int k, int index_base, float *dist, int32_t *idx, void *workspace,
                            size_t workspace_bytes, int32_t *stats, void *stream) {
    return nn_topk_checked(q, r, dtype, B, C, nq, nr, k, index_base, dist, idx, workspace, workspace_bytes, stats,
                           stream, "workspace");
}

namespace {
int nn_topk_checked(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr, int k,
                    int index_base, float *dist, int32_t *idx, void *workspace, size_t workspace_bytes,
                    int32_t *stats, void *stream, const char *workspace_name) {
    ISPA_REQUIRE(B >= 0 && nq >= 0, "negative size");
    ISPA_REQUIRE(k >= 1 && k <= kMaxK, "k = %d outside [1, %d]", k, kMaxK);
    if (B == 0 || nq == 0) return ISPA_OK;
    ISPA_REQUIRE(C >= 1 && nr >= 1, "need at least one channel and one reference (C=%d, nr=%lld)", C,
                 (long long)nr);
    ISPA_REQUIRE(nr < (1ll << 30) && (int64_t)B * nq < (1ll << 31) && B <= 65535, "problem too large");
    ISPA_REQUIRE(q && r && dist && idx && workspace, "null pointer");
    const int flags = dtype & ISPA_NN_NORMALIZE;
    dtype &= ~ISPA_NN_NORMALIZE;
    ISPA_REQUIRE(dtype == ISPA_F32 || dtype == ISPA_BF16 || dtype == ISPA_F32_SPLIT, "unknown dtype %d", dtype);
    ISPA_REQUIRE(!(flags && dtype == ISPA_BF16), "ISPA_NN_NORMALIZE takes float32 features");
    Plan p = make_plan(dtype | flags, B, C, nq, nr, k);
    if (p.bf16_mfma && (((uintptr_t)q | (uintptr_t)r) % 16 != 0)) {
        p = make_plan(ISPA_F32, B, C, nq, nr, k);       // unaligned bf16 input: converting tile loader
        p.bf16_mfma = false;
        p.t16 = false;
        p.split = false;
    }
    if (workspace_bytes < p.total)
        return ispa::fail(ISPA_EWORKSPACE, "%s of %zu bytes given, %zu needed", workspace_name, workspace_bytes, p.total);
    ISPA_REQUIRE((uintptr_t)workspace % 256 == 0, "%s must be 256-byte aligned", workspace_name);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ISPA_F32 || dtype == ISPA_F32_SPLIT)
        return run((const float *)q, (const float *)r, B, C, nq, nr, k, index_base, dist, idx, (char *)workspace,
                   p, stats, s);
    return run((const uint16_t *)q, (const uint16_t *)r, B, C, nq, nr, k, index_base, dist, idx,
               (char *)workspace, p, stats, s);
}
}  // namespace
