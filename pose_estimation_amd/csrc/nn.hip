// Dense nearest-neighbour correspondence between two feature maps (squared L2, top-k).
//
// Contract: the `KNN` layer of the Caffe-UCN fork as the reference uses it
// (ucn_code/nets/train.prototxt:2116-2141; consumer hard_negative_processing_layer.py:175-192).
//
// Structure (one call = 4 launches on the caller's stream):
//   1. norms      nh[j] = -0.5 * |x_j|^2 for queries and references, max reference norm per pair
//   2. filter     MFMA contraction  s[j][i] = r_j . q_i - 0.5 |r_j|^2  (the accumulator starts at
//                 nh[j]), reduced in registers to the (k+1) largest s per query and partial
//                 stream; the Nq x Nr score matrix never leaves the register file
//   3. finalize   merges the partial streams; a query whose consecutive score gaps all exceed
//                 the rounding bound of the filter is final, the others are queued
//   4. exact      float64 re-scoring of every reference for the queued queries, in the channel
//                 order the CPU oracle uses, so the index order is exact under all inputs
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTQ = 128;       // queries per workgroup tile
constexpr int kTR = 128;       // references per chunk
constexpr int kKC = 32;        // channels per LDS stage
constexpr int kThreads = 256;  // 4 waves: 2 (reference halves) x 2 (query halves)
constexpr int kSubStreams = 4; // partial result streams per (tile, split): wave-row x lane-half
constexpr int kMaxK = 8;
constexpr float kNegInf = -__builtin_inff();

// Optional timing of the filter kernel (the dominant launch) with HIP events on the caller's stream.
struct FilterTiming {
    int capacity = 0, used = 0;
    hipEvent_t *start = nullptr, *stop = nullptr;
} g_timing;

__device__ __forceinline__ float load_as_float(const float *p) { return *p; }
__device__ __forceinline__ float load_as_float(const uint16_t *p) {
    return __uint_as_float(((uint32_t)*p) << 16);
}

// ---------------------------------------------------------------------------------------------
// 1. norms
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nn_norms_kernel(const T *__restrict__ x, int C, int64_t n,
                                                       float *__restrict__ nh, int *__restrict__ maxbits) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (j < n) {
        const T *col = x + (int64_t)b * C * n + j;
        for (int c = 0; c < C; ++c) {
            const float v = load_as_float(col + (int64_t)c * n);
            s = fmaf(v, v, s);
        }
        nh[(int64_t)b * n + j] = -0.5f * s;
    }
    if (maxbits) {
        // non-negative floats order like their bit patterns
        float m = s;
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_int(m));
    }
}

// ---------------------------------------------------------------------------------------------
// 2. filter pass, float32 MFMA (v_mfma_f32_32x32x2_f32: exact float32 fma chain over channels)
// ---------------------------------------------------------------------------------------------
template <int K>
struct TopList {            // K+1 best scores (descending) and the codes of the first K... all K+1
    float v[K + 1];
    int c[K + 1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            v[s] = kNegInf;
            c[s] = -1;
        }
    }
    __device__ __forceinline__ void push(float x, int code) {
        if (K == 1) {
            v[1] = __builtin_amdgcn_fmed3f(v[0], v[1], x);
            const bool better = x > v[0];
            c[0] = better ? code : c[0];
            v[0] = better ? x : v[0];
        } else {
            // replace the last entry if better, then bubble it up (branch-free)
            const bool in = x > v[K];
            v[K] = in ? x : v[K];
            c[K] = in ? code : c[K];
#pragma unroll
            for (int s = K; s > 0; --s) {
                const bool sw = v[s] > v[s - 1];
                const float hi = sw ? v[s] : v[s - 1], lo = sw ? v[s - 1] : v[s];
                const int chi = sw ? c[s] : c[s - 1], clo = sw ? c[s - 1] : c[s];
                v[s - 1] = hi;
                v[s] = lo;
                c[s - 1] = chi;
                c[s] = clo;
            }
        }
    }
};

// Guarded tile loader: rows = channels [c0, c0+kKC), columns = [n0, n0+128) of x (C, n) -> LDS.
template <typename T>
__device__ __forceinline__ void stage_tile(const T *__restrict__ x, int C, int64_t n, int c0, int64_t n0,
                                           float (*__restrict__ dst)[kTQ], bool vec_ok) {
    // 32 rows x 128 columns = 1024 groups of 4; 4 groups per thread
#pragma unroll
    for (int e = 0; e < 4; ++e) {
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
        *reinterpret_cast<float4 *>(&dst[row][col]) = val;
    }
}

template <typename T, int K>
__global__ __launch_bounds__(kThreads, 2) void nn_filter_f32_kernel(
    const T *__restrict__ q, const T *__restrict__ r, const float *__restrict__ rnh, int C, int64_t nq,
    int64_t nr, int n_qtiles, int nsplit, int64_t refs_per_split, float *__restrict__ part_val,
    int *__restrict__ part_idx, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float Qs[kKC][kTQ];
    __shared__ __attribute__((aligned(16))) float Rs[kKC][kTR];
    __shared__ float Nh[kTR];

    // block -> (pair, split, query tile)
    int bid = blockIdx.x;
    const int qt = bid % n_qtiles;
    bid /= n_qtiles;
    const int split = bid % nsplit;
    const int b = bid / nsplit;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const T *qb = q + (int64_t)b * C * nq;
    const T *rb = r + (int64_t)b * C * nr;
    const float *nhb = rnh + (int64_t)b * nr;
    const int64_t q0 = (int64_t)qt * kTQ;
    const int64_t rbeg = (int64_t)split * refs_per_split;
    const int64_t rend = min(rbeg + refs_per_split, nr);

    TopList<K> best[2];
    best[0].init();
    best[1].init();

    int chunk = 0;
    for (int64_t r0 = rbeg; r0 < rend; r0 += kTR, ++chunk) {
        __syncthreads();  // previous chunk's Nh / tiles fully consumed
        if (threadIdx.x < kTR) {
            const int64_t j = r0 + threadIdx.x;
            Nh[threadIdx.x] = j < rend ? nhb[j] : kNegInf;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float init = Nh[wr * 64 + ri * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h];
                acc[ri][0][reg] = init;
                acc[ri][1][reg] = init;
            }
        for (int c0 = 0; c0 < C; c0 += kKC) {
            if (c0) __syncthreads();
            stage_tile(qb, C, nq, c0, q0, Qs, vec_ok);
            stage_tile(rb, C, nr, c0, r0, Rs, vec_ok);
            __syncthreads();
#pragma unroll 4
            for (int kk = 0; kk < kKC; kk += 2) {
                const float a0 = Rs[kk + h][wr * 64 + l31];
                const float a1 = Rs[kk + h][wr * 64 + 32 + l31];
                const float b0 = Qs[kk + h][wq * 64 + l31];
                const float b1 = Qs[kk + h][wq * 64 + 32 + l31];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        // epilogue: this lane owns query column (wq*64 + qi*32 + l31) and 16 reference rows per tile
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) best[qi].push(acc[ri][qi][reg], chunk * 32 + ri * 16 + reg);
    }

    // one partial stream per (split, wr, h); finalize merges them
    const int nparts = nsplit * kSubStreams;
    const int part = split * kSubStreams + wr * 2 + h;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int64_t qcol = q0 + wq * 64 + qi * 32 + l31;
        if (qcol >= nq) continue;
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            const int code = best[qi].c[s];
            int ref = -1;
            if (code >= 0) {
                const int reg = code & 15, ri = (code >> 4) & 1;
                ref = (int)(rbeg + (int64_t)(code >> 5) * kTR + wr * 64 + ri * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
            }
            const int64_t o = (((int64_t)b * nparts + part) * (K + 1) + s) * nq + qcol;
            part_val[o] = best[qi].v[s];
            part_idx[o] = ref;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3. finalize
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void nn_finalize_kernel(const float *__restrict__ part_val,
                                                          const int *__restrict__ part_idx, int nparts,
                                                          const float *__restrict__ qnh,
                                                          const int *__restrict__ rmaxbits, int C, int64_t nq,
                                                          int64_t nr, int k_out, int index_base,
                                                          float err_factor, float *__restrict__ dist,
                                                          int *__restrict__ idx, int *__restrict__ amb_count,
                                                          int *__restrict__ amb_list) {
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
    for (int p = 0; p < nparts; ++p)
        for (int s = 0; s <= K; ++s) {
            const int64_t o = (((int64_t)b * nparts + p) * (K + 1) + s) * nq + i;
            const float x = part_val[o];
            const unsigned code = (unsigned)part_idx[o];   // -1 sorts after every real index
            if (!(x > kNegInf) && x == x) break;            // absent (NaN falls through)
            if (x > v[K] || (x == v[K] && code < (unsigned)c[K]) || x != x) {
                v[K] = x;
                c[K] = (int)code;
#pragma unroll
                for (int t = K; t > 0; --t) {
                    const bool sw = v[t] > v[t - 1] || (v[t] == v[t - 1] && (unsigned)c[t] < (unsigned)c[t - 1]) ||
                                    v[t] != v[t];
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
    // rounding bound of one filter score: (C+2) roundings of magnitude <= u * (|q||r| + |r|^2/2)
    const float qq = -2.f * qnh[(int64_t)b * nq + i];
    const float rmax2 = __int_as_float(rmaxbits[b]);
    const float scale = sqrtf(qq) * sqrtf(rmax2) + 0.5f * rmax2;
    const float delta = err_factor * (float)(C + 2) * 5.9604645e-8f * scale;
    bool sure = true;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (s >= k_out) break;
        if (!(v[s] > kNegInf)) {                   // fewer references than k (or NaN: exact pass)
            if (v[s] != v[s]) sure = false;
            break;
        }
        if (c[s] < 0) sure = false;                // cannot happen for distinct scores; be safe
        if (v[s + 1] != v[s + 1]) sure = false;
        if (v[s + 1] > kNegInf && !(v[s] - v[s + 1] > delta)) sure = false;
    }
    if (sure) {
        for (int s = 0; s < k_out; ++s) {
            const int64_t o = ((int64_t)b * k_out + s) * nq + i;
            idx[o] = c[s] >= 0 ? c[s] + index_base : -1;
            dist[o] = c[s] >= 0 ? qq - 2.f * v[s] : __builtin_inff();
        }
    } else {
        const int slot = atomicAdd(amb_count, 1);
        amb_list[slot] = (int)((int64_t)b * nq + i);   // b*nq+i < 2^31 is checked on the host
    }
}

// ---------------------------------------------------------------------------------------------
// 4. exact float64 pass for the queued queries
// ---------------------------------------------------------------------------------------------
struct DKey {
    double d;
    int j;
};
__device__ __forceinline__ bool key_less(double d0, int j0, double d1, int j1) {
    return d0 < d1 || (d0 == d1 && j0 < j1);
}

template <typename T>
__global__ __launch_bounds__(256) void nn_exact_kernel(const T *__restrict__ q, const T *__restrict__ r, int C,
                                                       int64_t nq, int64_t nr, int k_out, int index_base,
                                                       const int *__restrict__ amb_count,
                                                       const int *__restrict__ amb_list,
                                                       float *__restrict__ dist, int *__restrict__ idx) {
    __shared__ double sd[256];
    __shared__ int sj[256];
    __shared__ int sown[256];
    const int count = *amb_count;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int64_t flat = amb_list[w];
        const int b = (int)(flat / nq);
        const int64_t i = flat % nq;
        const T *qcol = q + (int64_t)b * C * nq + i;
        const T *rb = r + (int64_t)b * C * nr;
        double qq = 0.0;
        for (int c = 0; c < C; ++c) {
            const double v = (double)load_as_float(qcol + (int64_t)c * nq);
            qq = fma(v, v, qq);
        }
        // thread-local top-k over its strided share of the references, ascending (d, j)
        double bd[kMaxK];
        int bj[kMaxK];
        for (int s = 0; s < kMaxK; ++s) {
            bd[s] = __builtin_inf();
            bj[s] = 0x7fffffff;
        }
        for (int64_t j = threadIdx.x; j < nr; j += 256) {
            double dot = 0.0, rr = 0.0;
            for (int c = 0; c < C; ++c) {
                const double qv = (double)load_as_float(qcol + (int64_t)c * nq);
                const double rv = (double)load_as_float(rb + (int64_t)c * nr + j);
                dot = fma(qv, rv, dot);
                rr = fma(rv, rv, rr);
            }
            const double d = (qq + rr) - 2.0 * dot;
            if (key_less(d, (int)j, bd[k_out - 1], bj[k_out - 1])) {
                int pos = k_out - 1;
                while (pos > 0 && key_less(d, (int)j, bd[pos - 1], bj[pos - 1])) {
                    bd[pos] = bd[pos - 1];
                    bj[pos] = bj[pos - 1];
                    --pos;
                }
                bd[pos] = d;
                bj[pos] = (int)j;
            }
        }
        // k_out rounds of block-wide lexicographic minimum over the thread-local heads
        int head = 0;
        for (int s = 0; s < k_out; ++s) {
            __syncthreads();
            sd[threadIdx.x] = head < k_out ? bd[head] : __builtin_inf();
            sj[threadIdx.x] = head < k_out ? bj[head] : 0x7fffffff;
            sown[threadIdx.x] = threadIdx.x;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
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
            const int owner = sown[0];
            if (threadIdx.x == owner) ++head;
            if (threadIdx.x == 0) {
                const int64_t o = ((int64_t)b * k_out + s) * nq + i;
                const bool have = sj[0] != 0x7fffffff;
                idx[o] = have ? sj[0] + index_base : -1;
                dist[o] = have ? (float)sd[0] : __builtin_inff();
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Plan {
    int n_qtiles, nsplit, nparts, ktemplate;
    int64_t refs_per_split;
    size_t off_qnh, off_rnh, off_rmax, off_count, off_list, off_pval, off_pidx, total;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

Plan make_plan(int B, int64_t nq, int64_t nr, int k) {
    Plan p;
    p.n_qtiles = (int)((nq + kTQ - 1) / kTQ);
    p.ktemplate = k == 1 ? 1 : (k <= 5 ? 5 : kMaxK);
    const int64_t chunks = (nr + kTR - 1) / kTR;
    // aim for >= 1024 workgroups (4 per CU) without going below 2 chunks per split
    int64_t nsplit = 1;
    const int64_t base = (int64_t)B * p.n_qtiles;
    while (base * nsplit < 1024 && chunks / (nsplit * 2) >= 2) nsplit *= 2;
    int64_t cps = (chunks + nsplit - 1) / nsplit;
    p.refs_per_split = cps * kTR;
    p.nsplit = (int)((chunks + cps - 1) / cps);
    p.nparts = p.nsplit * kSubStreams;
    size_t off = 0;
    p.off_qnh = off;
    off = align_up(off + sizeof(float) * B * nq, 256);
    p.off_rnh = off;
    off = align_up(off + sizeof(float) * B * nr, 256);
    p.off_rmax = off;
    off = align_up(off + sizeof(int) * B, 256);
    p.off_count = off;
    off = align_up(off + 16, 256);
    p.off_list = off;
    off = align_up(off + sizeof(int) * B * nq, 256);
    p.off_pval = off;
    off = align_up(off + sizeof(float) * (size_t)B * p.nparts * (p.ktemplate + 1) * nq, 256);
    p.off_pidx = off;
    off = align_up(off + sizeof(int) * (size_t)B * p.nparts * (p.ktemplate + 1) * nq, 256);
    p.total = off;
    return p;
}

template <typename T>
int run(const T *q, const T *r, int B, int C, int64_t nq, int64_t nr, int k, int index_base, float *dist,
        int32_t *idx, char *ws, const Plan &p, int32_t *stats, hipStream_t s, float err_factor) {
    float *qnh = (float *)(ws + p.off_qnh);
    float *rnh = (float *)(ws + p.off_rnh);
    int *rmax = (int *)(ws + p.off_rmax);
    int *count = (int *)(ws + p.off_count);
    int *list = (int *)(ws + p.off_list);
    float *pval = (float *)(ws + p.off_pval);
    int *pidx = (int *)(ws + p.off_pidx);

    // rmax and count are adjacent blocks; zero both (sizes are multiples of 256 bytes)
    ISPA_HIP(hipMemsetAsync(ws + p.off_rmax, 0, p.off_list - p.off_rmax, s));
    hipLaunchKernelGGL((nn_norms_kernel<T>), dim3(ispa::blocks_for(nq, 256), B), dim3(256), 0, s, q, C, nq, qnh,
                       (int *)nullptr);
    hipLaunchKernelGGL((nn_norms_kernel<T>), dim3(ispa::blocks_for(nr, 256), B), dim3(256), 0, s, r, C, nr, rnh,
                       rmax);
    if (int rc = ispa::check_launch("nn_norms_kernel")) return rc;

    const int elem = (int)sizeof(T);
    const int vec = 16 / elem <= 4 ? 4 : 4;  // groups of 4 elements
    const int vec_ok = (nq % vec == 0) && (nr % vec == 0) && ((uintptr_t)q % (4 * elem) == 0) &&
                       ((uintptr_t)r % (4 * elem) == 0);
    const int64_t nblocks = (int64_t)B * p.nsplit * p.n_qtiles;
    ISPA_REQUIRE(nblocks < (1ll << 31), "problem too large for one launch");
    dim3 grid((unsigned)nblocks), block(kThreads);
    const bool timed = g_timing.used < g_timing.capacity;
#define ISPA_LAUNCH_FILTER(KT)                                                                              \
    if (timed) (void)hipEventRecord(g_timing.start[g_timing.used], s);                                      \
    hipLaunchKernelGGL((nn_filter_f32_kernel<T, KT>), grid, block, 0, s, q, r, rnh, C, nq, nr, p.n_qtiles,  \
                       p.nsplit, p.refs_per_split, pval, pidx, vec_ok);                                     \
    if (timed) (void)hipEventRecord(g_timing.stop[g_timing.used++], s);                                     \
    hipLaunchKernelGGL((nn_finalize_kernel<KT>), dim3(ispa::blocks_for(nq, 256), B), dim3(256), 0, s, pval, \
                       pidx, p.nparts, qnh, rmax, C, nq, nr, k, index_base, err_factor, dist, idx, count, list)
    if (p.ktemplate == 1) {
        ISPA_LAUNCH_FILTER(1);
    } else if (p.ktemplate == 5) {
        ISPA_LAUNCH_FILTER(5);
    } else {
        ISPA_LAUNCH_FILTER(kMaxK);
    }
#undef ISPA_LAUNCH_FILTER
    if (int rc = ispa::check_launch("nn_filter/finalize")) return rc;

    const int64_t total_q = (int64_t)B * nq;
    const unsigned eblocks = (unsigned)(total_q < 2048 ? total_q : 2048);
    hipLaunchKernelGGL((nn_exact_kernel<T>), dim3(eblocks), dim3(256), 0, s, q, r, C, nq, nr, k, index_base,
                       count, list, dist, idx);
    if (int rc = ispa::check_launch("nn_exact_kernel")) return rc;
    if (stats) ISPA_HIP(hipMemcpyAsync(stats, count, sizeof(int), hipMemcpyDeviceToDevice, s));
    return ISPA_OK;
}

}  // namespace

extern "C" size_t ispa_nn_workspace_bytes(int B, int C, int64_t nq, int64_t nr, int k) {
    (void)C;
    if (B <= 0 || nq <= 0 || nr <= 0 || k < 1 || k > kMaxK) return 0;
    return make_plan(B, nq, nr, k).total;
}

extern "C" int ispa_nn_topk(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr,
                            int k, int index_base, float *dist, int32_t *idx, void *workspace,
                            size_t workspace_bytes, int32_t *stats, void *stream) {
    ISPA_REQUIRE(B >= 0 && nq >= 0, "negative size");
    ISPA_REQUIRE(k >= 1 && k <= kMaxK, "k = %d outside [1, %d]", k, kMaxK);
    if (B == 0 || nq == 0) return ISPA_OK;
    ISPA_REQUIRE(C >= 1 && nr >= 1, "need at least one channel and one reference (C=%d, nr=%lld)", C,
                 (long long)nr);
    ISPA_REQUIRE(nr < (1ll << 30) && (int64_t)B * nq < (1ll << 31), "problem too large");
    ISPA_REQUIRE(q && r && dist && idx && workspace, "null pointer");
    ISPA_REQUIRE(dtype == ISPA_F32 || dtype == ISPA_BF16, "unknown dtype %d", dtype);
    const Plan p = make_plan(B, nq, nr, k);
    if (workspace_bytes < p.total)
        return ispa::fail(ISPA_EWORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, p.total);
    ISPA_REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // error factor: 2 scores compared, 1.25 safety on the float32 fma-chain bound
    if (dtype == ISPA_F32)
        return run((const float *)q, (const float *)r, B, C, nq, nr, k, index_base, dist, idx, (char *)workspace,
                   p, stats, s, 2.5f);
    return run((const uint16_t *)q, (const uint16_t *)r, B, C, nq, nr, k, index_base, dist, idx,
               (char *)workspace, p, stats, s, 2.5f);
}

extern "C" int ispa_timing_enable(int max_records) {
    ISPA_REQUIRE(max_records >= 0 && max_records <= 65536, "max_records out of range");
    for (int i = 0; i < g_timing.capacity; ++i) {
        (void)hipEventDestroy(g_timing.start[i]);
        (void)hipEventDestroy(g_timing.stop[i]);
    }
    delete[] g_timing.start;
    delete[] g_timing.stop;
    g_timing = FilterTiming();
    if (max_records == 0) return ISPA_OK;
    g_timing.start = new hipEvent_t[max_records];
    g_timing.stop = new hipEvent_t[max_records];
    for (int i = 0; i < max_records; ++i) {
        ISPA_HIP(hipEventCreate(&g_timing.start[i]));
        ISPA_HIP(hipEventCreate(&g_timing.stop[i]));
    }
    g_timing.capacity = max_records;
    return ISPA_OK;
}

extern "C" int ispa_timing_collect(float *host_ms, int capacity, int *host_count) {
    ISPA_REQUIRE(host_ms && host_count && capacity >= 0, "null pointer");
    int n = g_timing.used < capacity ? g_timing.used : capacity;
    for (int i = 0; i < n; ++i) {
        ISPA_HIP(hipEventSynchronize(g_timing.stop[i]));
        ISPA_HIP(hipEventElapsedTime(&host_ms[i], g_timing.start[i], g_timing.stop[i]));
    }
    *host_count = n;
    g_timing.used = 0;
    return ISPA_OK;
}
