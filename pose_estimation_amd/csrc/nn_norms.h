// Dense nearest neighbour, pass 1: norms (included by nn.hip).
#pragma once

namespace {

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

}  // namespace
