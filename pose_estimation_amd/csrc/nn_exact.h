// Dense nearest neighbour, pass 4: float64 re-scoring of every reference for the queued queries (included by nn.hip).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// 4. exact float64 pass for the queued queries
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool key_less(double d0, int j0, double d1, int j1) {
    return d0 < d1 || (d0 == d1 && j0 < j1);
}

// Which of the two kernels below takes a queued query.  nn_exact_top1_kernel lets the queued queries of a pair
// share sweeps over the pair's reference map, which saves reads, not latency: it takes the pairs with at least
// `batch_min` queued queries, and only when queues are that long on average and the whole queue gives at least
// four dozen shared sweeps (a handful of long shared sweeps on an otherwise idle chip take longer than many short
// ones side by side: one 128x128 pair queues 140 queries, 18 sweeps).  Both kernels evaluate this same predicate.
constexpr int kBatchTotalMin = 384;
__device__ __forceinline__ bool batched_pass_runs(int total, int batch_min, int n_pair) {
    return total >= (int64_t)batch_min * n_pair && total >= kBatchTotalMin;
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
    // pairs with long queues belong to nn_exact_top1_kernel (skip_from is out of reach when it does not run)
    const bool some_batched = batched_pass_runs(count, skip_from, n_pair);
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
    if (count < kBatchMin || !batched_pass_runs(*total_count, kBatchMin, (int)gridDim.x)) return;
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

}  // namespace
