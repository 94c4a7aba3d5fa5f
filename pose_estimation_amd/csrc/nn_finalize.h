// Dense nearest neighbour, pass 3: finalize, re-scoring of known contenders, distance refinement (included by nn.hip).
#pragma once

namespace {

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
    constexpr int PB = K <= 5 ? 8 : 4;
    for (int p0 = 0; p0 < nparts; p0 += PB) {
        float xs[PB][K + 1];
        int cs[PB][K + 1];
        __builtin_amdgcn_sched_barrier(0);              // all loads of the batch first, then their uses: left
#pragma unroll                                          // alone the compiler sinks each load to its use
        for (int e = 0; e < PB; ++e)
#pragma unroll
            for (int s = 0; s <= K; ++s) {
                const int pp = p0 + e < nparts ? p0 + e : nparts - 1;
                const int64_t o = (((int64_t)b * nparts + pp) * (K + 1) + s) * nq + i;
                xs[e][s] = part_val[o];
                cs[e][s] = part_idx[o];
            }
        __builtin_amdgcn_sched_barrier(0);
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
    int C, int64_t nq, int64_t nr, int64_t nr_valid, int index_base, float err_rel, float tag_rel,
    const T *__restrict__ q, const T *__restrict__ r,
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
    // (cells from nr_valid on are padding: their norm entries are +inf)
    for (int64_t j = tid; j < nr_valid; j += 64 * kFsWaves) m = fmax(m, rr64[(int64_t)b * nr + j]);
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

}  // namespace
