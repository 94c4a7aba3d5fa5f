// Dense nearest neighbour, pass 2 / 2a': the float32 MFMA filter kernels (included by nn.hip).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// 2. filter pass, float32 MFMA (v_mfma_f32_32x32x2_f32: exact float32 fma chain over channels)
// ---------------------------------------------------------------------------------------------
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

}  // namespace
