// Dense nearest neighbour, pass 2d: float32 inputs through a split-bfloat16 top-1 filter (included by nn.hip).
#pragma once

namespace {

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

    // one chunk in ring slot `slot`; up to 192 channels a compile-time constant (the loop is unrolled over the
    // two slots: LDS addresses are loop-invariant lane registers plus immediates, the staging's m0 values
    // constants), above that a loop variable (the body laid out three times sends the fragment arrays to scratch)
    auto chunk_body = [&](int chunk, auto slot_c) {
        const int slot = slot_c;
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
    if constexpr (KS32 <= 6) {
        int chunk = 0;
        for (; chunk + 2 <= nchunks; chunk += 2) {
            chunk_body(chunk, std::integral_constant<int, 0>{});
            chunk_body(chunk + 1, std::integral_constant<int, 1>{});
        }
        if (chunk < nchunks) chunk_body(chunk, std::integral_constant<int, 0>{});
    } else {
        for (int chunk = 0; chunk < nchunks; ++chunk) chunk_body(chunk, chunk & 1);
    }
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

}  // namespace
