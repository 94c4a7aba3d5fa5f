// Dense nearest neighbour, pass 2c: the bfloat16 16x16x32 MFMA top-1 filter (included by nn.hip).
#pragma once

namespace {

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
    int *__restrict__ part_idx, float *__restrict__ qnh_out, double *__restrict__ q64_out) {
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
    query_norms_from_image<C, kNT>(lds, kQStride, kBQ, q0, nq, b, split, qnh_out, q64_out);
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
    // One chunk, in ring slot `slot`.  Up to 192 channels the slot is a compile-time constant: the loop below is
    // unrolled over the three slots, so every LDS address of the body is a loop-invariant lane register plus an
    // immediate offset, and the staging's m0 values are constants - no per-chunk address arithmetic.  Above that
    // the body is too long to be laid out five times (the compiler then indexes the fragment arrays through
    // scratch memory, a hundred times slower) and the slot stays a loop variable.
    auto chunk_body = [&](int chunk, auto slot_c) {
        const int slot = slot_c;
        const int next_slot = (slot + 1) % 3, after_slot = (slot + 2) % 3;
        const char *rimg = rbuf + slot * kRBuf;
        const char *next_img = rbuf + next_slot * kRBuf;
        const f32x4 init_b[2] = {load_init(nhbuf + slot * kBR, 1, 0), load_init(nhbuf + slot * kBR, 1, 1)};
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 0, rimg, 1, a, init_a, acc_a, acc_b, chunk - 1, 1);
        if (wq_u == 0) nhbuf[next_slot * kBR + (tid & (kBR - 1))] = pre_ok ? pre_nh : kNegInf;
        if (ISPA_ABLATE != 7 && ISPA_ABLATE != 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ISPA_ABLATE != 8 && ISPA_ABLATE != 9) __syncthreads();
        if (ISPA_ABLATE != 2 && ISPA_ABLATE != 123) stage(clamp_chunk(chunk + 2), after_slot);
        if (wq_u == 0) fetch_nh(clamp_chunk(chunk + 2));
        init_a[0] = load_init(nhbuf + next_slot * kBR, 0, 0);
        init_a[1] = load_init(nhbuf + next_slot * kBR, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 1, next_img, 0, a, init_b, acc_b, acc_a, chunk, 0);
    };
    if constexpr (KS32 <= 6) {
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
    } else {
        int slot = 0;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            chunk_body(chunk, slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
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

}  // namespace
