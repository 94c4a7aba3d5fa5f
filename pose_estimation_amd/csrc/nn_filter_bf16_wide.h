// Dense nearest neighbour, pass 2b': the bfloat16 32x32x16 MFMA filter for 192 to 512 channels (included by nn.hip).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// 2b'. filter pass, bfloat16 MFMA, 192 to 512 channels in multiples of 64, any k
//
// The scheme of nn_filter_bf16_kernel with what more than 256 channels force: the query fragments of a wave's
// 32 queries take C / 4 registers (128 at 512 channels), so a 512-thread workgroup (8 waves of 32 queries, two
// to a SIMD) owns the CU; the query tile goes through LDS in two halves of up to 256 channels (a 256-query tile
// of 512 channels is 288 KiB); a 64-reference chunk image is up to 64 KiB, so the ring holds TWO of them and
// the workgroup meets at the start of every chunk: chunk c + 1 is fetched (LDS-direct) while chunk c is
// contracted, and no fragment is read across a chunk boundary.  Same lists, same partial streams, same
// finalize as the kernel for up to 256 channels.  With 12 and more MFMAs per 32x32 tile the list update is a
// small share of a chunk and one barrier per chunk is cheap: at 192 and 256 channels this kernel is also the
// faster one (top-1 on the 32x32x16 shape included), so those counts come here too.
// ---------------------------------------------------------------------------------------------
template <int KS>
constexpr int bf16_wide_lds_bytes() {
    constexpr int ring = 2 * 16 * KS * kRStride + 2 * kBR * (int)sizeof(float);
    constexpr int qimage = (16 * KS < 256 ? 16 * KS : 256) * kQStride;
    return ring > qimage ? ring : qimage;
}

template <int KS, int K>
__global__ __launch_bounds__(512, 1) void nn_filter_bf16_wide_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ r, const float *__restrict__ rnh, int64_t nq,
    int64_t nr, int n_qtiles, int nsplit, int64_t refs_per_split, float *__restrict__ part_val,
    int *__restrict__ part_idx, float *__restrict__ qnh_out, double *__restrict__ q64_out) {
    constexpr int C = 16 * KS;
    constexpr int kRBuf = C * kRStride;                 // one reference chunk image
    constexpr int kNT = 512;                            // threads
    constexpr int kSeg = KS / 4;                        // 1 KiB pieces of a chunk image per wave
    static_assert(KS % 4 == 0 && KS >= 12 && KS <= 32, "C must be a multiple of 64 from 192 to 512");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *rbuf = lds;                                   // ring of 2 chunk images
    float *nhbuf = reinterpret_cast<float *>(lds + 2 * kRBuf);   // ring of 2 x kBR floats

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

    // ---- query tile -> LDS -> per-wave fragments in registers, 256 channels at a time ---------
    // The query norms on the way (the blocks of reference split 0 write them): thread t < 256 owns column t and
    // carries its float64 sum, in increasing channel order, over the two halves - bit for bit what
    // nn_norms_kernel computes from the map.
    bf16x8 bq[KS];
    double qsum = 0.0;
    static_for<0, 2>([&](auto ih) {
        constexpr int half = decltype(ih)::value;
        constexpr int rows = half == 0 ? (C < 256 ? C : 256) : (C > 256 ? C - 256 : 0);
        if (half) __syncthreads();                       // the first half's image is no longer needed
#pragma unroll
        for (int e = 0; e < rows / 16; ++e) {            // rows x 32 segments of 8 queries
            const int seg = tid + kNT * e;
            const int row = seg >> 5, col = (seg & 31) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q0 + col < nq) v = *reinterpret_cast<const uint4 *>(qb + (int64_t)(256 * half + row) * nq + q0 + col);
            *reinterpret_cast<uint4 *>(lds + row * kQStride + col * 2) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < rows / 16; ++ks)
            bq[16 * half + ks] = tr_frag(lds + (ks * 16 + tr_row) * kQStride + (wq * 32 + tr_col) * 2, kQStride);
        if (qnh_out != nullptr && split == 0 && tid < kBQ) {
#pragma unroll 16
            for (int c = 0; c < rows; ++c) {
                const float v = load_as_float(reinterpret_cast<const uint16_t *>(lds + c * kQStride) + tid);
                qsum = fma((double)v, (double)v, qsum);
            }
        }
    });
    if (qnh_out != nullptr && split == 0 && tid < kBQ && q0 + tid < nq) {
        qnh_out[(int64_t)b * nq + q0 + tid] = (float)(-0.5 * qsum);
        q64_out[(int64_t)b * nq + q0 + tid] = qsum;
    }
    __syncthreads();                                     // query image no longer needed

    // Reference chunk -> LDS slot, 1 KiB (8 rows x 128 B) per wave instruction (see nn_filter_bf16_kernel)
    const int st_seg = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    const int wq_u = __builtin_amdgcn_readfirstlane(wq);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)rbuf;
    const unsigned row_off = (unsigned)((int64_t)(lane >> 3) * nr * 2);        // bytes; nr < 2^27 on this path
    const unsigned lane_off = row_off + st_seg * 16;
    auto stage = [&](int64_t r0, int slot) {
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
            // a ragged last chunk: columns past the end re-read the row start (real, finite values whose scores
            // the masked norm entry pushes out of contention)
            const unsigned voff = row_off + (r0 + st_seg * 8 < rend ? (unsigned)(r0 + st_seg * 8) * 2 : 0);
#pragma unroll
            for (int e = 0; e < kSeg; ++e) {
                const int id = wq_u * kSeg + e;
                const uint64_t sbase = (uint64_t)rb + (uint64_t)(8 * id) * (uint64_t)nr * 2u;
                glds_b128(voff, sbase, __builtin_amdgcn_readfirstlane(lds_base + slot * kRBuf + id * 1024));
            }
        }
    };
    float pre_nh = 0.f;                                  // norm entry of the chunk in flight, parked in a register
    bool pre_ok = false;
    auto fetch_nh = [&](int64_t r0) {
        pre_ok = r0 + (tid & (kBR - 1)) < rend;
        pre_nh = nhb[pre_ok ? r0 + (tid & (kBR - 1)) : 0];
    };
    auto chunk_start = [&](int c) { return rbeg + (int64_t)(c < nchunks ? c : nchunks - 1) * kBR; };
    if (nchunks > 0) {
        stage(chunk_start(0), 0);
        if (wq_u == 0) fetch_nh(chunk_start(0));
    }

    typename BfListFor<K, KS>::type best;
    best.init();
    f32x16 acc_a, acc_b;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc_b[reg] = kNegInf;                        // nothing pending yet
    asm volatile("" : "+v"(acc_b));                      // opaque: keeps the steady-state loop in one piece

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
    // MFMAs of half-step (`rimg`, `ri`) into `now`, top-list update of `prev` (half-step id prev_half), in a
    // pinned program order; the transposed LDS reads run two k-steps ahead of their MFMAs.  a0 / a1 hold the
    // fragments of k-steps 0 and 1 on entry; on exit those of half-step (`rimg`, 1) when `more`, else nothing.
    auto half_step = [&](const char *rimg, int ri, bool more, bf16x8 &a0, bf16x8 &a1, const f32x16 &init,
                         f32x16 &now, const f32x16 &prev, int prev_half) {
        best.begin_half(prev_half);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a2 = a1;
            if (ks + 2 < KS) a2 = frag(rimg, ri, ks + 2);
            else if (more) a2 = frag(rimg, 1, ks + 2 - KS);
            now = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[ks], ks ? now : init, 0, 0, 0);
#pragma unroll
            for (int reg = ks * 16 / KS; reg < (ks + 1) * 16 / KS; ++reg) {
                if constexpr (K == 1) best.push4_step(reg & 3, prev, reg & ~3, reg & ~3);
                else best.push(prev[reg], reg);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1;
            a1 = a2;
        }
        best.end_half(prev);
    };

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int slot = chunk & 1;
        const char *rimg = rbuf + slot * kRBuf;
        if (wq_u == 0) nhbuf[slot * kBR + (tid & (kBR - 1))] = pre_ok ? pre_nh : kNegInf;   // one wave stages the norm entries
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this chunk's image has landed ...
        __syncthreads();                                     // ... for every wave, and the other slot's readers are done
        if (chunk + 1 < nchunks) {
            stage(chunk_start(chunk + 1), slot ^ 1);
            if (wq_u == 0) fetch_nh(chunk_start(chunk + 1));
        }
        bf16x8 a0 = frag(rimg, 0, 0), a1 = frag(rimg, 0, 1);
        const f32x16 init_a = load_init(nhbuf + slot * kBR, 0);
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 0, true, a0, a1, init_a, acc_a, acc_b, chunk * 2 - 1);
        const f32x16 init_b = load_init(nhbuf + slot * kBR, 1);
        __builtin_amdgcn_sched_barrier(0);
        half_step(rimg, 1, false, a0, a1, init_b, acc_b, acc_a, chunk * 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may still be landing in LDS at exit
    if (nchunks > 0) {
        best.begin_half(nchunks * 2 - 1);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) best.push(acc_b[reg], reg);
        best.end_half(acc_b);
        best.flush();
    }

    // one partial stream per (split, lane half); finalize merges them
    const int nparts = nsplit * kBSubStreams;
    const int part = split * kBSubStreams + h;
    // code = half-step * 16 + accumulator register
    auto to_ref = [&](int code) {
        const int reg = code & 15;
        return (int)(rbeg + (int64_t)(code >> 4) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
    };
    store_partial<K + 1>(best, to_ref, part_val, part_idx, b, nparts, part, nq, q0 + wq * 32 + l31);
}

}  // namespace
