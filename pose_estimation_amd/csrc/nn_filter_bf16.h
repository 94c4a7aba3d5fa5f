// Dense nearest neighbour, pass 2b: the bfloat16 32x32x16 MFMA filter, any k (included by nn.hip).
#pragma once

namespace {

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

// The query norms, from the block's query image in LDS (C rows of `tile` bfloat16 cells, zero beyond the map's
// end): -|q|^2 / 2 as float32 and |q|^2 as float64, the sum in increasing channel order in float64 - bit for bit
// what nn_norms_kernel computes from the map.  The filter kernels that stage their queries through LDS do this
// on the way (the blocks of reference split 0 write), which saves the norms pass a whole read of the query map.
template <int C, int NT>
__device__ __forceinline__ void query_norms_from_image(const char *image, int stride, int tile, int64_t q0, int64_t nq,
                                                       int b, int split, float *__restrict__ qnh_out,
                                                       double *__restrict__ q64_out) {
    if (qnh_out == nullptr || split != 0) return;       // uniform
    for (int col = threadIdx.x; col < tile; col += NT) {
        double s = 0.0;
#pragma unroll 16
        for (int c = 0; c < C; ++c) {
            const float v = load_as_float(reinterpret_cast<const uint16_t *>(image + c * stride) + col);
            s = fma((double)v, (double)v, s);
        }
        if (q0 + col < nq) {
            qnh_out[(int64_t)b * nq + q0 + col] = (float)(-0.5 * s);
            q64_out[(int64_t)b * nq + q0 + col] = s;
        }
    }
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
    int *__restrict__ part_idx, float *__restrict__ qnh_out, double *__restrict__ q64_out) {
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
    query_norms_from_image<C, kNT>(lds, kQStride, kBQ, q0, nq, b, split, qnh_out, q64_out);
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

}  // namespace
