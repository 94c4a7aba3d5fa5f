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
//                 2b' the same with a two-image ring (any k; C = 192, 256, ..., 512)
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
#include "nn_base.h"          // constants, loads, exact_d2, workgroup order
#include "nn_norms.h"         // 1.  norms
#include "nn_lists.h"         //     running top lists of the filter kernels
#include "nn_filter_f32.h"    // 2a  float32 MFMA filters
#include "nn_filter_bf16.h"   // 2b  bfloat16 32x32x16 filter (any k)
#include "nn_filter_bf16_wide.h"   // 2b' the same for 192 to 512 channels in multiples of 64
#include "nn_filter_t16.h"    // 2c  bfloat16 16x16x32 top-1 filter
#include "nn_filter_split.h"  // 2d  split-bfloat16 top-1 filter for float32 inputs
#include "nn_finalize.h"      // 3.  finalize, re-scoring, refinement
#include "nn_exact.h"         // 4.  float64 pass

namespace {

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

// Channel counts of nn_filter_bf16_wide_kernel: multiples of 64 from 192 on (measured against the kernels with a
// three-image ring at 192 and 256 channels: top-1 1.29x / 1.13x the speed, top-5 and top-8 1.02-1.04x)
inline bool bf16_wide_shape(int C) { return C >= 192 && C <= 512 && C % 64 == 0; }

Plan make_plan(int dtype_and_flags, int B, int C, int64_t nq, int64_t nr, int k) {
    Plan p;
    const int dtype = dtype_and_flags & ~ISPA_NN_NORMALIZE;
    p.normalize = (dtype_and_flags & ISPA_NN_NORMALIZE) != 0 && dtype != ISPA_BF16;
    const bool fast_shape = C % 32 == 0 && nq % 8 == 0 && nr % 8 == 0 && nr < (1 << 27);   // 32-bit lane offsets
    p.bf16_mfma = dtype == ISPA_BF16 && fast_shape && (C <= 256 || (C <= 512 && C % 64 == 0));
    p.split = dtype == ISPA_F32_SPLIT && fast_shape && C <= 256 && k == 1;
    p.tq = p.split ? kSQ : (p.bf16_mfma ? kBQ : kTQ);
    const int chunk_refs = (p.bf16_mfma || p.split) ? kBR : kTR;
    p.t16 = p.bf16_mfma && k == 1 && !bf16_wide_shape(C) && !ISPA_TOP1_32X32;   // top-1: the 16x16x32 kernel
    const int substreams =
        (p.t16 || p.split) ? kT16SubStreams : (p.bf16_mfma ? kBSubStreams : kSubStreams);
    p.n_qtiles = (int)((nq + p.tq - 1) / p.tq);
    p.ktemplate = k == 1 ? 1 : (k <= 5 ? 5 : kMaxK);
    const int64_t chunks = (nr + chunk_refs - 1) / chunk_refs;
    // enough workgroups to fill 256 CUs a few times over, without going below 2 chunks per split
    const int64_t want = 1024;
    int64_t nsplit = 1;
    const int64_t base = (int64_t)B * p.n_qtiles;
    // The LDS-staged float32 filter (four workgroups per CU) keeps the plain rule on large calls: about `want`
    // workgroups; small calls weigh the merge cost of the finalize pass like the other filters.
    const bool staged = !p.bf16_mfma && !p.split && !(p.ktemplate == 1 && C % 32 == 0 && C <= 128);
    if (!staged || (int64_t)B * nq <= 524288) {
        // The other filters hold two workgroups per CU up to 128 channels and one above, and every
        // split stream pays a fixed price again: its query fragments and, for top-k, a sorted insert
        // for nearly every score of its first few hundred (measured: what ~800 further references
        // would cost; ~256 for top-1).  So: rounds of workgroups over the chip x (references per
        // split + that price), smallest wins.
        // A split also gives the finalize pass `substreams` more lists to merge per query, one thread per query:
        // on calls too small to fill the chip with those threads that chain of loads is what the pass lasts
        // (measured on one 64x64 pair, top-5, 32 splits: 108 us for 128 lists, about what 32 further references
        // per list cost the filter; 6 for the two-entry lists of top-1).
        const int64_t slots = staged ? 1024 : 256 * (C > 128 ? 1 : 2);
        const int64_t warm = p.ktemplate > 1 ? 800 : 256;
        const int64_t merge = (int64_t)B * nq <= 524288 ? substreams * (p.ktemplate > 1 ? 32 : 6) : 0;
        int64_t best_cost = -1;
        for (int64_t n = 1; n == 1 || chunks / n >= 2; n *= 2) {
            const int64_t cost =
                ((base * n + slots - 1) / slots) * ((chunks + n - 1) / n * chunk_refs + warm) + n * merge;
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
                int64_t nr, const Plan &p, float *pval, int *pidx, float *qnh_out, double *q64_out) {
    constexpr int lds_bytes = bf16_filter_lds_bytes<KS>();
    // more than 128 channels (one workgroup per CU): 8 waves of 32 queries where the chunk pieces divide evenly
    constexpr int QT = (KS > 8 && KS % 4 == 0) ? 1 : 2;
    static unsigned long long configured = 0;       // per instantiation, one bit per device
    if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_filter_bf16_kernel<KS, KT, QT>), lds_bytes,
                                         &configured))
        return rc;
    hipLaunchKernelGGL((nn_filter_bf16_kernel<KS, KT, QT>), grid, dim3(512 / QT), lds_bytes, s, q, r, rnh, nq, nr,
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx, qnh_out, q64_out);
    return ISPA_OK;
}

template <int KS, int KT>
int launch_bf16_wide(dim3 grid, hipStream_t s, const uint16_t *q, const uint16_t *r, const float *rnh, int64_t nq,
                     int64_t nr, const Plan &p, float *pval, int *pidx, float *qnh_out, double *q64_out) {
    constexpr int lds_bytes = bf16_wide_lds_bytes<KS>();
    static unsigned long long configured = 0;       // per instantiation, one bit per device
    if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_filter_bf16_wide_kernel<KS, KT>), lds_bytes,
                                         &configured))
        return rc;
    hipLaunchKernelGGL((nn_filter_bf16_wide_kernel<KS, KT>), grid, dim3(512), lds_bytes, s, q, r, rnh, nq, nr,
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx, qnh_out, q64_out);
    return ISPA_OK;
}
template <int KT>
int launch_bf16_wide_ks(int ks, dim3 grid, hipStream_t s, const uint16_t *q, const uint16_t *r, const float *rnh,
                        int64_t nq, int64_t nr, const Plan &p, float *pval, int *pidx, float *qnh_out,
                        double *q64_out) {
    switch (ks) {
        case 12: return launch_bf16_wide<12, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 16: return launch_bf16_wide<16, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 20: return launch_bf16_wide<20, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 24: return launch_bf16_wide<24, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 28: return launch_bf16_wide<28, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        default: return launch_bf16_wide<32, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
    }
}

#ifndef ISPA_T16_NQ
#define ISPA_T16_NQ 4        // development switch: 2 = the 8-wave form of the top-1 kernel for 64 / 128 channels
#endif
template <int KS32>
int launch_t16(dim3 grid, hipStream_t s, const uint16_t *q, const uint16_t *r, const float *rnh, int64_t nq,
               int64_t nr, const Plan &p, float *pval, int *pidx, float *qnh_out, double *q64_out) {
    constexpr int lds_bytes = bf16_filter_lds_bytes<2 * KS32>();
    // 4 waves of 64 queries; the 8-wave form (64 or 128 channels) is slower: 1.88 against 1.77 ms per 512 pairs
    constexpr int NQ = (KS32 % 2 == 0 && KS32 <= 4) ? ISPA_T16_NQ : 4;
    static unsigned long long configured = 0;       // per instantiation, one bit per device
    if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(&nn_filter_bf16_t16_kernel<KS32, NQ>), lds_bytes,
                                         &configured))
        return rc;
    hipLaunchKernelGGL((nn_filter_bf16_t16_kernel<KS32, NQ>), grid, dim3(1024 / NQ), lds_bytes, s, q, r, rnh, nq, nr,
                       p.n_qtiles, p.nsplit, p.refs_per_split, pval, pidx, qnh_out, q64_out);
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
                   int64_t nq, int64_t nr, const Plan &p, float *pval, int *pidx, float *qnh_out, double *q64_out) {
    switch (ks) {
        case 2: return launch_bf16<2, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 4: return launch_bf16<4, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 6: return launch_bf16<6, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 8: return launch_bf16<8, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 10: return launch_bf16<10, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 12: return launch_bf16<12, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        case 14: return launch_bf16<14, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
        default: return launch_bf16<16, KT>(grid, s, q, r, rnh, nq, nr, p, pval, pidx, qnh_out, q64_out);
    }
}

// Reference cells [nr_valid, nr) of every pair are padding (see nn_topk_checked): unselectable for the filters
// (their accumulators start at the score of a masked cell) and at distance +inf for the float64 passes.
__global__ void nn_mask_padding_kernel(float *__restrict__ rnh, double *__restrict__ r64, int B, int64_t nr,
                                       int64_t nr_valid) {
    const int64_t npad = nr - nr_valid;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * npad) return;
    const int64_t o = (t / npad) * nr + nr_valid + t % npad;
    rnh[o] = kNegInf;
    r64[o] = __builtin_inf();
}

template <typename T>
int run(const T *q, const T *r, int B, int C, int64_t nq, int64_t nr, int64_t nr_valid, int k, int index_base,
        float *dist, int32_t *idx, char *ws, const Plan &p, int32_t *stats, hipStream_t s) {
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
    // The bfloat16 MFMA filters stage their query tile through LDS and compute the query norms there (first needed by
    // the finalize pass): the norms pass then reads the reference map only.
    const bool filter_qnorms = p.bf16_mfma && !small;
    float *fq_nh = filter_qnorms ? qnh : nullptr;
    double *fq_64 = filter_qnorms ? q64 : nullptr;
    (void)fq_nh;
    (void)fq_64;
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
        if (filter_qnorms) {
            if (r4)
                hipLaunchKernelGGL((nn_norms4_kernel<T>), dim3(ispa::blocks_for(nr / 4, 256), B), dim3(256), 0, s, r, C,
                                   nr, rnh, r64, rmax);
            else
                hipLaunchKernelGGL((nn_norms_kernel<T>), dim3(ispa::blocks_for(nr, 256), B), dim3(256), 0, s, r, C, nr,
                                   rnh, r64, rmax);
        } else if (q4 && r4) {
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
    if (nr_valid < nr)
        hipLaunchKernelGGL(nn_mask_padding_kernel, dim3(ispa::blocks_for((int64_t)B * (nr - nr_valid), 256)), dim3(256), 0,
                           s, rnh, r64, B, nr, nr_valid);

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
        if (bf16_wide_shape(C)) {
            if (p.ktemplate == 1) rc = launch_bf16_wide_ks<1>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64);
            else if (p.ktemplate == 5) rc = launch_bf16_wide_ks<5>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64);
            else rc = launch_bf16_wide_ks<kMaxK>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64);
        } else if (p.t16) {
            switch (C / 32) {
                case 1: rc = launch_t16<1>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                case 2: rc = launch_t16<2>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                case 3: rc = launch_t16<3>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                case 4: rc = launch_t16<4>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                case 5: rc = launch_t16<5>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                case 6: rc = launch_t16<6>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                case 7: rc = launch_t16<7>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
                default: rc = launch_t16<8>(grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64); break;
            }
        }
#if ISPA_TOP1_32X32
        else if (p.ktemplate == 1) rc = launch_bf16_ks<1>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64);
#endif
        else if (p.ktemplate == 5) rc = launch_bf16_ks<5>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64);
        else rc = launch_bf16_ks<kMaxK>(C / 16, grid, s, qh, rh, rnh, nq, nr, p, pval, pidx, fq_nh, fq_64);
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
                           pval, pidx, p.nparts, qnh, C, nq, nr, nr_valid, index_base, err_rel, tag_rel, q, r, q64, r64, dist, idx,
                           count, list, pcount, plist, count + 1);
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
    // their reference map among kQB queries; everything else goes one query per block.  Queues get long
    // behind the split filter (its bound is wider) and, with any filter, on maps with exactly tied cells
    // (masked or padded regions: every query whose nearest cell is one of them has to be settled here);
    // otherwise the launch finds nothing to do and ends at once.
    const bool batched = k == 1 && !small && nr % 4 == 0 && (uintptr_t)r % 16 == 0 &&
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

// Shapes off the grid of the fast kernels (channels a multiple of 32, cell counts multiples of 8: 16-byte rows)
// are brought onto it inside the call, in the workspace: zero channels (no product, norm or float64 sum changes),
// copies of the last query column (queries are independent; the copies' results are dropped) and copies of the
// last reference column that nn_mask_padding_kernel makes unselectable for any k.  ISPA_NN_NO_PAD hands the
// arrays to the kernels as they are (guarded loaders).
struct PadPlan {
    bool on;
    int Cp;
    int64_t nqp, nrp;
    bool copy_q, copy_r, tmp_out;
    size_t inner, off_q, off_r, off_dist, off_idx, total;
};

PadPlan make_pad_plan(int dtype_and_flags, int B, int C, int64_t nq, int64_t nr, int k) {
    PadPlan pp;
    const int dtype = dtype_and_flags & ~(ISPA_NN_NORMALIZE | ISPA_NN_NO_PAD);
    const int inner_flags = dtype_and_flags & ISPA_NN_NORMALIZE;
    const size_t elem = dtype == ISPA_BF16 ? 2 : 4;
    pp.Cp = (C + 31) / 32 * 32;
    // the bfloat16 kernels take multiples of 64 channels above 256
    if (dtype == ISPA_BF16 && pp.Cp > 256) pp.Cp = (C + 63) / 64 * 64;
    pp.nqp = (nq + 7) / 8 * 8;
    pp.nrp = (nr + 7) / 8 * 8;
    // bfloat16 beyond 512 channels runs through the converting float32 loader whatever the shape
    const bool reachable = dtype == ISPA_BF16 ? pp.Cp <= 512 : true;
    pp.on = !(dtype_and_flags & ISPA_NN_NO_PAD) && reachable && nr >= k && pp.nrp < (1 << 27) &&
            (pp.Cp != C || pp.nqp != nq || pp.nrp != nr);
    if (!pp.on) {
        pp.Cp = C;
        pp.nqp = nq;
        pp.nrp = nr;
    }
    pp.copy_q = pp.on && (pp.Cp != C || pp.nqp != nq);
    pp.copy_r = pp.on && (pp.Cp != C || pp.nrp != nr);
    pp.tmp_out = pp.on && pp.nqp != nq;
    int base = dtype == ISPA_F32_SPLIT_ALWAYS ? ISPA_F32_SPLIT : dtype;
    const size_t a = make_plan(base | inner_flags, B, pp.Cp, pp.nqp, pp.nrp, k).total;
    const size_t b = make_plan(ISPA_F32 | inner_flags, B, pp.Cp, pp.nqp, pp.nrp, k).total;   // the plan a call may fall back to
    pp.inner = align_up(a > b ? a : b, 256);
    size_t off = pp.inner;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    pp.off_q = pp.copy_q ? take(elem * (size_t)B * pp.Cp * pp.nqp) : 0;
    pp.off_r = pp.copy_r ? take(elem * (size_t)B * pp.Cp * pp.nrp) : 0;
    pp.off_dist = pp.tmp_out ? take(sizeof(float) * (size_t)B * k * pp.nqp) : 0;
    pp.off_idx = pp.tmp_out ? take(sizeof(int32_t) * (size_t)B * k * pp.nqp) : 0;
    pp.total = off;
    return pp;
}

// (B, C, n) -> (B, Cp, np): channels beyond C are zero, columns beyond n repeat column n - 1.  np is a multiple
// of 8 and the output rows are 16-byte aligned: a thread builds 8 consecutive output cells (scalar loads: the
// source rows are not aligned) and stores them as 16-byte vectors.
template <typename T>
__global__ __launch_bounds__(256) void nn_pad_map_kernel(const T *__restrict__ in, int C, int64_t n, T *__restrict__ out,
                                                         int Cp, int64_t np) {
    const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int c = blockIdx.y, b = blockIdx.z;
    if (j0 >= np) return;
    T v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = T(0);
    if (c < C) {
        const T *row = in + ((int64_t)b * C + c) * n;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = row[j0 + e < n ? j0 + e : n - 1];
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + ((int64_t)b * Cp + c) * np + j0);
    const uint4 *src = reinterpret_cast<const uint4 *>(v);
#pragma unroll
    for (int e = 0; e < (int)(8 * sizeof(T) / 16); ++e) dst[e] = src[e];
}
// (rows, np) -> (rows, n): the results of the real query columns
__global__ void nn_unpad_rows_kernel(const float *__restrict__ dist_p, const int32_t *__restrict__ idx_p, int64_t rows,
                                     int64_t np, int64_t n, float *__restrict__ dist, int32_t *__restrict__ idx) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = blockIdx.y;
    if (j >= n) return;
    dist[row * n + j] = dist_p[row * np + j];
    idx[row * n + j] = idx_p[row * np + j];
}
}  // namespace

extern "C" size_t ispa_nn_workspace_bytes_for(int dtype, int B, int C, int64_t nq, int64_t nr, int k) {
    if (B <= 0 || C <= 0 || nq <= 0 || nr <= 0 || k < 1 || k > kMaxK) return 0;
    int base = dtype & ~(ISPA_NN_NORMALIZE | ISPA_NN_NO_PAD);
    const int flag = dtype & ISPA_NN_NORMALIZE;
    if (base != ISPA_F32 && base != ISPA_BF16 && base != ISPA_F32_SPLIT && base != ISPA_F32_SPLIT_ALWAYS) return 0;
    if (flag && base == ISPA_BF16) return 0;
    return make_pad_plan(dtype, B, C, nq, nr, k).total;
}

extern "C" size_t ispa_nn_workspace_bytes(int B, int C, int64_t nq, int64_t nr, int k) {
    if (B <= 0 || C <= 0 || nq <= 0 || nr <= 0 || k < 1 || k > kMaxK) return 0;
    // the larger of the two dtype plans, so that one workspace serves both
    const size_t a = make_pad_plan(ISPA_F32, B, C, nq, nr, k).total;
    const size_t b = make_pad_plan(ISPA_BF16, B, C, nq, nr, k).total;
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
// One call on arrays the kernels can take as they are; reference cells [nr_valid, nr) are padding.
int nn_topk_inner(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr, int64_t nr_valid,
                  int k, int index_base, float *dist, int32_t *idx, void *workspace, size_t workspace_bytes,
                  int32_t *stats, void *stream, const char *workspace_name) {
    const int flags = dtype & ISPA_NN_NORMALIZE;
    dtype &= ~ISPA_NN_NORMALIZE;
    const bool split_always = dtype == ISPA_F32_SPLIT_ALWAYS;
    if (split_always) dtype = ISPA_F32_SPLIT;
    Plan p = make_plan(dtype | flags, B, C, nq, nr, k);
    if (p.bf16_mfma && (((uintptr_t)q | (uintptr_t)r) % 16 != 0)) {
        p = make_plan(ISPA_F32, B, C, nq, nr, k);       // unaligned bf16 input: converting tile loader
        p.bf16_mfma = false;
        p.t16 = false;
        p.split = false;
    }
    // The split filter pays for itself from about 65536 queries per call: its wider margins queue about 1 % of
    // the queries for the float64 pass, and a short queue of long sweeps (one 128x128 pair: 140 of them) costs
    // more than the filter saves; its extra passes (hi / lo images) are nine launches.  Smaller calls get the
    // same results sooner from the float32 filter and its few-launch path for small problems.
    if (p.split && !split_always && (int64_t)B * nq < 65536) {
        dtype = ISPA_F32;
        p = make_plan(ISPA_F32 | flags, B, C, nq, nr, k);
    }
    if (workspace_bytes < p.total)
        return ispa::fail(ISPA_EWORKSPACE, "%s of %zu bytes given, %zu needed", workspace_name, workspace_bytes, p.total);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ISPA_F32 || dtype == ISPA_F32_SPLIT)
        return run((const float *)q, (const float *)r, B, C, nq, nr, nr_valid, k, index_base, dist, idx,
                   (char *)workspace, p, stats, s);
    return run((const uint16_t *)q, (const uint16_t *)r, B, C, nq, nr, nr_valid, k, index_base, dist, idx,
               (char *)workspace, p, stats, s);
}

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
    const int base = dtype & ~(ISPA_NN_NORMALIZE | ISPA_NN_NO_PAD);
    ISPA_REQUIRE(base == ISPA_F32 || base == ISPA_BF16 || base == ISPA_F32_SPLIT || base == ISPA_F32_SPLIT_ALWAYS,
                 "unknown dtype %d", dtype);
    ISPA_REQUIRE(!((dtype & ISPA_NN_NORMALIZE) && base == ISPA_BF16), "ISPA_NN_NORMALIZE takes float32 features");
    ISPA_REQUIRE((uintptr_t)workspace % 256 == 0, "%s must be 256-byte aligned", workspace_name);
    const PadPlan pp = make_pad_plan(dtype, B, C, nq, nr, k);
    const int inner_dtype = dtype & ~ISPA_NN_NO_PAD;
    if (!pp.on)
        return nn_topk_inner(q, r, inner_dtype, B, C, nq, nr, nr, k, index_base, dist, idx, workspace, workspace_bytes,
                             stats, stream, workspace_name);
    if (workspace_bytes < pp.total)
        return ispa::fail(ISPA_EWORKSPACE, "%s of %zu bytes given, %zu needed", workspace_name, workspace_bytes, pp.total);
    ISPA_REQUIRE((int64_t)B * pp.nqp < (1ll << 31) && pp.Cp <= 65535, "problem too large");
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const void *qi = q, *ri = r;
    auto pad = [&](const void *in, int64_t n, void *out, int64_t np) {
        const dim3 grid(ispa::blocks_for(np / 8, 256), pp.Cp, B);
        if (base == ISPA_BF16)
            hipLaunchKernelGGL((nn_pad_map_kernel<uint16_t>), grid, dim3(256), 0, s, (const uint16_t *)in, C, n,
                               (uint16_t *)out, pp.Cp, np);
        else
            hipLaunchKernelGGL((nn_pad_map_kernel<float>), grid, dim3(256), 0, s, (const float *)in, C, n, (float *)out,
                               pp.Cp, np);
    };
    if (pp.copy_q) {
        pad(q, nq, ws + pp.off_q, pp.nqp);
        qi = ws + pp.off_q;
    }
    if (pp.copy_r) {
        pad(r, nr, ws + pp.off_r, pp.nrp);
        ri = ws + pp.off_r;
    }
    if (int rc = ispa::check_launch("nn_pad_map_kernel")) return rc;
    float *dist_i = pp.tmp_out ? (float *)(ws + pp.off_dist) : dist;
    int32_t *idx_i = pp.tmp_out ? (int32_t *)(ws + pp.off_idx) : idx;
    if (int rc = nn_topk_inner(qi, ri, inner_dtype, B, pp.Cp, pp.nqp, pp.nrp, nr, k, index_base, dist_i, idx_i, ws,
                               pp.inner, stats, stream, workspace_name))
        return rc;
    if (pp.tmp_out) {
        hipLaunchKernelGGL(nn_unpad_rows_kernel, dim3(ispa::blocks_for(nq, 256), (unsigned)(B * k)), dim3(256), 0, s,
                           dist_i, idx_i, (int64_t)B * k, pp.nqp, nq, dist, idx);
        if (int rc = ispa::check_launch("nn_unpad_rows_kernel")) return rc;
    }
    return ISPA_OK;
}
}  // namespace
