// Splat writers: proxy-optical-flow map, point depth map, packed z-buffer.
//
// The reference writes blocks sequentially in point order (make_disparity_syn.py:65-68,
// io_utils.py:119-124), so each pixel ends up with the value of the LAST point covering it.
// Here that is an integer atomicMax of (point index + 1) per pixel followed by a resolve pass,
// which is order-independent and therefore deterministic.
#include "common.h"
#include "geom_device.h"

namespace {

constexpr int kThreads = 256;

// Bounds of the Python slice [c-r : c+r] on an axis of length `size` (negative values wrap once).
__device__ __forceinline__ void py_slice(int c, int r, int size, int &lo, int &hi) {
    lo = c - r;
    hi = c + r;
    if (lo < 0) {
        lo += size;
        if (lo < 0) lo = 0;
    } else if (lo > size) {
        lo = size;
    }
    if (hi < 0) {
        hi += size;
        if (hi < 0) hi = 0;
    } else if (hi > size) {
        hi = size;
    }
    if (hi < lo) hi = lo;
}

// astype(int) / int() of a float64 that is finite; out-of-int-range values are rejected.
__device__ __forceinline__ bool to_int(double x, int &out) {
    const double t = trunc(x);
    if (!(t > -1.0e9 && t < 1.0e9)) return false;
    out = (int)t;
    return true;
}

// A point's block [c-r : c+r] x [c'-r : c'+r] is a (2r x 2r) window that may be cut at either end of
// the axis.  With side >= 2r every non-empty Python slice [lo, hi) is such a window anchored at
// `anchor` = lo when it is cut at the far end (hi == size), hi - 2r otherwise (possibly negative: cut
// at 0), i.e. it covers [anchor, anchor + 2r) intersected with [0, size).  So "highest point index
// covering a pixel" = maximum over the 2r x 2r anchors up and left of the pixel: ONE atomic per point
// on an anchor image, then a window maximum per pixel, instead of (2r)^2 atomics per point.
__device__ __forceinline__ bool slice_anchor(int c, int r, int size, int &anchor) {
    int lo, hi;
    py_slice(c, r, size, lo, hi);
    if (lo >= hi) return false;
    anchor = hi == size ? lo : hi - 2 * r;
    return true;
}
// window maximum for pixel (ly, x) of a band whose anchor image `img` has rows starting 2r-1 anchors
// above the band's first row and `pw` = side + 2r - 1 columns starting 2r-1 anchors left of column 0
__device__ __forceinline__ int window_owner(const int *img, int pw, int ly, int x, int w) {
    int o = 0;
    for (int dr = 0; dr < w; ++dr) {
        const int *row = img + (ly + w - 1 - dr) * pw + x + w - 1;
        for (int dc = 0; dc < w; ++dc) o = max(o, row[-dc]);
    }
    return o;
}

// spread 2 (the reference's 4 x 4 blocks): all 16 anchors read before the first comparison
__device__ __forceinline__ int window_owner4(const int *img, int pw, int ly, int x) {
    int t[16];
#pragma unroll
    for (int dr = 0; dr < 4; ++dr)
#pragma unroll
        for (int dc = 0; dc < 4; ++dc) t[4 * dr + dc] = img[(ly + 3 - dr) * pw + x + 3 - dc];
    int o = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) o = max(o, t[e]);
    return o;
}

__global__ __launch_bounds__(kThreads) void flow_owner_kernel(const double *__restrict__ uv1,
                                                              const uint8_t *__restrict__ truth, int64_t n,
                                                              int spread, int h, int w,
                                                              int32_t *__restrict__ owner) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int m = blockIdx.y;
    if (i >= n) return;
    if (!truth[(int64_t)m * n + i]) return;
    const double *r = uv1 + (int64_t)m * 3 * n;
    int u, v;
    if (!to_int(r[i], u) || !to_int(r[n + i], v)) return;
    int y0, y1, x0, x1;
    py_slice(v, spread, h, y0, y1);
    py_slice(u, spread, w, x0, x1);
    int32_t *o = owner + (int64_t)m * h * w;
    const int32_t tag = (int32_t)i + 1;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) atomicMax(&o[(int64_t)y * w + x], tag);
}

__global__ __launch_bounds__(kThreads) void flow_resolve_kernel(const double *__restrict__ uv1,
                                                                const double *__restrict__ uv2, int64_t n,
                                                                int64_t npix, const int32_t *__restrict__ owner,
                                                                double *__restrict__ flow) {
    const int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int m = blockIdx.y;
    if (pix >= npix) return;
    const int32_t o = owner[(int64_t)m * npix + pix];
    double2 val = make_double2(0.0, 0.0);
    if (o > 0) {
        const double *a = uv1 + (int64_t)m * 3 * n;
        const double *b = uv2 + (int64_t)m * 3 * n;
        const int64_t i = o - 1;
        val.x = b[i] - a[i];
        val.y = b[n + i] - a[n + i];
    }
    reinterpret_cast<double2 *>(flow)[(int64_t)m * npix + pix] = val;
}

// Flow splat with the owner image of one horizontal band of one map in LDS: every block scans all
// points of its map (coalesced), keeps per pixel the highest kept point index with LDS atomics, then
// resolves its band and writes the flow rows.  No global atomics, no scratch image in HBM.
constexpr int kBandPixels = 25088;   // 98 KiB owner band (half of a 224 x 224 map): one block per CU
constexpr int kBandThreads = 1024;
#ifndef DISP_BATCH
#define DISP_BATCH 4
#endif
constexpr int kMaxDilateSpread = 4;          // window maximum over at most 8 x 8 anchors per pixel
constexpr int kMaxBandLds = 150 * 1024;      // bytes of dynamic LDS a band kernel may ask for

template <bool DILATE>
__global__ __launch_bounds__(kBandThreads) void flow_band_kernel(const double *__restrict__ uv1,
                                                                 const double *__restrict__ uv2,
                                                                 const uint8_t *__restrict__ truth, int64_t n,
                                                                 int spread, int side, int band_rows,
                                                                 double *__restrict__ flow) {
    extern __shared__ int owner[];
    const int m = blockIdx.y;
    const int row0 = blockIdx.x * band_rows;
    const int row1 = min(row0 + band_rows, side);
    const int npix = (row1 - row0) * side;
    const int w = 2 * spread, pw = side + w - 1;         // DILATE: anchor image (band rows + w - 1) x pw
    const int nimg = DILATE ? (row1 - row0 + w - 1) * pw : npix;
    for (int p = threadIdx.x; p < nimg; p += kBandThreads) owner[p] = 0;
    __syncthreads();
    const double *a = uv1 + (int64_t)m * 3 * n;
    const double *b = uv2 + (int64_t)m * 3 * n;
    const uint8_t *t = truth + (int64_t)m * n;
    // batches of 8 points per thread: all flag loads first, then all coordinate loads, then the
    // LDS updates, so that many loads are in flight per wave
    constexpr int kBatch = 8;
    for (int64_t base = 0; base < n; base += (int64_t)kBandThreads * kBatch) {
        bool keep[kBatch];
        double pu[kBatch], pv[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int64_t i = base + (int64_t)e * kBandThreads + threadIdx.x;
            keep[e] = i < n && t[i < n ? i : 0] != 0;
        }
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int64_t i = base + (int64_t)e * kBandThreads + threadIdx.x;
            pu[e] = keep[e] ? a[i] : 0.0;
            pv[e] = keep[e] ? a[n + i] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            if (!keep[e]) continue;
            int u, v;
            if (!to_int(pu[e], u) || !to_int(pv[e], v)) continue;
            const int tag = (int)(base + (int64_t)e * kBandThreads + threadIdx.x) + 1;
            if (DILATE) {
                int ry, cx;
                if (!slice_anchor(v, spread, side, ry) || !slice_anchor(u, spread, side, cx)) continue;
                const int lr = ry - (row0 - (w - 1));
                if (lr < 0 || ry >= row1) continue;
                atomicMax(&owner[lr * pw + cx + w - 1], tag);
            } else {
                int y0, y1, x0, x1;
                py_slice(v, spread, side, y0, y1);
                y0 = max(y0, row0);
                y1 = min(y1, row1);
                if (y0 >= y1) continue;
                py_slice(u, spread, side, x0, x1);
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) atomicMax(&owner[(y - row0) * side + x], tag);
            }
        }
    }
    __syncthreads();
    double2 *out = reinterpret_cast<double2 *>(flow) + ((int64_t)m * side + row0) * side;
    for (int p = threadIdx.x; p < npix; p += kBandThreads) {
        const int o = DILATE ? (w == 4 ? window_owner4(owner, pw, p / side, p % side) : window_owner(owner, pw, p / side, p % side, w))
                             : owner[p];
        double2 val = make_double2(0.0, 0.0);
        if (o > 0) {
            const int64_t i = o - 1;
            val.x = b[i] - a[i];
            val.y = b[n + i] - a[n + i];
        }
        out[p] = val;
    }
}

// make_disparity in one launch (make_disparity_syn.py:44-68): per (map, half-map band) block,
// project every point at pose 1, depth-test it against the map's depth image, crop-remap it, and
// keep per band pixel the highest visible point index (LDS atomicMax); then, per owned pixel,
// re-project the owning point at both poses and write the flow.  Nothing but the flow map goes to
// HBM: the projected coordinates of the un-fused path (49 bytes per point and pose) never exist.
template <bool DILATE>
__global__ __launch_bounds__(kBandThreads) void make_disparity_kernel(
    const double *__restrict__ pts, int64_t n, const double *__restrict__ cams1, const double *__restrict__ cams2,
    const float *__restrict__ depth, int depth_per_map, int H, int W, int ylim, int xlim, double relax,
    const int32_t *__restrict__ crop, int crop_per_map, int spread, int band_rows, double *__restrict__ flow) {
    using namespace ispa_geom;
    extern __shared__ int owner[];
    constexpr int side = (int)kCropSide;
    const int m = blockIdx.y;
    const int row0 = blockIdx.x * band_rows;
    const int row1 = min(row0 + band_rows, side);
    const int npix = (row1 - row0) * side;
    const int w = 2 * spread, pw = side + w - 1;         // DILATE: anchor image (band rows + w - 1) x pw
    const int nimg = DILATE ? (row1 - row0 + w - 1) * pw : npix;
    for (int p = threadIdx.x; p < nimg; p += kBandThreads) owner[p] = 0;
    __syncthreads();
    const double *P1 = cams1 + 12 * (int64_t)m;
    const double *P2 = cams2 + 12 * (int64_t)m;
    const float *d = depth + (depth_per_map ? (int64_t)m * H * W : 0);
    const int32_t *c = crop + (crop_per_map ? 4 * (int64_t)m : 0);
    const CropXform cx = make_crop(c);
    const int tall = (c[1] - c[0]) >= (c[3] - c[2]);
    // Batches of kBatch points per thread in three phases, so that the point loads of a batch and then its
    // depth gathers are all in flight together: (1) project, crop, band membership -> texel index (0 for a
    // point that cannot write into this band, whose texel is discarded); (2) the gathers, unconditional;
    // (3) visibility and the LDS update.  Only a point that can write into this band is kept, so each
    // point is depth-tested by one band's block.
    constexpr int kBatch = DISP_BATCH;
    for (int64_t base = 0; base < n; base += (int64_t)kBandThreads * kBatch) {
        double px[kBatch], py[kBatch], pz[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int64_t i = base + (int64_t)e * kBandThreads + threadIdx.x;
            const int64_t j = i < n ? i : 0;
            px[e] = pts[3 * j];
            py[e] = pts[3 * j + 1];
            pz[e] = pts[3 * j + 2];
        }
        bool cand[kBatch];
        int64_t texel_at[kBatch];
        int iu_[kBatch], iv_[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int64_t i = base + (int64_t)e * kBandThreads + threadIdx.x;
            double u, v, z;
            project_point(P1, px[e], py[e], pz[e], u, v, z);
            pz[e] = z;
            double uc = u, vc = v;
            bool ok = i < n && crop_apply(cx, tall, 1, uc, vc);
            // A point inside the crop has remapped coordinates in [0, side] (strictly inside the box, times a
            // positive ratio, plus a non-negative pad), so the conversions cannot overflow, the Python slice
            // [c - r : c + r] cannot wrap from above, and slice_anchor() reduces to: empty for c < r (the lower
            // bound wraps past the upper one), anchor c - r otherwise.  Lanes with ok == false carry garbage
            // that is never used (cvt_i32_sat is defined for every input).
            const int iu = cvt_i32_sat(uc), iv = cvt_i32_sat(vc);
            if (DILATE) {
                ok = ok && iu >= spread && iv >= spread && iu < side + spread && iv < side + spread;
                const int lr = iv - spread - (row0 - (w - 1));
                ok = ok && lr >= 0 && iv - spread < row1;
                iv_[e] = lr * pw + iu - spread + w - 1;
            } else {
                int y0, y1;
                py_slice(iv, spread, side, y0, y1);
                ok = ok && max(y0, row0) < min(y1, row1);
                iu_[e] = iu;
                iv_[e] = iv;
            }
            // depth gate (pt2render.py:47-49) on the rounded coordinates as integers: u and v are integer-valued
            // (or not finite / huge, which saturate or become 0 and fail the gate like the float comparison)
            const int gx = cvt_i32_sat(u), gy = cvt_i32_sat(v);
            const bool gate = gy > 0 && gy < ylim && gx > 0 && gx < xlim;
            const int64_t at = gate ? (int64_t)gy * W + gx : 0;
            cand[e] = ok && gate;
            texel_at[e] = cand[e] ? at : 0;
        }
        float texel[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) texel[e] = d[texel_at[e]];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            if (!depth_decide_a(cand[e], pz[e], texel[e], relax)) continue;
            const int tag = (int)(base + (int64_t)e * kBandThreads + threadIdx.x) + 1;
            if (DILATE) {
                atomicMax(&owner[iv_[e]], tag);
            } else {
                int y0, y1, x0, x1;
                py_slice(iv_[e], spread, side, y0, y1);
                y0 = max(y0, row0);
                y1 = min(y1, row1);
                py_slice(iu_[e], spread, side, x0, x1);
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) atomicMax(&owner[(y - row0) * side + x], tag);
            }
        }
    }
    __syncthreads();
    double2 *out = reinterpret_cast<double2 *>(flow) + ((int64_t)m * side + row0) * side;
    for (int p = threadIdx.x; p < npix; p += kBandThreads) {
        const int o = DILATE ? (w == 4 ? window_owner4(owner, pw, p / side, p % side) : window_owner(owner, pw, p / side, p % side, w))
                             : owner[p];
        double2 val = make_double2(0.0, 0.0);
        if (o > 0) {
            const int64_t i = o - 1;
            const double x = pts[3 * i], y = pts[3 * i + 1], zz = pts[3 * i + 2];
            double u1, v1, u2, v2, z;
            project_point(P1, x, y, zz, u1, v1, z);
            crop_apply(cx, tall, 1, u1, v1);
            project_point(P2, x, y, zz, u2, v2, z);
            crop_apply(cx, tall, 1, u2, v2);
            val.x = u2 - u1;
            val.y = v2 - v1;
        }
        out[p] = val;
    }
}

// make_disparity in two launches, for callers that bring a workspace.
// make_disparity_kernel above is bound by vector instructions, and spends most of them twice: each of a map's two
// half-map bands projects every point.  Here the per-point part runs once per (map, point), in a launch with as
// many waves as the chip holds; a point that is inside the crop and visible at pose 1 becomes the record
// (remapped v << 16) | remapped u, and the band blocks only read records.
#ifndef DISP_PPT
#define DISP_PPT 4
#endif
constexpr int kPointsPerThread = DISP_PPT;    // per map: amortises the crop transform (two float64 divisions)
// The kernel leaves per (map, band) a COMPACT list of the kept points that can write into that band: (record, tag)
// pairs with tag = rank + 1 (the point's index in the reference's order; the
// position + 1 without ranks).  A workgroup reserves room for its entries with one atomic per band on the list's
// length (zeroed by the caller of the kernel), so a list is contiguous and in no particular order -- the owner of
// a pixel is a maximum, which does not depend on it.  The band blocks then read a few hundred KB of entries per
// map instead of scanning every point's record (and rank) once per band.
constexpr int kDispBandRows = kBandPixels / (int)ispa_geom::kCropSide;
constexpr int kDispBands = ((int)ispa_geom::kCropSide + kDispBandRows - 1) / kDispBandRows;
constexpr int kResolveChunk = 4;             // x 64 pixels per wave between two queue drains
constexpr int kMapRoundPoints = 4;           // points per thread and round of disparity_maps_kernel
constexpr size_t kResolveQueueBytes = (size_t)kBandThreads * kResolveChunk * sizeof(int2) + 8;
// base + the number of lanes below this one whose bit is set in `mask`
__device__ __forceinline__ int lanes_below(unsigned long long mask, int base) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, base));
}

// Per map, once: the crop transform (a float64 division that every thread of the two kernels below would
// otherwise repeat) and the empty lists.
struct DisparityMapSetup {
    ispa_geom::CropXform cx;
    int tall, unused;
};
__global__ __launch_bounds__(256) void disparity_setup_kernel(const int32_t *__restrict__ crop, int crop_per_map,
                                                              int n_map, DisparityMapSetup *__restrict__ setup,
                                                              int32_t *__restrict__ counts) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_map) return;
    const int32_t *c = crop + (crop_per_map ? 4 * (int64_t)m : 0);
    DisparityMapSetup su;
    su.cx = ispa_geom::make_crop(c);
    su.tall = (c[1] - c[0]) >= (c[3] - c[2]);
    su.unused = 0;
    setup[m] = su;
    for (int b = 0; b < kDispBands; ++b) counts[m * kDispBands + b] = 0;
}

__global__ __launch_bounds__(256) void disparity_points_kernel(
    const double *__restrict__ pts, int64_t n, const double *__restrict__ cams1, const float *__restrict__ depth,
    int depth_per_map, int H, int W, int ylim, int xlim, double relax, const DisparityMapSetup *__restrict__ setup,
    int n_map, int nbx, const int32_t *__restrict__ rank, int spread, int2 *__restrict__ lists,
    int32_t *__restrict__ counts) {
    using namespace ispa_geom;
    // 1-D grid; workgroups b and b + 8 run on the same XCD.  All point blocks of a map go to one XCD and follow
    // each other there, so that the depth texels the map's points gather (a few hundred KB around the object)
    // stay in that XCD's L2 instead of competing with the regions of eighty other maps.
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
    const int m = (wq / nbx) * 8 + xcd;
    if (m >= n_map) return;
    __shared__ int s_cnt[kDispBands][4], s_base[kDispBands];
    const double *P1 = cams1 + 12 * (int64_t)m;
    const float *d = depth + (depth_per_map ? (int64_t)m * H * W : 0);
    const CropXform cx = setup[m].cx;
    const int tall = setup[m].tall;
    const int blk = wq % nbx;
    // this workgroup's points: 32-bit offsets from a uniform base
    const int64_t first = (int64_t)blk * (256 * kPointsPerThread);
    const int here = (int)min((int64_t)(256 * kPointsPerThread), n - first);
    const double *pb = pts + 3 * first;
    const int32_t *rb = rank ? rank + first : nullptr;
    constexpr int kBatch = 4;                     // loads, then gathers, of four points in flight together
    static_assert(kPointsPerThread == kBatch, "one batch per thread: the compaction below indexes it by e");
    int word[kBatch], tag[kBatch];
    {
        double px[kBatch], py[kBatch], pz[kBatch];
        int rk[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int l = (int)threadIdx.x + e * 256;
            const int j = l < here ? l : 0;
            px[e] = pb[3 * j];
            py[e] = pb[3 * j + 1];
            pz[e] = pb[3 * j + 2];
            rk[e] = rb ? rb[j] : (int)first + j;
        }
        bool ok[kBatch], gate[kBatch];
        int at[kBatch];                               // H * W fits 31 bits (checked by the caller)
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            double u, v, z;
            project_point(P1, px[e], py[e], pz[e], u, v, z);
            pz[e] = z;
            double uc = u, vc = v;
            ok[e] = crop_apply(cx, tall, 1, uc, vc);
            // inside the crop the remapped coordinates lie in [0, kCropSide]: they fit 16 bits each
            word[e] = (cvt_i32_sat(vc) << 16) | (cvt_i32_sat(uc) & 0xffff);
            const int gx = cvt_i32_sat(u), gy = cvt_i32_sat(v);   // depth gate on the integer-valued pixel (pt2render.py:47-49)
            gate[e] = gy > 0 && gy < ylim && gx > 0 && gx < xlim;
            at[e] = gate[e] ? gy * W + gx : 0;
        }
        float texel[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) texel[e] = d[at[e]];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const bool keep = ok[e] && depth_decide_a(gate[e], pz[e], texel[e], relax);
            // a rank outside [0, n) drops its point (it can never own a pixel)
            const bool listed = keep && (int)threadIdx.x + e * 256 < here && (unsigned)rk[e] < (unsigned)n;
            tag[e] = rk[e] + 1;
            word[e] = listed ? word[e] : -1;
        }
    }
    // ---- compaction: per band, this workgroup's kept points that can reach the band's rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long mask[kDispBands][kBatch];
#pragma unroll
    for (int bnd = 0; bnd < kDispBands; ++bnd) {
        const int row0 = bnd * kDispBandRows, row1 = min(row0 + kDispBandRows, (int)kCropSide);
        int total = 0;
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int iv = word[e] >> 16;
            // the rows a point writes lie in [iv - spread, iv + spread)
            const bool in = word[e] >= 0 && iv + spread >= row0 && iv - spread < row1;
            mask[bnd][e] = __builtin_amdgcn_ballot_w64(in);
            total += __popcll(mask[bnd][e]);
        }
        if (lane == 0) s_cnt[bnd][wave] = total;
    }
    __syncthreads();
    if (threadIdx.x < kDispBands) {
        const int bnd = threadIdx.x;
        const int all = s_cnt[bnd][0] + s_cnt[bnd][1] + s_cnt[bnd][2] + s_cnt[bnd][3];
        s_base[bnd] = all ? atomicAdd(&counts[m * kDispBands + bnd], all) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int bnd = 0; bnd < kDispBands; ++bnd) {
        int at = s_base[bnd];
#pragma unroll
        for (int w2 = 0; w2 < 3; ++w2) at += w2 < wave ? s_cnt[bnd][w2] : 0;
        int2 *dst = lists + ((int64_t)m * kDispBands + bnd) * n;
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            if ((mask[bnd][e] >> lane) & 1ull) dst[lanes_below(mask[bnd][e], at)] = make_int2(word[e], tag[e]);
            at += __popcll(mask[bnd][e]);
        }
    }
}

// ---- one band of one map in LDS: shared by the kernel that reads lists and the kernel that owns a whole map
struct BandGeom {
    int row0, row1, npix;     // map rows [row0, row1) and their pixel count
    int w, pw, nimg;          // DILATE: window side, pitch and size of the anchor image
};
template <bool DILATE>
__device__ __forceinline__ BandGeom band_geom(int band, int spread) {
    constexpr int side = (int)ispa_geom::kCropSide;
    BandGeom g;
    g.row0 = band * kDispBandRows;
    g.row1 = min(g.row0 + kDispBandRows, side);
    g.npix = (g.row1 - g.row0) * side;
    // DILATE: anchor image of (band rows + w - 1) rows of side + w - 1 anchors, at a pitch of whole 16-byte groups
    g.w = 2 * spread;
    g.pw = (side + g.w - 1 + 3) & ~3;
    g.nimg = DILATE ? (g.row1 - g.row0 + g.w - 1) * g.pw : g.npix;
    return g;
}

// a kept point (its record and tag) writes what it owns of the band
template <bool DILATE>
__device__ __forceinline__ void band_splat(int *owner, const BandGeom &g, int spread, int word, int tag) {
    constexpr int side = (int)ispa_geom::kCropSide;
    if (word < 0) return;
    const int iu = word & 0xffff, iv = word >> 16;
    if (DILATE) {
        // as in make_disparity_kernel: for remapped coordinates in [0, side] the Python slice is empty
        // below the spread and anchored at c - spread otherwise
        if (iu < spread || iv < spread || iu >= side + spread || iv >= side + spread) return;
        const int lr = iv - spread - (g.row0 - (g.w - 1)), lc = iu - spread + g.w - 1;
        if (lr < 0 || iv - spread >= g.row1) return;
        // the anchor's own position (mod 8) rides in the low bits: the pixel that finds this value as
        // its window maximum knows which of its w x w anchors it came from, i.e. the owner's pixel
        atomicMax(&owner[lr * g.pw + lc], (tag << 6) | ((lr & 7) << 3) | (lc & 7));
    } else {
        int y0, y1, x0, x1;
        py_slice(iv, spread, side, y0, y1);
        y0 = max(y0, g.row0);
        y1 = min(y1, g.row1);
        if (y0 >= y1) return;
        py_slice(iu, spread, side, x0, x1);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) atomicMax(&owner[(y - g.row0) * side + x], tag);
    }
}

// After the last band_splat (the caller has synchronised): owners of the band's pixels, flow values, stores.
// Called by all threads of the workgroup; `owner` has room for the image and kResolveQueueBytes behind it.
template <bool DILATE, bool ORDERED>
__device__ __forceinline__ void band_resolve(int *owner, const BandGeom &g, int spread, const double *__restrict__ pts,
                                             int64_t n, const double *__restrict__ P1, const double *__restrict__ P2,
                                             const ispa_geom::CropXform &cx, int tall, double2 *__restrict__ out,
                                             const int32_t *__restrict__ inverse) {
    using namespace ispa_geom;
    constexpr int side = (int)kCropSide;
    const int w = g.w, pw = g.pw;
    if (DILATE && w == 4) {
        // The window maximum is separable.  In place: every anchor becomes the maximum of itself and its three
        // right-hand neighbours (16-byte reads and writes, four anchors per thread and round; all rounds are read
        // before the first is written), and a pixel then takes the maximum of four values of its column.
        constexpr int kGroups = side / 4;                          // per row; the anchors right of them are only read
        constexpr int kRounds = ((kDispBandRows + 3) * kGroups + kBandThreads - 1) / kBandThreads;
        const int ngroups = (g.row1 - g.row0 + 3) * kGroups;
        int4 h[kRounds];
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int q = k * kBandThreads + (int)threadIdx.x;
            if (q < ngroups) {
                const int *src = owner + (q / kGroups) * pw + (q % kGroups) * 4;
                const int4 a = *reinterpret_cast<const int4 *>(src), b = *reinterpret_cast<const int4 *>(src + 4);
                const int zw = max(a.z, a.w), xy = max(b.x, b.y);
                h[k] = make_int4(max(max(a.x, a.y), zw), max(max(a.y, b.x), zw), max(zw, xy), max(max(a.w, b.z), xy));
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int q = k * kBandThreads + (int)threadIdx.x;
            if (q < ngroups) *reinterpret_cast<int4 *>(owner + (q / kGroups) * pw + (q % kGroups) * 4) = h[k];
        }
        __syncthreads();
    }
    // Less than half of a map's pixels have an owner, and the float64 projection is what the resolve costs: each
    // wave first finds the owners of kResolveChunk x 64 pixels (empty pixels get their zeros at once), queues the
    // owned ones in its own stretch of LDS, and then projects with all lanes busy.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int2 *queue = reinterpret_cast<int2 *>(owner + ((g.nimg + 1) & ~1)) + wave * (64 * kResolveChunk);
    for (int p0 = 0; p0 < g.npix; p0 += kBandThreads * kResolveChunk) {
        int queued = 0;
#pragma unroll
        for (int c = 0; c < kResolveChunk; ++c) {
            const int p = p0 + c * kBandThreads + (int)threadIdx.x;
            int o = 0;
            if (p < g.npix) {
                const int ly = p / side, x = p % side;
                if (DILATE && w == 4) {
                    const int *col = owner + ly * pw + x;
                    o = max(max(col[0], col[pw]), max(col[2 * pw], col[3 * pw]));
                } else {
                    o = DILATE ? window_owner(owner, pw, ly, x, w) : owner[p];
                }
                if (o <= 0) out[p] = make_double2(0.0, 0.0);
            }
            const unsigned long long owned = __builtin_amdgcn_ballot_w64(o > 0);
            if (o > 0) queue[lanes_below(owned, queued)] = make_int2(p, o);
            queued += __popcll(owned);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < queued; k += 64) {
            const int2 item = queue[k];
            const int p = item.x, o = item.y;
            const int ly = p / side, x = p % side;
            double2 val = make_double2(0.0, 0.0);
            int64_t i = (DILATE ? (o >> 6) : o) - 1;
            if (ORDERED) {
                i = inverse[i];
                if (i >= n) i = -1;
            }
            if (i >= 0) {
                const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
                double u1, v1, u2, v2, z;
                if (DILATE) {
                    // the winning anchor sits dr rows and dc columns up-left of this pixel's last anchor, and an
                    // anchor is its point's remapped pose-1 pixel less the spread: no second projection
                    const int dr = (ly + w - 1 - (o >> 3)) & 7, dc = (x + w - 1 - o) & 7;
                    u1 = (double)(x - dc + spread);
                    v1 = (double)(g.row0 + ly - dr + spread);
                } else {
                    project_point(P1, px, py, pz, u1, v1, z);
                    crop_apply(cx, tall, 1, u1, v1);
                }
                project_point(P2, px, py, pz, u2, v2, z);
                crop_apply(cx, tall, 1, u2, v2);
                val.x = u2 - u1;
                val.y = v2 - v1;
            }
            out[p] = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ORDERED: the points arrive in a processing order of the caller's choice (e.g. along a space-filling curve, so
// that the points kernel's depth gathers coalesce) and rank[i] is point i's index in the reference's order: the
// kept point with the highest RANK wins a pixel, and inverse[rank] finds it again for the flow value.  Ranks
// outside [0, n) drop their point and a bad inverse entry leaves the pixel empty: no read outside the arrays.
template <bool DILATE, bool ORDERED>
__global__ __launch_bounds__(kBandThreads) void disparity_bands_kernel(
    const double *__restrict__ pts, int64_t n, const double *__restrict__ cams1, const double *__restrict__ cams2,
    const DisparityMapSetup *__restrict__ setup, int spread, double *__restrict__ flow,
    const int2 *__restrict__ lists, const int32_t *__restrict__ counts, const int32_t *__restrict__ inverse) {
    extern __shared__ int owner[];
    constexpr int side = (int)ispa_geom::kCropSide;
    const int m = blockIdx.y;
    const BandGeom g = band_geom<DILATE>(blockIdx.x, spread);
    const int2 *lst = lists + ((int64_t)m * kDispBands + blockIdx.x) * n;
    // Two batches of entries per thread travel while the image is cleared, and they are asked for before the
    // list's length is known (a list has room for n entries, so the read is in bounds; what lies beyond the
    // length is dropped below): the length and the entries arrive together instead of one after the other.
    constexpr int kBatch = 8;
    int2 ent[2][kBatch];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int f = (h * kBatch + e) * kBandThreads + threadIdx.x;
            ent[h][e] = lst[f < n ? f : 0];
        }
    const int total = min(counts[m * kDispBands + blockIdx.x], (int)n);
    for (int p = threadIdx.x; p < g.nimg; p += kBandThreads) owner[p] = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 2 * kBandThreads * kBatch) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int2 cur[kBatch];
#pragma unroll
            for (int e = 0; e < kBatch; ++e) cur[e] = ent[h][e];
            // the same half of the next round leaves now
            const int next = base + (2 + h) * kBatch * kBandThreads;
            if (next < total) {
#pragma unroll
                for (int e = 0; e < kBatch; ++e) {
                    const int f = next + e * kBandThreads + threadIdx.x;
                    ent[h][e] = lst[f < total ? f : 0];
                }
            }
#pragma unroll
            for (int e = 0; e < kBatch; ++e) {
                const int f = base + (h * kBatch + e) * kBandThreads + threadIdx.x;
                band_splat<DILATE>(owner, g, spread, f < total ? cur[e].x : -1, cur[e].y);
            }
        }
    }
    __syncthreads();
    band_resolve<DILATE, ORDERED>(owner, g, spread, pts, n, cams1 + 12 * (int64_t)m, cams2 + 12 * (int64_t)m,
                                  setup[m].cx, setup[m].tall,
                                  reinterpret_cast<double2 *>(flow) + ((int64_t)m * side + g.row0) * side, inverse);
}

// One workgroup per MAP, for calls with about as many maps as the chip has compute units (or whole multiples):
// the workgroup projects the map's points once, splats the first band's points straight into LDS while each wave
// keeps the second band's in a list of its own (global memory that only this wave touches: no counters, no
// atomics, read back from the L2 it was written to), resolves the first band, then splats and resolves the second.
// One launch, and no list ever waits in HBM for another kernel.
template <bool DILATE, bool ORDERED>
__global__ __launch_bounds__(kBandThreads) void disparity_maps_kernel(
    const double *__restrict__ pts, int64_t n, const double *__restrict__ cams1, const double *__restrict__ cams2,
    const float *__restrict__ depth, int depth_per_map, int H, int W, int ylim, int xlim, double relax,
    const int32_t *__restrict__ crop, int crop_per_map, int spread, double *__restrict__ flow,
    int2 *__restrict__ lists, int64_t wave_pitch, const int32_t *__restrict__ rank, const int32_t *__restrict__ inverse) {
    using namespace ispa_geom;
    static_assert(kDispBands == 2, "the first band goes to LDS, the second to the lists");
    extern __shared__ int owner[];
    constexpr int side = (int)kCropSide;
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *P1 = cams1 + 12 * (int64_t)m;
    const double *P2 = cams2 + 12 * (int64_t)m;
    const float *d = depth + (depth_per_map ? (int64_t)m * H * W : 0);
    const int32_t *c = crop + (crop_per_map ? 4 * (int64_t)m : 0);
    const CropXform cx = make_crop(c);
    const int tall = (c[1] - c[0]) >= (c[3] - c[2]);
    double2 *out = reinterpret_cast<double2 *>(flow) + (int64_t)m * side * side;
    BandGeom g = band_geom<DILATE>(0, spread);
    for (int p = threadIdx.x; p < g.nimg; p += kBandThreads) owner[p] = 0;
    __syncthreads();
    int2 *mine = lists + ((int64_t)m * (kBandThreads / 64) + wave) * wave_pitch;
    int listed1 = 0;                                  // entries in this wave's list (uniform)
    constexpr int kBatch = kMapRoundPoints;           // loads, then gathers, of four points in flight together
    // A round's depth texels are gathered one round ahead of their use: the visibility decision, the splat and
    // the list entry of round r run after round r + 1's projections, so that the gather's latency is covered by
    // this wave's own arithmetic (four waves per SIMD do not cover it for each other).
    int c_word[kBatch], c_tag[kBatch];                // round r, waiting for its texels:
    double c_z[kBatch];                               //   record if the point is kept, tag, depth at pose 1
    float c_texel[kBatch];
    bool c_pass[kBatch];                              //   inside the crop, inside the gate, a valid rank
    auto finish = [&]() {
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const bool keep = c_pass[e] && depth_decide_a(true, c_z[e], c_texel[e], relax);
            const int rec = keep ? c_word[e] : -1;
            band_splat<DILATE>(owner, g, spread, rec, c_tag[e]);
            // the rows a point writes lie in [iv - spread, iv + spread)
            const bool second = rec >= 0 && (rec >> 16) + spread >= kDispBandRows;
            const unsigned long long take = __builtin_amdgcn_ballot_w64(second);
            if (second) mine[lanes_below(take, listed1)] = make_int2(rec, c_tag[e]);
            listed1 += __popcll(take);
        }
    };
    for (int64_t first = 0; first < n; first += kBandThreads * kBatch) {
        const int here = (int)min((int64_t)(kBandThreads * kBatch), n - first);
        const double *pb = pts + 3 * first;           // this round's points: 32-bit offsets from a uniform base
        const int32_t *rb = rank ? rank + first : nullptr;
        double px[kBatch], py[kBatch], pz[kBatch];
        int rk[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const int l = (int)threadIdx.x + e * kBandThreads;
            const int j = l < here ? l : 0;
            px[e] = pb[3 * j];
            py[e] = pb[3 * j + 1];
            pz[e] = pb[3 * j + 2];
            rk[e] = rb ? rb[j] : (int)first + j;
        }
        int word[kBatch], at[kBatch];                 // H * W fits 31 bits (checked by the caller)
        bool pass[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            double u, v, z;
            project_point(P1, px[e], py[e], pz[e], u, v, z);
            pz[e] = z;
            double uc = u, vc = v;
            const bool ok = crop_apply(cx, tall, 1, uc, vc);
            word[e] = (cvt_i32_sat(vc) << 16) | (cvt_i32_sat(uc) & 0xffff);
            const int gx = cvt_i32_sat(u), gy = cvt_i32_sat(v);   // pt2render.py:47-49
            const bool gate = gy > 0 && gy < ylim && gx > 0 && gx < xlim;
            at[e] = gate ? gy * W + gx : 0;
            // a rank outside [0, n) drops its point (it can never own a pixel)
            pass[e] = ok && gate && (int)threadIdx.x + e * kBandThreads < here && (unsigned)rk[e] < (unsigned)n;
        }
        float texel[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) texel[e] = d[at[e]];
        if (first > 0) finish();
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            c_word[e] = word[e];
            c_tag[e] = rk[e] + 1;
            c_z[e] = pz[e];
            c_texel[e] = texel[e];
            c_pass[e] = pass[e];
        }
    }
    if (n > 0) finish();
    __syncthreads();
    band_resolve<DILATE, ORDERED>(owner, g, spread, pts, n, P1, P2, cx, tall, out, inverse);
    // this wave's list, written above, is read back by the same wave: no wider scope is needed (a fence at
    // device scope would write back and invalidate the L2 that 31 other workgroups are working from)
    __threadfence_block();
    // sixteen entries per lane in flight, the first sixteen while the image is cleared, and the next sixteen leave
    // before these are used
    constexpr int kDepth = 16;
    int2 en[kDepth];
#pragma unroll
    for (int e = 0; e < kDepth; ++e) {
        const int k = e * 64 + lane;
        en[e] = k < listed1 ? mine[k] : make_int2(-1, 0);
    }
    __syncthreads();
    g = band_geom<DILATE>(1, spread);
    for (int p = threadIdx.x; p < g.nimg; p += kBandThreads) owner[p] = 0;
    __syncthreads();
    {
        for (int k0 = 0; k0 < listed1; k0 += 64 * kDepth) {
            int2 cur[kDepth];
#pragma unroll
            for (int e = 0; e < kDepth; ++e) cur[e] = en[e];
            if (k0 + 64 * kDepth < listed1) {
#pragma unroll
                for (int e = 0; e < kDepth; ++e) {
                    const int k = k0 + (kDepth + e) * 64 + lane;
                    en[e] = k < listed1 ? mine[k] : make_int2(-1, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < kDepth; ++e) band_splat<DILATE>(owner, g, spread, cur[e].x, cur[e].y);
        }
    }
    __syncthreads();
    band_resolve<DILATE, ORDERED>(owner, g, spread, pts, n, P1, P2, cx, tall, out + (int64_t)g.row0 * side, inverse);
}

__global__ __launch_bounds__(kThreads) void depth_owner_kernel(const double *__restrict__ pt_render,
                                                               int64_t n, int H, int W, int radius,
                                                               int32_t *__restrict__ owner) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    int u, v;
    if (!to_int(pt_render[i], u) || !to_int(pt_render[n + i], v)) return;
    int y0, y1, x0, x1;
    py_slice(v, radius, H, y0, y1);
    py_slice(u, radius, W, x0, x1);
    const int32_t tag = (int32_t)i + 1;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) atomicMax(&owner[(int64_t)y * W + x], tag);
}

__global__ __launch_bounds__(kThreads) void depth_resolve_kernel(const double *__restrict__ pt_render,
                                                                 int64_t n, int64_t npix,
                                                                 const int32_t *__restrict__ owner,
                                                                 double *__restrict__ out) {
    const int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (pix >= npix) return;
    const int32_t o = owner[pix];
    out[pix] = o > 0 ? pt_render[2 * n + (o - 1)] : 0.0;
}

__device__ __forceinline__ bool zbuf_pixel(const double *__restrict__ r, int64_t n, int64_t i, int H, int W,
                                           int &x, int &y, double &z) {
    z = r[2 * n + i];
    if (!to_int(r[i], x) || !to_int(r[n + i], y)) return false;
    // z must be finite and positive; NaN fails the first comparison
    return (z > 0.0) && (z < __builtin_inf()) && x >= 0 && x < W && y >= 0 && y < H;
}

__global__ __launch_bounds__(kThreads) void zbuffer_kernel(const double *__restrict__ pt_render, int64_t n,
                                                           int H, int W, int radius,
                                                           unsigned long long *__restrict__ packed) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int p = blockIdx.y;
    if (i >= n) return;
    int x, y;
    double z;
    if (!zbuf_pixel(pt_render + (int64_t)p * 3 * n, n, i, H, W, x, y, z)) return;
    const unsigned long long key =
        ((unsigned long long)__float_as_uint((float)z) << 32) | (unsigned long long)(uint32_t)i;
    unsigned long long *img = packed + (int64_t)p * H * W;
    const int y0 = max(y - radius, 0), y1 = min(y + radius, H - 1);
    const int x0 = max(x - radius, 0), x1 = min(x + radius, W - 1);
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) atomicMin(&img[(int64_t)yy * W + xx], key);
}

__global__ __launch_bounds__(kThreads) void zbuffer_resolve_kernel(const unsigned long long *__restrict__ packed,
                                                                   int64_t total, int32_t *__restrict__ winner,
                                                                   float *__restrict__ depth) {
    const int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (pix >= total) return;
    const unsigned long long k = packed[pix];
    const bool empty = k == ~0ull;
    if (winner) winner[pix] = empty ? -1 : (int32_t)(uint32_t)(k & 0xffffffffull);
    if (depth) depth[pix] = empty ? __builtin_inff() : __uint_as_float((uint32_t)(k >> 32));
}

// PACKED: the depth comes straight from the high half of the packed z-buffer keys (no resolve pass)
template <bool PACKED>
__global__ __launch_bounds__(kThreads) void zbuffer_visibility_kernel(const double *__restrict__ pt_render,
                                                                      int64_t n, const void *__restrict__ image,
                                                                      int H, int W, double relax,
                                                                      uint8_t *__restrict__ truth) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int p = blockIdx.y;
    if (i >= n) return;
    int x, y;
    double z;
    bool vis = zbuf_pixel(pt_render + (int64_t)p * 3 * n, n, i, H, W, x, y, z);
    if (vis) {
        const int64_t pix = ((int64_t)p * H + y) * W + x;
        float d;
        if (PACKED) {
            // the 4 high bytes of the little-endian key; an empty pixel (all ones) counts as +inf, as
            // in the resolved depth image
            const uint32_t hi = reinterpret_cast<const uint32_t *>(image)[2 * pix + 1];
            d = hi == 0xffffffffu ? __builtin_inff() : __uint_as_float(hi);
        } else {
            d = reinterpret_cast<const float *>(image)[pix];
        }
        vis = (z - (double)d) <= relax;
    }
    truth[(int64_t)p * n + i] = vis;
}

}  // namespace

namespace {

// rows x cols flow maps; square maps of up to 8 LDS bands take the banded kernel
int flow_splat_impl(const double *uv1, const double *uv2, const uint8_t *truth, int64_t n, int n_map, int spread,
                    int rows, int cols, int32_t *owner_ws, double *flow, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_map >= 0 && n < (1ll << 31) - 1, "bad point count");
    ISPA_REQUIRE(rows > 0 && rows <= 16384 && cols > 0 && cols <= 16384 && spread >= 0 && spread <= 64,
                 "bad map size / spread");
    if (n_map == 0) return ISPA_OK;
    // up to 8 LDS bands per map (side <= 448): each band block scans all points once, no scratch image;
    // larger (and rectangular) maps take one global atomic per covered pixel on the owner image instead of
    // hundreds of scans
    const int side = rows;
    const bool banded = rows == cols && (int64_t)side * side <= 8ll * kBandPixels;
    ISPA_REQUIRE(flow && (owner_ws || banded),
                 "null pointer (rectangular maps and maps of more than %d pixels need owner_ws)", 8 * kBandPixels);
    ISPA_REQUIRE(n_map <= 65535, "too many maps per call (%d)", n_map);
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)rows * cols;
    if (banded) {
        // banded LDS path (owner_ws is not touched)
        ISPA_REQUIRE(n == 0 || (uv1 && uv2 && truth), "null pointer");
        const int band_rows = kBandPixels / side;
        const int nbands = (side + band_rows - 1) / band_rows;
        // small windows: one atomic per point on an anchor image + a window maximum per pixel
        const size_t dilate_bytes = (size_t)(band_rows + 2 * spread - 1) * (side + 2 * spread - 1) * sizeof(int);
        const bool dilate = spread >= 1 && spread <= kMaxDilateSpread && side >= 2 * spread && dilate_bytes <= kMaxBandLds;
        const size_t lds_bytes = dilate ? dilate_bytes : (size_t)band_rows * side * sizeof(int);
        static unsigned long long configured[2] = {0, 0};
        if (int rc = ispa::allow_dynamic_lds(dilate ? reinterpret_cast<const void *>(&flow_band_kernel<true>)
                                                    : reinterpret_cast<const void *>(&flow_band_kernel<false>),
                                             kMaxBandLds, &configured[dilate]))
            return rc;
        const ispa::TimingSlot timed = ispa::timing_begin(s);
        hipLaunchKernelGGL(dilate ? flow_band_kernel<true> : flow_band_kernel<false>, dim3(nbands, n_map),
                           dim3(kBandThreads), lds_bytes, s, uv1, uv2, truth, n, spread, side, band_rows, flow);
        ispa::timing_end(timed, s);
        return ispa::check_launch("flow_band_kernel");
    }
    ISPA_HIP(hipMemsetAsync(owner_ws, 0, npix * n_map * sizeof(int32_t), s));
    if (n > 0) {
        ISPA_REQUIRE(uv1 && uv2 && truth, "null pointer");
        hipLaunchKernelGGL(flow_owner_kernel, dim3(ispa::blocks_for(n, kThreads), n_map), dim3(kThreads), 0, s,
                           uv1, truth, n, spread, rows, cols, owner_ws);
        if (int rc = ispa::check_launch("flow_owner_kernel")) return rc;
    }
    hipLaunchKernelGGL(flow_resolve_kernel, dim3(ispa::blocks_for((int64_t)npix, kThreads), n_map),
                       dim3(kThreads), 0, s, uv1, uv2, n, (int64_t)npix, owner_ws, flow);
    return ispa::check_launch("flow_resolve_kernel");
}

}  // namespace

extern "C" int ispa_flow_splat(const double *uv1, const double *uv2, const uint8_t *truth, int64_t n,
                               int n_map, int spread, int side, int32_t *owner_ws, double *flow,
                               void *stream) {
    return flow_splat_impl(uv1, uv2, truth, n, n_map, spread, side, side, owner_ws, flow, stream);
}

extern "C" int ispa_flow_splat_hw(const double *uv1, const double *uv2, const uint8_t *truth, int64_t n,
                                  int n_map, int spread, int H, int W, int32_t *owner_ws, double *flow,
                                  void *stream) {
    return flow_splat_impl(uv1, uv2, truth, n, n_map, spread, H, W, owner_ws, flow, stream);
}

extern "C" int ispa_point_depth_splat(const double *pt_render, int64_t n, int H, int W, int radius,
                                      int32_t *owner_ws, double *depth_out, void *stream) {
    ISPA_REQUIRE(n >= 0 && n < (1ll << 31) - 1, "bad point count");
    ISPA_REQUIRE(H > 0 && W > 0 && radius >= 0 && radius <= 64, "bad shape/radius");
    ISPA_REQUIRE(owner_ws && depth_out, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)H * W;
    ISPA_HIP(hipMemsetAsync(owner_ws, 0, (size_t)npix * sizeof(int32_t), s));
    if (n > 0) {
        ISPA_REQUIRE(pt_render, "null pointer");
        hipLaunchKernelGGL(depth_owner_kernel, dim3(ispa::blocks_for(n, kThreads)), dim3(kThreads), 0, s,
                           pt_render, n, H, W, radius, owner_ws);
        if (int rc = ispa::check_launch("depth_owner_kernel")) return rc;
    }
    hipLaunchKernelGGL(depth_resolve_kernel, dim3(ispa::blocks_for(npix, kThreads)), dim3(kThreads), 0, s,
                       pt_render, n, npix, owner_ws, depth_out);
    return ispa::check_launch("depth_resolve_kernel");
}

extern "C" int ispa_zbuffer(const double *pt_render, int64_t n, int n_pose, int H, int W, int radius,
                            uint64_t *packed, int32_t *winner, float *depth, void *stream) {
    ISPA_REQUIRE(n >= 0 && n < (1ll << 32) && n_pose >= 0, "bad point count");
    ISPA_REQUIRE(H > 0 && W > 0 && radius >= 0 && radius <= 64, "bad shape/radius");
    if (n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(packed, "null pointer");
    ISPA_REQUIRE(n_pose <= 65535, "too many poses per call (%d)", n_pose);
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n_pose * H * W;
    ISPA_HIP(hipMemsetAsync(packed, 0xff, (size_t)total * sizeof(uint64_t), s));
    if (n > 0) {
        ISPA_REQUIRE(pt_render, "null pointer");
        const ispa::TimingSlot timed = ispa::timing_begin(s);
        hipLaunchKernelGGL(zbuffer_kernel, dim3(ispa::blocks_for(n, kThreads), n_pose), dim3(kThreads), 0, s,
                           pt_render, n, H, W, radius, (unsigned long long *)packed);
        ispa::timing_end(timed, s);
        if (int rc = ispa::check_launch("zbuffer_kernel")) return rc;
    }
    if (winner || depth) {
        hipLaunchKernelGGL(zbuffer_resolve_kernel, dim3(ispa::blocks_for(total, kThreads)), dim3(kThreads), 0,
                           s, (const unsigned long long *)packed, total, winner, depth);
        return ispa::check_launch("zbuffer_resolve_kernel");
    }
    return ISPA_OK;
}

extern "C" int ispa_zbuffer_visibility(const double *pt_render, int64_t n, int n_pose, const float *depth,
                                       int H, int W, double relax, uint8_t *truth, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_pose >= 0, "negative size");
    if (n == 0 || n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(pt_render && depth && truth, "null pointer");
    ISPA_REQUIRE(H > 0 && W > 0, "bad shape");
    ISPA_REQUIRE(n_pose <= 65535, "too many poses per call (%d)", n_pose);
    hipLaunchKernelGGL(zbuffer_visibility_kernel<false>, dim3(ispa::blocks_for(n, kThreads), n_pose), dim3(kThreads),
                       0, (hipStream_t)stream, pt_render, n, depth, H, W, relax, truth);
    return ispa::check_launch("zbuffer_visibility_kernel");
}

extern "C" int ispa_zbuffer_visibility_packed(const double *pt_render, int64_t n, int n_pose, const uint64_t *packed,
                                              int H, int W, double relax, uint8_t *truth, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_pose >= 0, "negative size");
    if (n == 0 || n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(pt_render && packed && truth, "null pointer");
    ISPA_REQUIRE(H > 0 && W > 0, "bad shape");
    ISPA_REQUIRE(n_pose <= 65535, "too many poses per call (%d)", n_pose);
    hipLaunchKernelGGL(zbuffer_visibility_kernel<true>, dim3(ispa::blocks_for(n, kThreads), n_pose), dim3(kThreads),
                       0, (hipStream_t)stream, pt_render, n, packed, H, W, relax, truth);
    return ispa::check_launch("zbuffer_visibility_kernel");
}

namespace {
int make_disparity_impl(const double *pts, int64_t n, const double *cams1, const double *cams2, int n_map,
                        const float *depth, int depth_per_map, int H, int W, int ylim, int xlim, double relax,
                        const int32_t *crop, int crop_per_map, int spread, double *flow, void *workspace, void *stream,
                        const int32_t *rank = nullptr, const int32_t *inverse = nullptr);
}

// entries a wave of disparity_maps_kernel can list: one per lane and point of a round
static size_t disparity_wave_pitch(int64_t n) {
    constexpr int64_t round = (int64_t)kMapRoundPoints * kBandThreads;
    return (size_t)((n + round - 1) / round) * (64 * kMapRoundPoints);
}
// workspace of the paths with lists: their lengths per (map, band), the per-map set-up, then the lists themselves
// at their worst-case size
struct DisparityPlan {
    size_t off_setup, off_lists, total;
};
static DisparityPlan disparity_plan(int64_t n, int n_map) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    DisparityPlan p;
    p.off_setup = up((size_t)n_map * kDispBands * sizeof(int32_t));
    p.off_lists = p.off_setup + up((size_t)n_map * sizeof(DisparityMapSetup));
    // a list per (map, band) with room for every point, or a list per (map, wave) of disparity_maps_kernel
    const size_t per_map = std::max((size_t)kDispBands * (size_t)n, (size_t)(kBandThreads / 64) * disparity_wave_pitch(n));
    p.total = p.off_lists + (size_t)n_map * per_map * sizeof(int2);
    return p;
}

extern "C" size_t ispa_make_disparity_workspace_bytes(int64_t n, int n_map) {
    if (n <= 0 || n_map <= 0) return 0;
    return disparity_plan(n, n_map).total;
}

extern "C" int ispa_make_disparity_ws(const double *pts, int64_t n, const double *cams1, const double *cams2,
                                      int n_map, const float *depth, int depth_per_map, int H, int W, int ylim,
                                      int xlim, double relax, const int32_t *crop, int crop_per_map, int spread,
                                      double *flow, void *workspace, size_t workspace_bytes, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_map >= 0, "negative size");
    if (workspace_bytes < ispa_make_disparity_workspace_bytes(n, n_map))
        return ispa::fail(ISPA_EWORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes,
                          ispa_make_disparity_workspace_bytes(n, n_map));
    ISPA_REQUIRE(workspace != nullptr || n == 0 || n_map == 0, "null workspace");
    return make_disparity_impl(pts, n, cams1, cams2, n_map, depth, depth_per_map, H, W, ylim, xlim, relax, crop,
                               crop_per_map, spread, flow, workspace, stream);
}

extern "C" int ispa_make_disparity_ordered(const double *pts, const int32_t *rank, const int32_t *inverse, int64_t n,
                                           const double *cams1, const double *cams2, int n_map, const float *depth,
                                           int depth_per_map, int H, int W, int ylim, int xlim, double relax,
                                           const int32_t *crop, int crop_per_map, int spread, double *flow,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_map >= 0, "negative size");
    ISPA_REQUIRE((rank != nullptr) == (inverse != nullptr), "rank and inverse come together (both NULL = array order)");
    if (workspace_bytes < ispa_make_disparity_workspace_bytes(n, n_map))
        return ispa::fail(ISPA_EWORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes,
                          ispa_make_disparity_workspace_bytes(n, n_map));
    ISPA_REQUIRE(workspace != nullptr || n == 0 || n_map == 0, "null workspace");
    return make_disparity_impl(pts, n, cams1, cams2, n_map, depth, depth_per_map, H, W, ylim, xlim, relax, crop,
                               crop_per_map, spread, flow, workspace, stream, rank, inverse);
}

extern "C" int ispa_make_disparity(const double *pts, int64_t n, const double *cams1, const double *cams2,
                                   int n_map, const float *depth, int depth_per_map, int H, int W, int ylim,
                                   int xlim, double relax, const int32_t *crop, int crop_per_map, int spread,
                                   double *flow, void *stream) {
    return make_disparity_impl(pts, n, cams1, cams2, n_map, depth, depth_per_map, H, W, ylim, xlim, relax, crop,
                               crop_per_map, spread, flow, nullptr, stream);
}

namespace {
int make_disparity_impl(const double *pts, int64_t n, const double *cams1, const double *cams2, int n_map,
                        const float *depth, int depth_per_map, int H, int W, int ylim, int xlim, double relax,
                        const int32_t *crop, int crop_per_map, int spread, double *flow, void *workspace, void *stream,
                        const int32_t *rank, const int32_t *inverse) {
    ISPA_REQUIRE(n >= 0 && n < (1ll << 31) - 1 && n_map >= 0, "bad point count");
    ISPA_REQUIRE(spread >= 0 && spread <= 64, "bad spread");
    if (n_map == 0) return ISPA_OK;
    ISPA_REQUIRE(cams1 && cams2 && depth && crop && flow && (n == 0 || pts), "null pointer");
    ISPA_REQUIRE((int64_t)H * W < (1ll << 31), "depth map of %d x %d texels is too large", H, W);
    ISPA_REQUIRE(H > 0 && W > 0 && ylim >= 1 && ylim <= H && xlim >= 1 && xlim <= W,
                 "visibility gate (%d,%d) exceeds the depth map (%d,%d)", ylim, xlim, H, W);
    ISPA_REQUIRE(n_map <= 65535, "too many maps per call (%d)", n_map);
    hipStream_t s = (hipStream_t)stream;
    const int side = (int)ispa_geom::kCropSide;
    const int band_rows = kBandPixels / side;
    const int nbands = (side + band_rows - 1) / band_rows;
    const size_t dilate_bytes = (size_t)(band_rows + 2 * spread - 1) * (side + 2 * spread - 1) * sizeof(int);
    const bool dilate = spread >= 1 && spread <= kMaxDilateSpread && dilate_bytes <= kMaxBandLds;
    const size_t lds_bytes = dilate ? dilate_bytes : (size_t)band_rows * side * sizeof(int);
    static unsigned long long configured[2] = {0, 0};
    if (int rc = ispa::allow_dynamic_lds(dilate ? reinterpret_cast<const void *>(&make_disparity_kernel<true>)
                                                : reinterpret_cast<const void *>(&make_disparity_kernel<false>),
                                         kMaxBandLds, &configured[dilate]))
        return rc;
    if (workspace && n > 0) {
        // with a workspace: the per-point part once per (map, point), then the band blocks on its lists
        ISPA_REQUIRE(((uintptr_t)workspace & 7) == 0, "the workspace must be 8-byte aligned");
        // the anchor values carry 6 position bits beside the point's rank
        const bool dilate_ws = dilate && n < (1 << 25) - 1;
        const size_t pitch = (size_t)((side + 2 * spread - 1 + 3) & ~3);
        const size_t lds_ws = (dilate_ws ? (size_t)(band_rows + 2 * spread - 1) * pitch * sizeof(int)
                                         : (size_t)band_rows * side * sizeof(int)) + kResolveQueueBytes;
        static unsigned long long configured_b[4] = {0, 0, 0, 0};
        const bool ordered = rank != nullptr;
        void (*bands)(const double *, int64_t, const double *, const double *, const DisparityMapSetup *, int, double *,
                      const int2 *, const int32_t *, const int32_t *) =
            ordered ? (dilate_ws ? disparity_bands_kernel<true, true> : disparity_bands_kernel<false, true>)
                    : (dilate_ws ? disparity_bands_kernel<true, false> : disparity_bands_kernel<false, false>);
        if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(bands), kMaxBandLds,
                                             &configured_b[2 * ordered + dilate_ws]))
            return rc;
        const DisparityPlan plan = disparity_plan(n, n_map);
        // One workgroup per map when that keeps the chip busy: at least 85 % of the compute units in every round.
        int dev = 0, cus = 0;
        ISPA_HIP(hipGetDevice(&dev));
        ISPA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int rounds = (n_map + cus - 1) / std::max(cus, 1);
        // (spreads of 1 to 4 only: the other spreads' pixel-by-pixel splat is not worth a second kernel)
        if (dilate_ws && cus > 0 && (int64_t)n_map * 100 >= (int64_t)rounds * cus * 85) {
            static unsigned long long configured_m[2] = {0, 0};
            void (*maps)(const double *, int64_t, const double *, const double *, const float *, int, int, int, int, int,
                         double, const int32_t *, int, int, double *, int2 *, int64_t, const int32_t *, const int32_t *) =
                ordered ? disparity_maps_kernel<true, true> : disparity_maps_kernel<true, false>;
            if (int rc = ispa::allow_dynamic_lds(reinterpret_cast<const void *>(maps), kMaxBandLds,
                                                 &configured_m[ordered]))
                return rc;
            int2 *lists = reinterpret_cast<int2 *>(reinterpret_cast<char *>(workspace) + plan.off_lists);
            const ispa::TimingSlot timed = ispa::timing_begin(s);
            hipLaunchKernelGGL(maps, dim3(n_map), dim3(kBandThreads), lds_ws, s, pts, n, cams1, cams2, depth,
                               depth_per_map, H, W, ylim, xlim, relax, crop, crop_per_map, spread, flow, lists,
                               (int64_t)disparity_wave_pitch(n), rank, inverse);
            ispa::timing_end(timed, s);
            return ispa::check_launch("disparity_maps_kernel");
        }
        int32_t *counts = reinterpret_cast<int32_t *>(workspace);
        DisparityMapSetup *setup = reinterpret_cast<DisparityMapSetup *>(reinterpret_cast<char *>(workspace) + plan.off_setup);
        int2 *lists = reinterpret_cast<int2 *>(reinterpret_cast<char *>(workspace) + plan.off_lists);
        const ispa::TimingSlot timed = ispa::timing_begin(s);
        hipLaunchKernelGGL(disparity_setup_kernel, dim3((unsigned)ispa::blocks_for(n_map, 256)), dim3(256), 0, s, crop,
                           crop_per_map, n_map, setup, counts);
        const int nbx = (int)ispa::blocks_for(n, 256 * kPointsPerThread);
        const int64_t nblocks = (int64_t)((n_map + 7) / 8) * 8 * nbx;
        ISPA_REQUIRE(nblocks < (1ll << 31), "too many points x maps for one launch");
        hipLaunchKernelGGL(disparity_points_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, pts, n, cams1, depth,
                           depth_per_map, H, W, ylim, xlim, relax, setup, n_map, nbx, rank, spread, lists, counts);
        hipLaunchKernelGGL(bands, dim3(kDispBands, n_map), dim3(kBandThreads), lds_ws, s, pts, n, cams1, cams2, setup,
                           spread, flow, lists, counts, inverse);
        ispa::timing_end(timed, s);
        return ispa::check_launch("disparity_bands_kernel");
    }
    const ispa::TimingSlot timed = ispa::timing_begin(s);
    hipLaunchKernelGGL(dilate ? make_disparity_kernel<true> : make_disparity_kernel<false>, dim3(nbands, n_map),
                       dim3(kBandThreads), lds_bytes, s, pts, n, cams1, cams2, depth, depth_per_map, H, W, ylim, xlim,
                       relax, crop, crop_per_map, spread, band_rows, flow);
    ispa::timing_end(timed, s);
    return ispa::check_launch("make_disparity_kernel");
}
}  // namespace
