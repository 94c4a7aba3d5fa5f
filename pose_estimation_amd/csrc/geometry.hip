// Projection, depth visibility and crop remap kernels (float64, bandwidth-bound).
//
// Arithmetic follows the reference's numpy code operation by operation so that rounded pixel
// coordinates and visibility flags are bit-exact:
//   pt2render.py:6-13   P @ [p;1] as a k-sequential fma chain (what dgemm does), divide, rint
//   pt2render.py:18-38  crop: subtract, multiply, add pad, rint -- separate roundings (this
//                       file is compiled with -ffp-contract=off; every fused op is explicit)
//   pt2render.py:43-55  (A) and classifier_code/.../utils/pt2render.py:43-53 (B) visibility
#include "common.h"
#include "geom_device.h"

namespace {

using namespace ispa_geom;

constexpr int kThreads = 256;
// 16-byte streaming store: the projected coordinates are written once and read by later kernels
// only, so they need not displace anything in the caches.
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream(double *dst, const double2 &v) {
    f64x2 t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<f64x2 *>(dst));
}

__global__ __launch_bounds__(kThreads) void project_kernel(const double *__restrict__ pts, int64_t n,
                                                           const double *__restrict__ cams,
                                                           int n_pose, int poses_per_block,
                                                           double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int p0 = blockIdx.y * poses_per_block;
    const int p1 = min(p0 + poses_per_block, n_pose);
    for (int p = p0; p < p1; ++p) {
        double u, v, z;
        project_point(cams + 12 * (int64_t)p, px, py, pz, u, v, z);
        double *o = out + (int64_t)p * 3 * n;
        o[i] = u;
        o[n + i] = v;
        o[2 * n + i] = z;
    }
}

// Two consecutive points per thread: 16-byte loads and stores (the widest coalesced access),
// used when n is even so that every (pose, row) starts 16-byte aligned.
__global__ __launch_bounds__(kThreads) void project2_kernel(const double *__restrict__ pts, int64_t n,
                                                            const double *__restrict__ cams, int n_pose,
                                                            int poses_per_block, double *__restrict__ out) {
    const int64_t i2 = (int64_t)blockIdx.x * kThreads + threadIdx.x;       // pair index
    if (2 * i2 >= n) return;
    const double2 *src = reinterpret_cast<const double2 *>(pts + 6 * i2);
    const double2 l0 = src[0], l1 = src[1], l2 = src[2];                      // x0 y0 | z0 x1 | y1 z1
    const int p0 = blockIdx.y * poses_per_block;
    const int p1 = min(p0 + poses_per_block, n_pose);
    for (int p = p0; p < p1; ++p) {
        double2 u, v, z;
        project_point(cams + 12 * (int64_t)p, l0.x, l0.y, l1.x, u.x, v.x, z.x);
        project_point(cams + 12 * (int64_t)p, l1.y, l2.x, l2.y, u.y, v.y, z.y);
        double *o = out + (int64_t)p * 3 * n + 2 * i2;
        store_stream(o, u);
        store_stream(o + n, v);
        store_stream(o + 2 * n, z);
    }
}

template <int VARIANT>
__global__ __launch_bounds__(kThreads) void depth_visibility_kernel(
    const double *__restrict__ pt_render, int64_t n, int n_pose, const float *__restrict__ depth,
    int depth_per_pose, int H, int W, int ylim, int xlim, double relax, uint8_t *__restrict__ truth) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int p = blockIdx.y;
    if (i >= n) return;
    const double *r = pt_render + (int64_t)p * 3 * n;
    const float *d = depth + (depth_per_pose ? (int64_t)p * H * W : 0);
    truth[(int64_t)p * n + i] = depth_test<VARIANT>(r[i], r[n + i], r[2 * n + i], d, H, W, ylim, xlim, relax);
}

__global__ __launch_bounds__(kThreads) void crop_remap_kernel(double *__restrict__ pt_render, int64_t n,
                                                              const int32_t *__restrict__ crop,
                                                              int crop_per_pose, int resize,
                                                              uint8_t *__restrict__ truth) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int p = blockIdx.y;
    if (i >= n) return;
    const int32_t *c = crop + (crop_per_pose ? 4 * (int64_t)p : 0);
    const CropXform cx = make_crop(c);
    double *r = pt_render + (int64_t)p * 3 * n;
    double u = r[i], v = r[n + i];
    const bool inside = crop_apply(cx, (c[1] - c[0]) >= (c[3] - c[2]), resize, u, v);
    r[i] = u;
    r[n + i] = v;
    if (truth) truth[(int64_t)p * n + i] = inside;
}

template <int VARIANT>
__global__ __launch_bounds__(kThreads) void project_prune_kernel(
    const double *__restrict__ pts, int64_t n, const double *__restrict__ cams, int n_pose,
    int poses_per_block, const float *__restrict__ depth, int depth_per_pose, int H, int W, int ylim,
    int xlim, double relax, const int32_t *__restrict__ crop, int crop_per_pose,
    double *__restrict__ out, uint8_t *__restrict__ truth, uint8_t *__restrict__ truth_depth,
    uint8_t *__restrict__ truth_crop) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int p0 = blockIdx.y * poses_per_block;
    const int p1 = min(p0 + poses_per_block, n_pose);
    for (int p = p0; p < p1; ++p) {
        double u, v, z;
        project_point(cams + 12 * (int64_t)p, px, py, pz, u, v, z);
        const float *d = depth + (depth_per_pose ? (int64_t)p * H * W : 0);
        const bool vis = depth_test<VARIANT>(u, v, z, d, H, W, ylim, xlim, relax);
        bool inside = true;
        if (crop) {
            const int32_t *c = crop + (crop_per_pose ? 4 * (int64_t)p : 0);
            const CropXform cx = make_crop(c);
            inside = crop_apply(cx, (c[1] - c[0]) >= (c[3] - c[2]), 1, u, v);
        }
        double *o = out + (int64_t)p * 3 * n;
        o[i] = u;
        o[n + i] = v;
        o[2 * n + i] = z;
        const int64_t t = (int64_t)p * n + i;
        truth[t] = vis && inside;
        if (truth_depth) truth_depth[t] = vis;
        if (truth_crop) truth_crop[t] = inside;
    }
}

template <int VARIANT>
__global__ __launch_bounds__(kThreads) void project_prune2_kernel(
    const double *__restrict__ pts, int64_t n, const double *__restrict__ cams, int n_pose,
    int poses_per_block, const float *__restrict__ depth, int depth_per_pose, int H, int W, int ylim,
    int xlim, double relax, const int32_t *__restrict__ crop, int crop_per_pose,
    double *__restrict__ out, uint8_t *__restrict__ truth, uint8_t *__restrict__ truth_depth,
    uint8_t *__restrict__ truth_crop, int nb_x) {
    // 1-D grid; blocks b and b + 8 run on the same XCD.  With one depth map per pose, all point
    // blocks of one pose group are given to one XCD, so that the depth texels gathered for those poses
    // stay in a single L2 (measured -15..-20 %); with a shared depth map the plain order is kept.
    int group, xb;
    if (depth_per_pose) {
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        group = (w / nb_x) * 8 + xcd;
        xb = w % nb_x;
    } else {
        group = blockIdx.x / nb_x;
        xb = blockIdx.x % nb_x;
    }
    const int64_t i2 = (int64_t)xb * kThreads + threadIdx.x;
    const int p0 = group * poses_per_block;
    if (2 * i2 >= n || p0 >= n_pose) return;
    const double2 *src = reinterpret_cast<const double2 *>(pts + 6 * i2);
    const double2 l0 = src[0], l1 = src[1], l2 = src[2];
    const int p1 = min(p0 + poses_per_block, n_pose);
    // crop, remap and store one pose's pair of points
    auto finish = [&](int p, double2 u, double2 v, double2 z, bool vis0, bool vis1) {
        bool in0 = true, in1 = true;
        if (crop) {
            const int32_t *c = crop + (crop_per_pose ? 4 * (int64_t)p : 0);
            const CropXform cx = make_crop(c);
            const int tall = (c[1] - c[0]) >= (c[3] - c[2]);
            in0 = crop_apply(cx, tall, 1, u.x, v.x);
            in1 = crop_apply(cx, tall, 1, u.y, v.y);
        }
        double *o = out + (int64_t)p * 3 * n + 2 * i2;
        store_stream(o, u);
        store_stream(o + n, v);
        store_stream(o + 2 * n, z);
        const int64_t t = (int64_t)p * n + 2 * i2;
        *reinterpret_cast<uchar2 *>(truth + t) = make_uchar2(vis0 && in0, vis1 && in1);
        if (truth_depth) *reinterpret_cast<uchar2 *>(truth_depth + t) = make_uchar2(vis0, vis1);
        if (truth_crop) *reinterpret_cast<uchar2 *>(truth_crop + t) = make_uchar2(in0, in1);
    };
    int p = p0;
    if (VARIANT == ISPA_VIS_A) {
        // two poses per iteration: the four depth gathers are issued together, after both
        // projections, and waited for once
        for (; p + 1 < p1; p += 2) {
            double2 ua, va, za, ub, vb, zb;
            project_point(cams + 12 * (int64_t)p, l0.x, l0.y, l1.x, ua.x, va.x, za.x);
            project_point(cams + 12 * (int64_t)p, l1.y, l2.x, l2.y, ua.y, va.y, za.y);
            project_point(cams + 12 * (int64_t)(p + 1), l0.x, l0.y, l1.x, ub.x, vb.x, zb.x);
            project_point(cams + 12 * (int64_t)(p + 1), l1.y, l2.x, l2.y, ub.y, vb.y, zb.y);
            const float *da = depth + (depth_per_pose ? (int64_t)p * H * W : 0);
            const float *db = depth + (depth_per_pose ? (int64_t)(p + 1) * H * W : 0);
            bool ga0, ga1, gb0, gb1;
            const int64_t ia0 = depth_gate_index(ua.x, va.x, W, ylim, xlim, ga0);
            const int64_t ia1 = depth_gate_index(ua.y, va.y, W, ylim, xlim, ga1);
            const int64_t ib0 = depth_gate_index(ub.x, vb.x, W, ylim, xlim, gb0);
            const int64_t ib1 = depth_gate_index(ub.y, vb.y, W, ylim, xlim, gb1);
            const float ta0 = da[ia0], ta1 = da[ia1], tb0 = db[ib0], tb1 = db[ib1];
            finish(p, ua, va, za, depth_decide_a(ga0, za.x, ta0, relax), depth_decide_a(ga1, za.y, ta1, relax));
            finish(p + 1, ub, vb, zb, depth_decide_a(gb0, zb.x, tb0, relax), depth_decide_a(gb1, zb.y, tb1, relax));
        }
    }
    for (; p < p1; ++p) {
        double2 u, v, z;
        project_point(cams + 12 * (int64_t)p, l0.x, l0.y, l1.x, u.x, v.x, z.x);
        project_point(cams + 12 * (int64_t)p, l1.y, l2.x, l2.y, u.y, v.y, z.y);
        const float *d = depth + (depth_per_pose ? (int64_t)p * H * W : 0);
        const bool vis0 = depth_test<VARIANT>(u.x, v.x, z.x, d, H, W, ylim, xlim, relax);
        const bool vis1 = depth_test<VARIANT>(u.y, v.y, z.y, d, H, W, ylim, xlim, relax);
        finish(p, u, v, z, vis0, vis1);
    }
}

int check_depth_args(const float *depth, int H, int W, int ylim, int xlim, int variant) {
    ISPA_REQUIRE(depth != nullptr, "depth map pointer is null");
    ISPA_REQUIRE(H > 0 && W > 0, "depth map shape (%d,%d) invalid", H, W);
    ISPA_REQUIRE(ylim >= 1 && ylim <= H && xlim >= 1 && xlim <= W,
                 "visibility gate (%d,%d) exceeds the depth map (%d,%d)", ylim, xlim, H, W);
    ISPA_REQUIRE(variant == ISPA_VIS_A || variant == ISPA_VIS_B, "unknown visibility variant %d", variant);
    return ISPA_OK;
}

// Poses per block: enough blocks to fill the chip, while re-using each loaded point.
int pick_poses_per_block(int64_t n, int n_pose) {
    const int64_t point_blocks = (n + kThreads - 1) / kThreads;
    int ppb = 1;
    while (ppb < 16 && point_blocks * ((n_pose + 2 * ppb - 1) / (2 * ppb)) >= 2048) ppb *= 2;
    return ppb;
}

}  // namespace

extern "C" int ispa_project(const double *pts, int64_t n, const double *cams, int n_pose, double *out,
                            void *stream) {
    ISPA_REQUIRE(n >= 0 && n_pose >= 0, "negative size");
    if (n == 0 || n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(pts && cams && out, "null pointer");
    const bool vec2 = n % 2 == 0 && ((uintptr_t)pts | (uintptr_t)out) % 16 == 0;
    const int64_t work = vec2 ? n / 2 : n;
    const int ppb = pick_poses_per_block(work, n_pose);
    dim3 grid(ispa::blocks_for(work, kThreads), (n_pose + ppb - 1) / ppb);
    ISPA_REQUIRE(grid.y <= 65535, "too many poses per call (%d)", n_pose);
    const ispa::TimingSlot timed = ispa::timing_begin((hipStream_t)stream);
    hipLaunchKernelGGL(vec2 ? project2_kernel : project_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, pts, n,
                       cams, n_pose, ppb, out);
    ispa::timing_end(timed, (hipStream_t)stream);
    return ispa::check_launch("project_kernel");
}

extern "C" int ispa_depth_visibility(const double *pt_render, int64_t n, int n_pose, const float *depth,
                                     int depth_per_pose, int H, int W, int ylim, int xlim, double relax,
                                     int variant, uint8_t *truth, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_pose >= 0, "negative size");
    if (n == 0 || n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(pt_render && truth, "null pointer");
    if (int rc = check_depth_args(depth, H, W, ylim, xlim, variant)) return rc;
    ISPA_REQUIRE(n_pose <= 65535, "too many poses per call (%d)", n_pose);
    dim3 grid(ispa::blocks_for(n, kThreads), n_pose);
    if (variant == ISPA_VIS_A)
        hipLaunchKernelGGL(depth_visibility_kernel<ISPA_VIS_A>, grid, dim3(kThreads), 0, (hipStream_t)stream,
                           pt_render, n, n_pose, depth, depth_per_pose, H, W, ylim, xlim, relax, truth);
    else
        hipLaunchKernelGGL(depth_visibility_kernel<ISPA_VIS_B>, grid, dim3(kThreads), 0, (hipStream_t)stream,
                           pt_render, n, n_pose, depth, depth_per_pose, H, W, ylim, xlim, relax, truth);
    return ispa::check_launch("depth_visibility_kernel");
}

extern "C" int ispa_crop_remap(double *pt_render, int64_t n, int n_pose, const int32_t *crop,
                               int crop_per_pose, int resize, uint8_t *truth, void *stream) {
    ISPA_REQUIRE(n >= 0 && n_pose >= 0, "negative size");
    if (n == 0 || n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(pt_render && crop, "null pointer");
    ISPA_REQUIRE(n_pose <= 65535, "too many poses per call (%d)", n_pose);
    dim3 grid(ispa::blocks_for(n, kThreads), n_pose);
    hipLaunchKernelGGL(crop_remap_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, pt_render, n, crop,
                       crop_per_pose, resize, truth);
    return ispa::check_launch("crop_remap_kernel");
}

extern "C" int ispa_project_prune(const double *pts, int64_t n, const double *cams, int n_pose,
                                  const float *depth, int depth_per_pose, int H, int W, int ylim, int xlim,
                                  double relax, int variant, const int32_t *crop, int crop_per_pose,
                                  double *out, uint8_t *truth, uint8_t *truth_depth, uint8_t *truth_crop,
                                  void *stream) {
    ISPA_REQUIRE(n >= 0 && n_pose >= 0, "negative size");
    if (n == 0 || n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(pts && cams && out && truth, "null pointer");
    if (int rc = check_depth_args(depth, H, W, ylim, xlim, variant)) return rc;
    uintptr_t align = (uintptr_t)pts | (uintptr_t)out;
    const bool vec2 = n % 2 == 0 && align % 16 == 0 && (uintptr_t)truth % 2 == 0 &&
                      (uintptr_t)truth_depth % 2 == 0 && (uintptr_t)truth_crop % 2 == 0;
    const int64_t work = vec2 ? n / 2 : n;
    const int ppb = pick_poses_per_block(work, n_pose);
    dim3 grid(ispa::blocks_for(work, kThreads), (n_pose + ppb - 1) / ppb);
    ISPA_REQUIRE(grid.y <= 65535, "too many poses per call (%d)", n_pose);
    const ispa::TimingSlot timed = ispa::timing_begin((hipStream_t)stream);
    if (vec2) {
        const int64_t nblocks = (int64_t)grid.x * (depth_per_pose ? 8 * ((grid.y + 7) / 8) : grid.y);
        ISPA_REQUIRE(nblocks < (1ll << 31), "problem too large for one launch");
        hipLaunchKernelGGL(variant == ISPA_VIS_A ? project_prune2_kernel<ISPA_VIS_A> : project_prune2_kernel<ISPA_VIS_B>,
                           dim3((unsigned)nblocks), dim3(kThreads), 0, (hipStream_t)stream, pts, n, cams, n_pose, ppb,
                           depth, depth_per_pose, H, W, ylim, xlim, relax, crop, crop_per_pose, out, truth,
                           truth_depth, truth_crop, (int)grid.x);
    } else {
        hipLaunchKernelGGL(variant == ISPA_VIS_A ? project_prune_kernel<ISPA_VIS_A> : project_prune_kernel<ISPA_VIS_B>,
                           grid, dim3(kThreads), 0, (hipStream_t)stream, pts, n, cams, n_pose, ppb, depth,
                           depth_per_pose, H, W, ylim, xlim, relax, crop, crop_per_pose, out, truth, truth_depth,
                           truth_crop);
    }
    ispa::timing_end(timed, (hipStream_t)stream);
    return ispa::check_launch("project_prune_kernel");
}
