// Pruning of dense 2-D skeleton keypoints on annotated real images, batched over images:
//   data_creator/utils.py:274-290   surface_pruned  (points of the leg bones inside the seat outline)
//   data_creator/utils.py:292-325   dir_pruned      (points under the thick lines of the far-side bones)
//   data_creator/utils.py:327-348   surface_dir_prune (whole bones dropped by facing direction)
// The reference decides the first two through OpenCV rasterisers (polylines + findContours +
// pointPolygonTest; line of thickness 5 + pixel lookup).  Here both are closed-form per point:
// inside-or-on the polygon by crossing parity with float64 cross products, and distance from the
// truncated pixel to the segment in exact integer arithmetic.  One thread per (image, point); the
// few polygon corners and segments of an image sit in LDS.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCoord = 8192;       // image sides and |corner / end point coordinates| the kernel accepts

__global__ __launch_bounds__(kThreads) void prune_points_2d_kernel(
    double *__restrict__ coords, int64_t n, const uint8_t *__restrict__ flags, const int32_t *__restrict__ polygon,
    int n_vert, const int32_t *__restrict__ segments, const int32_t *__restrict__ n_seg, int max_seg, int thickness,
    int H, int W, uint8_t *__restrict__ pruned) {
    extern __shared__ int32_t sh[];                      // 2 * n_vert polygon ints, then 4 * max_seg segment ints
    const int b = blockIdx.y;
    int32_t *poly = sh, *seg = sh + 2 * n_vert;
    const int ns = n_seg ? min(max(n_seg[b], 0), max_seg) : 0;
    // corners and end points clamped to +-kMaxCoord: the int64 distance products below cannot overflow
    for (int t = threadIdx.x; t < 2 * n_vert; t += kThreads)
        poly[t] = min(max(polygon[(int64_t)b * 2 * n_vert + t], -kMaxCoord), kMaxCoord);
    for (int t = threadIdx.x; t < 4 * ns; t += kThreads)
        seg[t] = min(max(segments[(int64_t)b * 4 * max_seg + t], -kMaxCoord), kMaxCoord);
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (m >= n) return;
    const int f = flags[(int64_t)b * n + m];
    double *cx = coords + (int64_t)b * 2 * n + m, *cy = cx + n;
    const double px = *cx, py = *cy;
    bool drop = (f & ISPA_PRUNE_DROP) != 0;

    if ((f & ISPA_PRUNE_POLYGON) && n_vert > 0) {
        bool parity = false, on_edge = false;
        for (int e = 0; e < n_vert; ++e) {
            const int e1 = e + 1 == n_vert ? 0 : e + 1;
            const int ax = poly[2 * e], ay = poly[2 * e + 1], bx = poly[2 * e1], by = poly[2 * e1 + 1];
            // two rounded products and one rounded difference, as numpy evaluates the expression
            const double cross = (double)(bx - ax) * (py - (double)ay) - (double)(by - ay) * (px - (double)ax);
            if (cross == 0.0 && (double)min(ax, bx) <= px && px <= (double)max(ax, bx) &&
                (double)min(ay, by) <= py && py <= (double)max(ay, by))
                on_edge = true;
            if (((double)ay > py) != ((double)by > py))
                if ((by > ay && cross > 0.0) || (by < ay && cross < 0.0)) parity = !parity;
        }
        drop = drop || on_edge || parity;
    }

    if ((f & ISPA_PRUNE_BAND) && ns > 0) {
        // ndarray.astype(int) truncates toward zero; a negative index wraps once, as Python's does
        int ix = (int)px, iy = (int)py;
        if (ix < 0) ix += W;
        if (iy < 0) iy += H;
        if (ix >= 0 && ix < W && iy >= 0 && iy < H) {
            const int64_t t2 = (int64_t)thickness * thickness;
            for (int s = 0; s < ns; ++s) {
                const int64_t x0 = seg[4 * s], y0 = seg[4 * s + 1], x1 = seg[4 * s + 2], y1 = seg[4 * s + 3];
                const int64_t dx = x1 - x0, dy = y1 - y0, ux = ix - x0, uy = iy - y0;
                const int64_t l2 = dx * dx + dy * dy, dot = ux * dx + uy * dy;
                bool hit;
                if (l2 == 0 || dot <= 0) hit = 4 * (ux * ux + uy * uy) <= t2;
                else if (dot >= l2) hit = 4 * ((ix - x1) * (ix - x1) + (iy - y1) * (iy - y1)) <= t2;
                else {
                    const int64_t cr = ux * dy - uy * dx;
                    hit = 4 * cr * cr <= t2 * l2;
                }
                drop = drop || hit;
            }
        }
    }
    if (drop) {
        *cx = 0.0;
        *cy = 0.0;
    }
    if (pruned) pruned[(int64_t)b * n + m] = drop ? 1 : 0;
}

}  // namespace

extern "C" int ispa_prune_points_2d(double *coords, int B, int64_t n, const uint8_t *flags, const int32_t *polygon,
                                    int n_vert, const int32_t *segments, const int32_t *n_seg, int max_seg,
                                    int thickness, int H, int W, uint8_t *pruned, void *stream) {
    ISPA_REQUIRE(B >= 0 && n >= 0, "negative size");
    ISPA_REQUIRE(n_vert >= 0 && n_vert <= ISPA_PRUNE_MAX_VERTICES, "polygon of %d vertices (at most %d)", n_vert,
                 ISPA_PRUNE_MAX_VERTICES);
    ISPA_REQUIRE(max_seg >= 0 && max_seg <= ISPA_PRUNE_MAX_SEGMENTS, "%d segments per image (at most %d)", max_seg,
                 ISPA_PRUNE_MAX_SEGMENTS);
    ISPA_REQUIRE(thickness >= 1 && thickness <= 255, "line thickness %d outside [1,255]", thickness);
    ISPA_REQUIRE(H > 0 && W > 0 && H <= kMaxCoord && W <= kMaxCoord, "image size (%d,%d) outside (0,%d]", H, W,
                 kMaxCoord);
    ISPA_REQUIRE(B <= 65535, "too many images per call (%d)", B);
    if (B == 0 || n == 0) return ISPA_OK;
    ISPA_REQUIRE(coords && flags, "null pointer");
    ISPA_REQUIRE(n_vert == 0 || polygon, "polygon pointer is null");
    ISPA_REQUIRE(max_seg == 0 || (segments && n_seg), "segment pointers are null");
    const size_t lds = (size_t)(2 * n_vert + 4 * max_seg) * sizeof(int32_t);
    hipLaunchKernelGGL(prune_points_2d_kernel, dim3(ispa::blocks_for(n, kThreads), B), dim3(kThreads), lds,
                       (hipStream_t)stream, coords, n, flags, polygon, n_vert, segments, n_seg, max_seg, thickness, H,
                       W, pruned);
    return ispa::check_launch("prune_points_2d_kernel");
}
