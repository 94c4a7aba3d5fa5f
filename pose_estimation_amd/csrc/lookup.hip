// Consumers of the correspondence stage that the reference runs as numpy one-liners per item:
//   proxy_renderer.py:100-106,138-151   nearest stored render to a requested pose
//   hard_negative_processing_layer.py:44-58,188-190,218-234   1-NN index -> image coordinate -> PCK hit counts
//   ucn_code/nets/train.prototxt:2116-2127   ExtractFeature: query features at the ground-truth coordinates
// Batched here so that neither the stored-pose table nor the index maps leave the device.
// Arithmetic follows numpy operation by operation (float64, separately rounded, sums left to right,
// correctly rounded sqrt), so indices and counts are bit-exact.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// angle difference wrapped as proxy_renderer.normalize does: >= half -> minus a turn, < -half -> plus
__device__ __forceinline__ double wrap(double d, double half) {
    if (d >= half) return d - 2.0 * half;
    if (d < -half) return d + 2.0 * half;
    return d;
}

__device__ __forceinline__ bool before(double d0, int j0, double d1, int j1) {
    // np.argmin: the first minimum; a NaN is the minimum as soon as it appears
    const bool n0 = d0 != d0, n1 = d1 != d1;
    if (n0 || n1) return n0 && (!n1 || j0 < j1);
    return d0 < d1 || (d0 == d1 && j0 < j1);
}

// one block per requested pose, threads stride over the stored poses
__global__ __launch_bounds__(kThreads) void nearest_view_kernel(const double *__restrict__ view,
                                                                 const double *__restrict__ stored, int n_stored,
                                                                 int32_t *__restrict__ index,
                                                                 double *__restrict__ distance) {
    __shared__ double sd[kThreads];
    __shared__ int sj[kThreads];
    const int b = blockIdx.x;
    const double v0 = view[3 * b], v1 = view[3 * b + 1], v2 = view[3 * b + 2];
    double best = __builtin_inf();
    int best_j = 0x7fffffff;
    for (int j = threadIdx.x; j < n_stored; j += kThreads) {
        const double d0 = wrap(stored[3 * j] - v0, 180.0);
        const double d1 = wrap(stored[3 * j + 1] - v1, 90.0) / 8.0;      // azimuth has priority
        const double d2 = wrap(stored[3 * j + 2] - v2, 180.0) / 8.0;
        const double dist = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        if (before(dist, j, best, best_j)) {
            best = dist;
            best_j = j;
        }
    }
    sd[threadIdx.x] = best;
    sj[threadIdx.x] = best_j;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const int o = threadIdx.x + off;
            if (before(sd[o], sj[o], sd[threadIdx.x], sj[threadIdx.x])) {
                sd[threadIdx.x] = sd[o];
                sj[threadIdx.x] = sj[o];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        index[b] = sj[0];
        if (distance) distance[b] = sd[0];
    }
}

// one block per batch item: counts[r * B + b] = #{valid rows whose 1-NN lands within radii[r]}
__global__ __launch_bounds__(kThreads) void pck_counts_kernel(const int32_t *__restrict__ ind, int k, int64_t n,
                                                               const double *__restrict__ corr, int width,
                                                               double downsampling, int index_base,
                                                               const double *__restrict__ radii, int n_radii, int B,
                                                               int32_t *__restrict__ counts) {
    __shared__ int hits[ISPA_PCK_MAX_RADII];
    const int b = blockIdx.x;
    if (threadIdx.x < ISPA_PCK_MAX_RADII) hits[threadIdx.x] = 0;
    __syncthreads();
    int mine[ISPA_PCK_MAX_RADII];
#pragma unroll
    for (int r = 0; r < ISPA_PCK_MAX_RADII; ++r) mine[r] = 0;
    const int32_t *row = ind + (int64_t)b * k * n;                       // nearest neighbour = row 0 of k
    for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const double *c = corr + ((int64_t)b * n + i) * 5;
        if (!(c[4] == 1.0)) continue;
        const int j = row[i] - index_base;
        // Python's % and floor(/): results for a negative index (no neighbour) follow the sign rules
        int x = j % width;
        if (x < 0) x += width;
        const double y = floor((double)j / (double)width);
        const double dx = (double)x * downsampling - c[2];
        const double dy = y * downsampling - c[3];
        const double dist = sqrt(dx * dx + dy * dy);
#pragma unroll
        for (int r = 0; r < ISPA_PCK_MAX_RADII; ++r)
            if (r < n_radii && dist < radii[r]) ++mine[r];
    }
#pragma unroll
    for (int r = 0; r < ISPA_PCK_MAX_RADII; ++r)
        if (r < n_radii && mine[r]) atomicAdd(&hits[r], mine[r]);
    __syncthreads();
    if (threadIdx.x < n_radii) counts[threadIdx.x * B + b] = hits[threadIdx.x];
}

// Per correspondence row: is the 1-NN a hard negative (a valid row whose nearest cell lies farther than `thresh`
// FEATURE cells from the ground truth), and the image coordinate of that nearest cell.  numpy's order of operations:
// ground truth / downsampling, differences, squares added left to right, correctly rounded sqrt.
__global__ __launch_bounds__(kThreads) void hard_negative_kernel(const int32_t *__restrict__ ind, int k, int64_t n,
                                                                  const double *__restrict__ corr, int width,
                                                                  double downsampling, int index_base, double thresh,
                                                                  uint8_t *__restrict__ mask, double *__restrict__ knn_xy) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const double *c = corr + ((int64_t)b * n + i) * 5;
    const int j = ind[(int64_t)b * k * n + i] - index_base;              // nearest neighbour = row 0 of k
    int x = j % width;
    if (x < 0) x += width;
    const double y = floor((double)j / (double)width);
    const double dx = (double)x - c[2] / downsampling;
    const double dy = y - c[3] / downsampling;
    const double dist = sqrt(dx * dx + dy * dy);
    mask[(int64_t)b * n + i] = (c[4] == 1.0) && dist > thresh;
    if (knn_xy) {
        double2 *o = reinterpret_cast<double2 *>(knn_xy) + (int64_t)b * n + i;
        *o = make_double2((double)x * downsampling, y * downsampling);
    }
}

// out[b][c][i] = feature[b][c] sampled at (x, y) = coords[b][i][start .. start+1] / downsampling.
// One thread per (b, i); the channel loop walks a column of the (C, H*W) map (stride H*W), the
// stores of a wave are contiguous along i.
__global__ __launch_bounds__(kThreads) void extract_feature_kernel(const float *__restrict__ feature, int C, int H,
                                                                    int W, const float *__restrict__ coords,
                                                                    int64_t n, int ncol, int start, float ds,
                                                                    int bilinear, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float *c = coords + ((int64_t)b * n + i) * ncol + start;
    const float fx = c[0] / ds, fy = c[1] / ds;
    const float x0f = floorf(fx), y0f = floorf(fy);
    // clamp in float first: NaN and out-of-range coordinates land on an edge cell
    const int x0 = (int)fminf(fmaxf(x0f, 0.f), (float)(W - 1)), y0 = (int)fminf(fmaxf(y0f, 0.f), (float)(H - 1));
    const int x1 = (int)fminf(fmaxf(x0f + 1.f, 0.f), (float)(W - 1)), y1 = (int)fminf(fmaxf(y0f + 1.f, 0.f), (float)(H - 1));
    const float wx = fx - x0f, wy = fy - y0f;
    const int64_t hw = (int64_t)H * W;
    const float *f = feature + (int64_t)b * C * hw;
    float *o = out + (int64_t)b * C * n + i;
    for (int ch = 0; ch < C; ++ch, f += hw, o += n) {
        if (!bilinear) {
            *o = f[(int64_t)y0 * W + x0];
        } else {
            const float f00 = f[(int64_t)y0 * W + x0], f01 = f[(int64_t)y0 * W + x1];
            const float f10 = f[(int64_t)y1 * W + x0], f11 = f[(int64_t)y1 * W + x1];
            const float top = (1.f - wx) * f00 + wx * f01;      // separately rounded (no contraction)
            const float bot = (1.f - wx) * f10 + wx * f11;
            *o = (1.f - wy) * top + wy * bot;
        }
    }
}

}  // namespace

extern "C" int ispa_extract_feature(const float *feature, int B, int C, int H, int W, const float *coords, int64_t n,
                                    int num_coord_elements, int coord_start_ind, float downsampling, int bilinear,
                                    float *out, void *stream) {
    ISPA_REQUIRE(B >= 0 && n >= 0, "negative size");
    ISPA_REQUIRE(C > 0 && H > 0 && W > 0, "feature map shape (%d,%d,%d) invalid", C, H, W);
    ISPA_REQUIRE(coord_start_ind >= 0 && coord_start_ind + 2 <= num_coord_elements,
                 "coordinate columns %d..%d outside the %d-element rows", coord_start_ind, coord_start_ind + 1,
                 num_coord_elements);
    ISPA_REQUIRE(downsampling > 0.f, "downsampling factor must be positive");
    ISPA_REQUIRE(B <= 65535, "too many batch items per call (%d)", B);
    if (B == 0 || n == 0) return ISPA_OK;
    ISPA_REQUIRE(feature && coords && out, "null pointer");
    hipLaunchKernelGGL(extract_feature_kernel, dim3(ispa::blocks_for(n, kThreads), B), dim3(kThreads), 0,
                       (hipStream_t)stream, feature, C, H, W, coords, n, num_coord_elements, coord_start_ind,
                       downsampling, bilinear, out);
    return ispa::check_launch("extract_feature_kernel");
}

extern "C" int ispa_nearest_view(const double *view_params, int n_view, const double *stored, int n_stored,
                                 int32_t *index, double *distance, void *stream) {
    ISPA_REQUIRE(n_view >= 0 && n_stored >= 0, "negative size");
    if (n_view == 0) return ISPA_OK;
    ISPA_REQUIRE(n_stored > 0, "no stored poses to choose from");
    ISPA_REQUIRE(view_params && stored && index, "null pointer");
    hipLaunchKernelGGL(nearest_view_kernel, dim3(n_view), dim3(kThreads), 0, (hipStream_t)stream, view_params, stored,
                       n_stored, index, distance);
    return ispa::check_launch("nearest_view_kernel");
}

extern "C" int ispa_pck_counts(const int32_t *ind, int B, int k, int64_t n, const double *correspondences, int width,
                               double downsampling, int index_base, const double *radii, int n_radii,
                               int32_t *counts, void *stream) {
    ISPA_REQUIRE(B >= 0 && n >= 0, "negative size");
    ISPA_REQUIRE(k >= 1 && width >= 1, "k (%d) and width (%d) must be positive", k, width);
    ISPA_REQUIRE(n_radii >= 0 && n_radii <= ISPA_PCK_MAX_RADII, "at most %d radii per call (got %d)",
                 ISPA_PCK_MAX_RADII, n_radii);
    if (B == 0 || n_radii == 0) return ISPA_OK;
    ISPA_REQUIRE(counts && radii && (n == 0 || (ind && correspondences)), "null pointer");
    hipLaunchKernelGGL(pck_counts_kernel, dim3(B), dim3(kThreads), 0, (hipStream_t)stream, ind, k, n, correspondences,
                       width, downsampling, index_base, radii, n_radii, B, counts);
    return ispa::check_launch("pck_counts_kernel");
}

extern "C" int ispa_hard_negatives(const int32_t *ind, int B, int k, int64_t n, const double *correspondences, int width,
                                   double downsampling, int index_base, double thresh, uint8_t *mask, double *knn_xy,
                                   void *stream) {
    ISPA_REQUIRE(B >= 0 && n >= 0, "negative size");
    ISPA_REQUIRE(k >= 1 && width >= 1, "k (%d) and width (%d) must be positive", k, width);
    ISPA_REQUIRE(downsampling > 0.0, "downsampling factor must be positive");
    if (B == 0 || n == 0) return ISPA_OK;
    ISPA_REQUIRE(B <= 65535, "too many batch items per call (%d)", B);
    ISPA_REQUIRE(ind && correspondences && mask, "null pointer");
    hipLaunchKernelGGL(hard_negative_kernel, dim3(ispa::blocks_for(n, kThreads), B), dim3(kThreads), 0,
                       (hipStream_t)stream, ind, k, n, correspondences, width, downsampling, index_base, thresh, mask,
                       knn_xy);
    return ispa::check_launch("hard_negative_kernel");
}
