// Correlation maps between two feature maps, plus the feature-side helpers they need.
//
//   resize      torch.nn.UpsamplingBilinear2d (align_corners=True), eccv_net.py:184-185
//   eccv        eccv_net.py:181-231: K template cells against every cell of the real image,
//               relu, divide by the spatial sum of exp, times 10
//   acmm        net_final_old.py:33-61: all-pairs correlation, relu, L2-normalise over the first
//               map's cells, multiply by the 2x2 max-pooled mask
//   l2norm      ucn_code/lib/layers/l2norm_layer.py:18-29
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ in, int64_t planes,
                                                              int H, int W, int oh, int ow, float sy, float sx,
                                                              float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = planes * oh * ow;
    if (t >= total) return;
    const int x = (int)(t % ow);
    const int y = (int)((t / ow) % oh);
    const int64_t p = t / ((int64_t)ow * oh);
    // source coordinate, lower index by truncation, weights (1-l, l): all in float like torch
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float *src = in + p * H * W;
    const float top = src[y0 * W + x0] * hx + src[y0 * W + x1] * lx;
    const float bot = src[y1 * W + x0] * hx + src[y1 * W + x1] * lx;
    out[t] = hy * top + ly * bot;
}

__device__ __forceinline__ float block_sum(float v, float *scratch) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w];
    return s;
}

// one block per (b, k): out[b][k][:] = relu(t . f1[:, p]) / sum_p exp(...) * 10
__global__ __launch_bounds__(256) void corr_eccv_kernel(const float *__restrict__ f1r, const float *__restrict__ f2r,
                                                        const int64_t *__restrict__ knn, int C, int grid, int K,
                                                        float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tmpl = sm;            // C floats
    float *scratch = sm + C;     // 4 floats
    const int k = blockIdx.x, b = blockIdx.y;
    const int cells = grid * grid;
    const int kx = (int)knn[2 * k], ky = (int)knn[2 * k + 1];
    const int64_t base = (int64_t)b * C * cells;
    for (int c = threadIdx.x; c < C; c += 256) tmpl[c] = f2r[base + (int64_t)c * cells + ky * grid + kx];
    __syncthreads();
    // pass 1: relu(correlation) to the output, sum of exp; pass 2: every thread rescales the
    // cells it wrote itself
    float *o = out + ((int64_t)b * K + k) * cells;
    float esum = 0.f;
    for (int p = threadIdx.x; p < cells; p += 256) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = fmaf(tmpl[c], f1r[base + (int64_t)c * cells + p], acc);
        acc = fmaxf(acc, 0.f);
        o[p] = acc;
        esum += expf(acc);
    }
    const float denom = block_sum(esum, scratch);
    for (int p = threadIdx.x; p < cells; p += 256) o[p] = o[p] / denom * 10.f;
}

// one block per (b, 64 columns of the second map); all rows p of the first map
__global__ __launch_bounds__(256) void corr_acmm_kernel(const float *__restrict__ f1r, const float *__restrict__ f2r,
                                                        const float *__restrict__ mask, int C, int grid, int H,
                                                        int W, float *__restrict__ out) {
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int cells = grid * grid;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;          // row group: rows p = rg, rg+4, ...
    const bool live = col < cells;
    const int64_t base = (int64_t)b * C * cells;
    float *o = out + (int64_t)b * cells * cells;
    float sumsq = 0.f;
    for (int p = rg; p < cells; p += 4) {
        float acc = 0.f;
        if (live)
            for (int c = 0; c < C; ++c)
                acc = fmaf(f1r[base + (int64_t)c * cells + p], f2r[base + (int64_t)c * cells + col], acc);
        acc = fmaxf(acc, 0.f);
        sumsq = fmaf(acc, acc, sumsq);
        if (live) o[(int64_t)p * cells + col] = acc;
    }
    part[rg][threadIdx.x & 63] = sumsq;
    __syncthreads();
    const float norm = sqrtf(part[0][threadIdx.x & 63] + part[1][threadIdx.x & 63] + part[2][threadIdx.x & 63] +
                             part[3][threadIdx.x & 63]);
    if (!live) return;
    // max_pool2d(mask, 2, 2, ceil_mode=True) at this column
    const int y = col / grid, x = col % grid;
    const float *m = mask + (int64_t)b * H * W;
    float mv = -__builtin_inff();
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int yy = 2 * y + dy, xx = 2 * x + dx;
            if (yy < H && xx < W) mv = fmaxf(mv, m[yy * W + xx]);
        }
    // each thread re-reads only what it wrote itself
    for (int p = rg; p < cells; p += 4) {
        const int64_t a = (int64_t)p * cells + col;
        o[a] = o[a] / norm * mv;
    }
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float *__restrict__ x, int C, int64_t hw,
                                                     float *__restrict__ out) {
    const int b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const float *src = x + (int64_t)b * C * hw + p;
    float *dst = out + (int64_t)b * C * hw + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = src[(int64_t)c * hw];
        s = fmaf(v, v, s);
    }
    const float n = sqrtf(s);
    for (int c = 0; c < C; ++c) dst[(int64_t)c * hw] = src[(int64_t)c * hw] / n;
}

int launch_resize(const float *in, int64_t planes, int H, int W, int oh, int ow, float *out, hipStream_t s) {
    const float sy = oh > 1 ? (float)(H - 1) / (float)(oh - 1) : 0.f;
    const float sx = ow > 1 ? (float)(W - 1) / (float)(ow - 1) : 0.f;
    const int64_t total = planes * oh * ow;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ispa::blocks_for(total, 256, 1ll << 30)), dim3(256), 0, s, in,
                       planes, H, W, oh, ow, sy, sx, out);
    return ispa::check_launch("resize_bilinear_kernel");
}

}  // namespace

extern "C" int ispa_resize_bilinear(const float *in, int B, int C, int H, int W, int oh, int ow, float *out,
                                    void *stream) {
    ISPA_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "bad shape");
    if (B == 0 || C == 0) return ISPA_OK;
    ISPA_REQUIRE(in && out, "null pointer");
    return launch_resize(in, (int64_t)B * C, H, W, oh, ow, out, (hipStream_t)stream);
}

extern "C" int ispa_corr_map_eccv(const float *f1, const float *f2, const int64_t *knn, int B, int C, int H,
                                  int W, int K, int grid, float *out, float *workspace, void *stream) {
    ISPA_REQUIRE(B >= 0 && K >= 0 && C > 0 && H > 0 && W > 0 && grid > 0, "bad shape");
    ISPA_REQUIRE(grid <= 4096, "grid %d too large", grid);
    ISPA_REQUIRE(C <= 8192 && B <= 65535, "C or B too large");
    if (B == 0 || K == 0) return ISPA_OK;
    ISPA_REQUIRE(f1 && f2 && knn && out && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    float *f1r = workspace, *f2r = workspace + (int64_t)B * C * grid * grid;
    if (int rc = launch_resize(f1, (int64_t)B * C, H, W, grid, grid, f1r, s)) return rc;
    if (int rc = launch_resize(f2, (int64_t)B * C, H, W, grid, grid, f2r, s)) return rc;
    hipLaunchKernelGGL(corr_eccv_kernel, dim3(K, B), dim3(256), (C + 4) * sizeof(float), s, f1r, f2r, knn, C, grid,
                       K, out);
    return ispa::check_launch("corr_eccv_kernel");
}

extern "C" int ispa_corr_map_acmm(const float *f1, const float *f2, const float *mask, int B, int C, int H,
                                  int W, int grid, float *out, float *workspace, void *stream) {
    ISPA_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && grid > 0, "bad shape");
    ISPA_REQUIRE((H + 1) / 2 == grid && (W + 1) / 2 == grid,
                 "mask pooled 2x2 from (%d,%d) does not give a %d x %d grid", H, W, grid, grid);
    ISPA_REQUIRE(B <= 65535, "B too large");
    if (B == 0) return ISPA_OK;
    ISPA_REQUIRE(f1 && f2 && mask && out && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int cells = grid * grid;
    float *f1r = workspace, *f2r = workspace + (int64_t)B * C * cells;
    if (int rc = launch_resize(f1, (int64_t)B * C, H, W, grid, grid, f1r, s)) return rc;
    if (int rc = launch_resize(f2, (int64_t)B * C, H, W, grid, grid, f2r, s)) return rc;
    hipLaunchKernelGGL(corr_acmm_kernel, dim3((cells + 63) / 64, B), dim3(256), 0, s, f1r, f2r, mask, C, grid, H, W,
                       out);
    return ispa::check_launch("corr_acmm_kernel");
}

extern "C" int ispa_l2norm(const float *x, int B, int C, int64_t hw, float *out, void *stream) {
    ISPA_REQUIRE(B >= 0 && C > 0 && hw >= 0 && B <= 65535, "bad shape");
    if (B == 0 || hw == 0) return ISPA_OK;
    ISPA_REQUIRE(x && out, "null pointer");
    hipLaunchKernelGGL(l2norm_kernel, dim3(ispa::blocks_for(hw, 256), B), dim3(256), 0, (hipStream_t)stream, x, C,
                       hw, out);
    return ispa::check_launch("l2norm_kernel");
}
