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

// torch.nn.UpsamplingBilinear2d (align_corners=True) read straight from the source map: the value of cell
// (y, x) of the (oh, ow) grid.  Source coordinate, lower index by truncation, weights (1-l, l), rows combined
// after columns: all in float, the operation order of resize_bilinear_kernel (same bits).  A map already at
// the grid's size has scale 1: weights exactly 1 and 0.
struct Resample {
    int H, W;
    float sy, sx;
};
__host__ __device__ inline Resample make_resample(int H, int W, int oh, int ow) {
    Resample r;
    r.H = H;
    r.W = W;
    r.sy = oh > 1 ? (float)(H - 1) / (float)(oh - 1) : 0.f;
    r.sx = ow > 1 ? (float)(W - 1) / (float)(ow - 1) : 0.f;
    return r;
}
struct Taps {                 // the four source offsets and the weights of one grid cell
    int o00, o01, o10, o11;
    float hx, lx, hy, ly;
};
__device__ __forceinline__ Taps make_taps(const Resample &r, int y, int x) {
    const float fy = r.sy * (float)y, fx = r.sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, r.H - 1), x1 = min(x0 + 1, r.W - 1);
    Taps t;
    t.ly = fy - (float)y0;
    t.lx = fx - (float)x0;
    t.hy = 1.f - t.ly;
    t.hx = 1.f - t.lx;
    t.o00 = y0 * r.W + x0;
    t.o01 = y0 * r.W + x1;
    t.o10 = y1 * r.W + x0;
    t.o11 = y1 * r.W + x1;
    return t;
}
__device__ __forceinline__ float sample(const float *__restrict__ plane, const Taps &t) {
    const float top = plane[t.o00] * t.hx + plane[t.o01] * t.lx;
    const float bot = plane[t.o10] * t.hx + plane[t.o11] * t.lx;
    return t.hy * top + t.ly * bot;
}

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

// Both feature maps of a correlation call to the grid's size in one launch (eccv_net.py:184-185 /
// net_final_old.py:33-34): thread = one grid cell of one (item, channel) plane of either map.
__global__ __launch_bounds__(256) void resample2_kernel(const float *__restrict__ in1, const float *__restrict__ in2,
                                                        int64_t planes, int grid, Resample rs,
                                                        float *__restrict__ out1, float *__restrict__ out2) {
    const int cells = grid * grid;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= planes * cells) return;
    const int cell = (int)(t % cells);
    const int64_t p = t / cells;
    const Taps tp = make_taps(rs, cell / grid, cell % grid);
    const float *src = (blockIdx.y ? in2 : in1) + p * rs.H * rs.W;
    (blockIdx.y ? out2 : out1)[t] = sample(src, tp);
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

// one block per (b, k): out[b][k][:] = relu(t . f1[:, p]) / sum_p exp(...) * 10, the maps resampled to the
// grid while they are read (item b of a (B, views, ...) batch uses the keypoint list of view b % views;
// a keypoint off the grid gives a template of zeros: the Python layer raises IndexError before it gets here)
__global__ __launch_bounds__(256) void corr_eccv_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                        const int64_t *__restrict__ knn, int views, int C, int grid,
                                                        Resample rs, int K, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tmpl = sm;            // C floats
    float *scratch = sm + C;     // 4 floats
    const int k = blockIdx.x, b = blockIdx.y;
    const int cells = grid * grid;
    const int64_t plane = (int64_t)rs.H * rs.W;
    const int64_t *kn = knn + (int64_t)(b % views) * K * 2;
    const int64_t kx = kn[2 * k], ky = kn[2 * k + 1];
    const bool on_grid = kx >= 0 && kx < grid && ky >= 0 && ky < grid;
    const int64_t base = (int64_t)b * C * plane;
    const Taps tk = make_taps(rs, on_grid ? (int)ky : 0, on_grid ? (int)kx : 0);
    for (int c = threadIdx.x; c < C; c += 256) tmpl[c] = on_grid ? sample(f2 + base + (int64_t)c * plane, tk) : 0.f;
    __syncthreads();
    // pass 1: relu(correlation) to the output, sum of exp; pass 2: every thread rescales the
    // cells it wrote itself
    float *o = out + ((int64_t)b * K + k) * cells;
    float esum = 0.f;
    for (int p = threadIdx.x; p < cells; p += 256) {
        float acc = 0.f;
        const Taps tp = make_taps(rs, p / grid, p % grid);
        const float *col = f1 + base;
        int c = 0;
        for (; c + 8 <= C; c += 8) {                 // 8 cells' loads in flight, then the ordered chain
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sample(col + (int64_t)(c + u) * plane, tp);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(tmpl[c + u], v[u], acc);
        }
        for (; c < C; ++c) acc = fmaf(tmpl[c], sample(col + (int64_t)c * plane, tp), acc);
        acc = fmaxf(acc, 0.f);
        o[p] = acc;
        esum += expf(acc);
    }
    const float denom = block_sum(esum, scratch);
    for (int p = threadIdx.x; p < cells; p += 256) o[p] = o[p] / denom * 10.f;
}

// Up to 16 keypoints: corr_eccv_kernel would resample and stream the whole first map once per keypoint.
// Here a block owns 64 cells and all K <= 16 keypoints, so the map is read (and resampled) once, by many
// CUs; its four waves each take a quarter of the channels (a cell's correlation is then the sum of four
// partial fmaf chains, added in wave order: deterministic, last-bit different from corr_eccv_kernel's
// single chain); the per-(b, k) normalisation runs afterwards (corr_eccv_scale_kernel).
constexpr int kEccvMaxK = 16;
constexpr int kEccvCells = 64;
__global__ __launch_bounds__(256) void corr_eccv_tile_kernel(const float *__restrict__ f1,
                                                             const float *__restrict__ f2,
                                                             const int64_t *__restrict__ knn, int views, int C,
                                                             int grid, Resample rs, int K, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];           // tm[c][16]: keypoint kk's channel c
    __shared__ float red[3][kEccvMaxK][kEccvCells];                      // partial sums of waves 1..3
    const int b = blockIdx.y;
    const int cells = grid * grid;
    const int64_t plane = (int64_t)rs.H * rs.W;
    const int64_t base = (int64_t)b * C * plane;
    const int64_t *kn = knn + (int64_t)(b % views) * K * 2;
    {
        // thread t fills tm[c][kk] for kk = t & 15: its keypoint's taps are computed once
        const int kk = threadIdx.x & 15;
        const int64_t kx = kk < K ? kn[2 * kk] : -1, ky = kk < K ? kn[2 * kk + 1] : -1;
        const bool on_grid = kx >= 0 && kx < grid && ky >= 0 && ky < grid;
        const Taps tk = make_taps(rs, on_grid ? (int)ky : 0, on_grid ? (int)kx : 0);
        for (int t = threadIdx.x; t < C * kEccvMaxK; t += 256)
            sm[t] = on_grid ? sample(f2 + base + (int64_t)(t >> 4) * plane, tk) : 0.f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * kEccvCells + lane;
    const bool live = p < cells;
    float acc[kEccvMaxK];
#pragma unroll
    for (int kk = 0; kk < kEccvMaxK; ++kk) acc[kk] = 0.f;
    const Taps tp = make_taps(rs, (live ? p : 0) / grid, (live ? p : 0) % grid);
    const float *col = f1 + base;
    const float4 *tm4 = reinterpret_cast<const float4 *>(sm);
    const int groups = (K + 3) >> 2;
    auto channel = [&](int c, float v) {
#pragma unroll
        for (int g = 0; g < kEccvMaxK / 4; ++g) {
            if (g < groups) {
                const float4 t = tm4[c * 4 + g];
                acc[4 * g + 0] = fmaf(t.x, v, acc[4 * g + 0]);
                acc[4 * g + 1] = fmaf(t.y, v, acc[4 * g + 1]);
                acc[4 * g + 2] = fmaf(t.z, v, acc[4 * g + 2]);
                acc[4 * g + 3] = fmaf(t.w, v, acc[4 * g + 3]);
            }
        }
    };
    const int per = (C + 3) / 4;
    const int c_end = min(C, (wave + 1) * per);
    int c = wave * per;
    for (; c + 8 <= c_end; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sample(col + (int64_t)(c + u) * plane, tp);
#pragma unroll
        for (int u = 0; u < 8; ++u) channel(c + u, v[u]);
    }
    for (; c < c_end; ++c) channel(c, sample(col + (int64_t)c * plane, tp));
    if (wave > 0)
#pragma unroll
        for (int kk = 0; kk < kEccvMaxK; ++kk) red[wave - 1][kk][lane] = acc[kk];
    __syncthreads();
    if (wave == 0 && live)
#pragma unroll
        for (int kk = 0; kk < kEccvMaxK; ++kk)
            if (kk < K) {
                const float v = ((acc[kk] + red[0][kk][lane]) + red[1][kk][lane]) + red[2][kk][lane];
                out[((int64_t)b * K + kk) * cells + p] = fmaxf(v, 0.f);
            }
}

// one block per (b, k): out[b][k][:] (relu'd correlations) / sum exp * 10, as corr_eccv_kernel ends
__global__ __launch_bounds__(256) void corr_eccv_scale_kernel(float *__restrict__ out, int cells, int K) {
    __shared__ float scratch[4];
    float *o = out + ((int64_t)blockIdx.y * K + blockIdx.x) * cells;
    float esum = 0.f;
    for (int p = threadIdx.x; p < cells; p += 256) esum += expf(o[p]);
    const float denom = block_sum(esum, scratch);
    for (int p = threadIdx.x; p < cells; p += 256) o[p] = o[p] / denom * 10.f;
}

// one block per (b, 64 columns of the second map); all rows p of the first map (grids whose cell count is
// not a multiple of 4: the MFMA kernel below takes the others); the maps are resampled while they are read
__global__ __launch_bounds__(256) void corr_acmm_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                        const float *__restrict__ mask, int C, int grid, Resample rs,
                                                        float *__restrict__ out) {
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int cells = grid * grid;
    const int H = rs.H, W = rs.W;
    const int64_t plane = (int64_t)H * W;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;          // row group: rows p = rg, rg+4, ...
    const bool live = col < cells;
    const int64_t base = (int64_t)b * C * plane;
    const Taps tc = make_taps(rs, (live ? col : 0) / grid, (live ? col : 0) % grid);
    float *o = out + (int64_t)b * cells * cells;
    float sumsq = 0.f;
    for (int p = rg; p < cells; p += 4) {
        float acc = 0.f;
        const Taps tp = make_taps(rs, p / grid, p % grid);
        if (live)
            for (int c = 0; c < C; ++c)
                acc = fmaf(sample(f1 + base + (int64_t)c * plane, tp), sample(f2 + base + (int64_t)c * plane, tc), acc);
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

// ---- acmm correlation as an MFMA GEMM ------------------------------------------------------------
// raw[b][p][yx] = relu(sum_c f1[b][c][p] * f2[b][c][yx]): a (cells x C) x (C x cells) product per
// batch item on v_mfma_f32_32x32x2_f32.  Both operands are channel-major, which is the operand
// order the instruction wants (lane = cell, k = channel): no transposes.  128 x 128 output tile
// per 256-thread block, 4 waves as 2 (rows p) x 2 (columns yx) of 64 x 64.  Every block also
// writes the column sums of squares of its rows to a partial buffer; a second kernel adds the
// partials in a fixed order (deterministic), normalises and applies the pooled mask.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kCK = 16;    // channels per LDS stage

// Operands: maps already at the grid's size (resample2_kernel below brings them there in one launch for
// both maps).  Resampling inside this loader was measured and dropped: in an all-pairs product every
// operand element is staged by 7 tiles at the reference's 28 x 28 grid, so the 4-tap gather was done 7
// times over (83 us against 30 us + one 9 us resampling launch).
// WT = 32 x 32 MFMA tiles per wave and direction: the 2 x 2 waves of a block cover a (64 WT)^2 tile.  WT = 2
// for large grids; WT = 1 at the reference's 28 x 28 grid, where 784 cells are 6.1 tiles of 128 (a quarter
// of a 7 x 7 tile grid would multiply zeros) and 392 blocks are a round and a half on 256 CUs.
template <int WT>
__global__ __launch_bounds__(256) void corr_acmm_gemm_kernel(const float *__restrict__ f1r,
                                                             const float *__restrict__ f2r, int C, int cells,
                                                             float *__restrict__ out,
                                                             float *__restrict__ colsq /* (B, 2*ptiles, cells) */) {
    constexpr int kT = 64 * WT;                                   // tile edge of the block
    __shared__ __attribute__((aligned(16))) float As[kCK][kT];    // f1 tile: rows p
    __shared__ __attribute__((aligned(16))) float Bs[kCK][kT];    // f2 tile: columns yx
    const int b = blockIdx.z;
    const int p0 = blockIdx.y * kT, x0 = blockIdx.x * kT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const float *a = f1r + (int64_t)b * C * cells;
    const float *bb = f2r + (int64_t)b * C * cells;
    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int c0 = 0; c0 < C; c0 += kCK) {
        __syncthreads();
        // 16 rows x kT columns per tile in groups of 4 floats, WT per thread; cells % 4 == 0 is
        // checked on the host, so a group is entirely inside or outside the map
#pragma unroll
        for (int e = 0; e < WT; ++e) {
            const int g = threadIdx.x + 256 * e;
            const int row = g / (kT / 4), col = (g % (kT / 4)) * 4;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (c0 + row < C) {
                if (p0 + col < cells) va = *reinterpret_cast<const float4 *>(a + (int64_t)(c0 + row) * cells + p0 + col);
                if (x0 + col < cells) vb = *reinterpret_cast<const float4 *>(bb + (int64_t)(c0 + row) * cells + x0 + col);
            }
            *reinterpret_cast<float4 *>(&As[row][col]) = va;
            *reinterpret_cast<float4 *>(&Bs[row][col]) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kCK; kk += 2) {
            float av[WT], bv[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                av[i] = As[kk + h][wr * 32 * WT + 32 * i + l31];
                bv[i] = Bs[kk + h][wc * 32 * WT + 32 * i + l31];
            }
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // this lane owns column x0 + wc*32*WT + j*32 + l31 and rows p0 + wr*32*WT + i*32 + (r&3) + 8*(r>>2) + 4*h
    float *o = out + (int64_t)b * cells * cells;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        const int col = x0 + wc * 32 * WT + j * 32 + l31;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = p0 + wr * 32 * WT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaxf(acc[i][j][r], 0.f);
                if (row < cells && col < cells) {
                    o[(int64_t)row * cells + col] = v;
                    sq = fmaf(v, v, sq);
                }
            }
        sq += __shfl_xor(sq, 32);                 // the two lane halves hold the other rows of the column
        if (h == 0 && col < cells)
            colsq[((int64_t)b * 2 * gridDim.y + 2 * blockIdx.y + wr) * cells + col] = sq;
    }
}

__global__ __launch_bounds__(256) void corr_acmm_scale_kernel(float *__restrict__ out, const float *__restrict__ colsq,
                                                              int nparts, const float *__restrict__ mask, int grid,
                                                              int H, int W) {
    const int b = blockIdx.z;
    const int cells = grid * grid;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cells) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += colsq[((int64_t)b * nparts + p) * cells + col];
    const float norm = sqrtf(s);
    const int y = col / grid, x = col % grid;
    const float *m = mask + (int64_t)b * H * W;
    float mv = -__builtin_inff();                 // max_pool2d(mask, 2, 2, ceil_mode=True)
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int yy = 2 * y + dy, xx = 2 * x + dx;
            if (yy < H && xx < W) mv = fmaxf(mv, m[yy * W + xx]);
        }
    // rows blockIdx.y, blockIdx.y + gridDim.y, ...: four loads in flight at a time
    float *o = out + (int64_t)b * cells * cells + col;
    const int64_t step = (int64_t)gridDim.y * cells;
    int p = blockIdx.y;
    for (; p + 3 * (int)gridDim.y < cells; p += 4 * gridDim.y) {
        float *o0 = o + (int64_t)p * cells;
        const float v0 = o0[0], v1 = o0[step], v2 = o0[2 * step], v3 = o0[3 * step];
        o0[0] = v0 / norm * mv;
        o0[step] = v1 / norm * mv;
        o0[2 * step] = v2 / norm * mv;
        o0[3 * step] = v3 / norm * mv;
    }
    for (; p < cells; p += gridDim.y) o[(int64_t)p * cells] = o[(int64_t)p * cells] / norm * mv;
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

extern "C" int ispa_corr_map_eccv_views(const float *f1, const float *f2, const int64_t *knn, int B, int V, int C,
                                        int H, int W, int K, int grid, float *out, float *workspace, void *stream) {
    ISPA_REQUIRE(B >= 0 && V >= 1 && K >= 0 && C > 0 && H > 0 && W > 0 && grid > 0, "bad shape");
    ISPA_REQUIRE(grid <= 4096, "grid %d too large", grid);
    ISPA_REQUIRE(C <= 8192 && (int64_t)B * V <= 65535, "C or B x V too large");
    if (B == 0 || K == 0) return ISPA_OK;
    ISPA_REQUIRE(f1 && f2 && knn && out, "null pointer");
    (void)workspace;                                  // the maps are resampled while they are read
    hipStream_t s = (hipStream_t)stream;
    const int items = B * V;
    const Resample rs = make_resample(H, W, grid, grid);
    const int cells = grid * grid;
    const size_t tile_lds = (size_t)C * kEccvMaxK * sizeof(float);
    const ispa::TimingSlot timed = ispa::timing_begin(s);
    if (K <= kEccvMaxK && tile_lds <= 48 * 1024) {
        hipLaunchKernelGGL(corr_eccv_tile_kernel, dim3((cells + kEccvCells - 1) / kEccvCells, items), dim3(256), tile_lds, s, f1, f2, knn,
                           V, C, grid, rs, K, out);
        hipLaunchKernelGGL(corr_eccv_scale_kernel, dim3(K, items), dim3(256), 0, s, out, cells, K);
    } else {
        hipLaunchKernelGGL(corr_eccv_kernel, dim3(K, items), dim3(256), (C + 4) * sizeof(float), s, f1, f2, knn, V, C,
                           grid, rs, K, out);
    }
    ispa::timing_end(timed, s);
    return ispa::check_launch("corr_eccv_kernel");
}

extern "C" int ispa_corr_map_eccv(const float *f1, const float *f2, const int64_t *knn, int B, int C, int H,
                                  int W, int K, int grid, float *out, float *workspace, void *stream) {
    return ispa_corr_map_eccv_views(f1, f2, knn, B, 1, C, H, W, K, grid, out, workspace, stream);
}

extern "C" size_t ispa_corr_map_acmm_workspace_floats(int B, int C, int grid) {
    if (B <= 0 || C <= 0 || grid <= 0) return 0;
    const size_t cells = (size_t)grid * grid;
    const size_t tile = cells <= 2048 ? 64 : 128;        // the GEMM's block tile, as chosen in ispa_corr_map_acmm
    return 2 * (size_t)B * C * cells + 2 * (size_t)B * ((cells + tile - 1) / tile) * cells;
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
    const Resample rs = make_resample(H, W, grid, grid);
    if (cells % 4 == 0) {
        // one launch brings both maps to the grid's size (skipped when they are there already), then the
        // MFMA product and the deterministic column normalisation; workspace = resampled maps + partials
        const bool small_tiles = cells <= 2048;             // 64 x 64 block tiles: less padding, more blocks
        const int tile = small_tiles ? 64 : 128;
        const int ptiles = (cells + tile - 1) / tile;
        const float *f1r = f1, *f2r = f2;
        float *colsq = workspace;
        if (H != grid || W != grid) {
            float *r1 = workspace, *r2 = workspace + (int64_t)B * C * cells;
            colsq = workspace + 2 * (int64_t)B * C * cells;
            hipLaunchKernelGGL(resample2_kernel, dim3(ispa::blocks_for((int64_t)B * C * cells, 256, 1ll << 30), 2),
                               dim3(256), 0, s, f1, f2, (int64_t)B * C, grid, rs, r1, r2);
            if (int rc = ispa::check_launch("resample2_kernel")) return rc;
            f1r = r1;
            f2r = r2;
        }
        const ispa::TimingSlot timed = ispa::timing_begin(s);
        if (small_tiles)
            hipLaunchKernelGGL(corr_acmm_gemm_kernel<1>, dim3(ptiles, ptiles, B), dim3(256), 0, s, f1r, f2r, C, cells,
                               out, colsq);
        else
            hipLaunchKernelGGL(corr_acmm_gemm_kernel<2>, dim3(ptiles, ptiles, B), dim3(256), 0, s, f1r, f2r, C, cells,
                               out, colsq);
        ispa::timing_end(timed, s);
        if (int rc = ispa::check_launch("corr_acmm_gemm_kernel")) return rc;
        hipLaunchKernelGGL(corr_acmm_scale_kernel, dim3((cells + 255) / 256, (cells + 27) / 28, B), dim3(256), 0, s, out,
                           colsq, 2 * ptiles, mask, grid, H, W);
        return ispa::check_launch("corr_acmm_scale_kernel");
    }
    hipLaunchKernelGGL(corr_acmm_kernel, dim3((cells + 63) / 64, B), dim3(256), 0, s, f1, f2, mask, C, grid, rs, out);
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
