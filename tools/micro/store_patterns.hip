// Store-only micro-benchmark: how fast does the chip take 16-byte non-temporal stores laid out
//   0  linearly (every workgroup a contiguous 64 KiB run, runs in launch order)
//   1  as 128 x 64-pixel tiles of u64 in a 960-wide image (64 pieces of 1 KiB, 7680 bytes apart)
//   2  linearly, plain (temporal) stores
//   3  tiles, plain stores
// with persistent workgroups of 512 threads, two per CU, taking items off an atomic counter.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_patterns.hip -o build_dev/store_patterns && build_dev/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
constexpr int W = 960, H = 540, TW = 128, TH = 64;
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(unsigned long long *out, int n_img, int *counter) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long long items = (long long)n_img * tiles_x * tiles_y;
    __shared__ int item_s;
    const u64x2 v = {~0ull, ~0ull};
    for (;;) {
        if (threadIdx.x == 0) item_s = atomicAdd(counter, 1);
        __syncthreads();
        const long long item = item_s;
        __syncthreads();
        if (item >= items) break;
        if (MODE == 0 || MODE == 2) {
            // same bytes per item as a full tile: 64 KiB contiguous
            u64x2 *dst = reinterpret_cast<u64x2 *>(out) + item * (TW * TH / 2);
            const long long limit = (long long)n_img * W * H / 2;
            for (int e = threadIdx.x; e < TW * TH / 2; e += 512) {
                if (item * (TW * TH / 2) + e < limit) {
                    if (MODE == 0) __builtin_nontemporal_store(v, dst + e);
                    else dst[e] = v;
                }
            }
        } else {
            const int p = item / (tiles_x * tiles_y), tl = item % (tiles_x * tiles_y);
            const int tx0 = (tl % tiles_x) * TW, ty0 = (tl / tiles_x) * TH;
            const int tw = min(TW, W - tx0), th = min(TH, H - ty0);
            const int pairs = tw / 2, rows_per = 512 / pairs;
            const int lx = (threadIdx.x % pairs) * 2, ly0 = threadIdx.x / pairs;
            for (int ly = ly0; ly < th; ly += rows_per) {
                u64x2 *dst = reinterpret_cast<u64x2 *>(out + ((long long)p * H + ty0 + ly) * W + tx0 + lx);
                if (MODE == 1) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
    }
}
// one workgroup per item (no counter, no barriers): 4 = linear, 5 = tiles; non-temporal
template <int MODE>
__global__ __launch_bounds__(512) void k1(unsigned long long *out, int n_img) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long long item = blockIdx.x;
    const u64x2 v = {~0ull, ~0ull};
    if (MODE == 4) {
        u64x2 *dst = reinterpret_cast<u64x2 *>(out) + item * (TW * TH / 2);
        const long long limit = (long long)n_img * W * H / 2;
        for (int e = threadIdx.x; e < TW * TH / 2; e += 512)
            if (item * (TW * TH / 2) + e < limit) __builtin_nontemporal_store(v, dst + e);
    } else {
        const int p = item / (tiles_x * tiles_y), tl = item % (tiles_x * tiles_y);
        const int tx0 = (tl % tiles_x) * TW, ty0 = (tl / tiles_x) * TH;
        const int tw = min(TW, W - tx0), th = min(TH, H - ty0);
        const int pairs = tw / 2, rows_per = 512 / pairs;
        const int lx = (threadIdx.x % pairs) * 2, ly0 = threadIdx.x / pairs;
        for (int ly = ly0; ly < th; ly += rows_per)
            __builtin_nontemporal_store(v, reinterpret_cast<u64x2 *>(out + ((long long)p * H + ty0 + ly) * W + tx0 + lx));
    }
}
// grid-stride linear fill, 256 threads, 4 stores per thread per step (what a library fill does); 6 = nt, 7 = plain
template <int MODE>
__global__ __launch_bounds__(256) void k2(unsigned long long *out, long long n16) {
    const u64x2 v = {~0ull, ~0ull};
    u64x2 *dst = reinterpret_cast<u64x2 *>(out);
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; e < n16; e += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e + u < n16) {
                if (MODE == 6) __builtin_nontemporal_store(v, dst + e + u);
                else dst[e + u] = v;
            }
    }
}
int main() {
    const int n_img = 3600;
    unsigned long long *out;
    int *counter;
    const size_t bytes = (size_t)n_img * W * H * 8;
    hipMalloc(&out, bytes);
    hipMalloc(&counter, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mode = 0; mode < 8; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemset(counter, 0, 4);
            hipEventRecord(a);
            if (mode == 0) k<0><<<512, 512>>>(out, n_img, counter);
            if (mode == 1) k<1><<<512, 512>>>(out, n_img, counter);
            if (mode == 2) k<2><<<512, 512>>>(out, n_img, counter);
            if (mode == 3) k<3><<<512, 512>>>(out, n_img, counter);
            const int items = n_img * 8 * 9;
            if (mode == 4) k1<4><<<(unsigned)((bytes + 65535) / 65536), 512>>>(out, n_img);
            if (mode == 5) k1<5><<<items, 512>>>(out, n_img);
            if (mode == 6) k2<6><<<256 * 16, 256>>>(out, (long long)(bytes / 16));
            if (mode == 7) k2<7><<<256 * 16, 256>>>(out, (long long)(bytes / 16));
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("mode %d: %.3f ms  %.2f TB/s\n", mode, best, bytes / best / 1e9);
    }
    hipMemset(out, 0, bytes);   // the driver's own fill for comparison
    hipEventRecord(a);
    hipMemsetAsync(out, 0xff, bytes, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemsetAsync: %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    return 0;
}
