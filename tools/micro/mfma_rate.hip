// Bare bf16 MFMA loops on random operands: TFLOP/s of the whole chip and cycles per MFMA (s_memtime) for
//   shape 32x32x16 or 16x16x32, NACC independent accumulators per wave (the MFMAs of one accumulator form a
//   dependent chain), WPS waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o build_dev/mfma_rate && build_dev/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 16384;
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(const unsigned *seed, float *out, unsigned long long *cyc) {
    unsigned s = seed[threadIdx.x + blockIdx.x * 256];
    union { unsigned u[4]; bf16x8 v; } a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u;
            a[i].u[j] = (s & 0x807f807fu) | 0x3d003d00u;      // random mantissas / signs, magnitude ~ 0.03
            s = s * 1664525u + 1013904223u;
            b[i].u[j] = (s & 0x807f807fu) | 0x3d003d00u;
        }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int n = 0; n < NACC; ++n)
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < ITERS; it += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < NACC; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + n) & 3].v, b[n & 3].v, acc[n], 0, 0, 0);
        }
        for (int n = 0; n < NACC; ++n)
            for (int r = 0; r < 16; ++r) sum += acc[n][r];
    } else {
        f32x4 acc[NACC];
        for (int n = 0; n < NACC; ++n)
            for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < ITERS; it += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < NACC; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + n) & 3].v, b[n & 3].v, acc[n], 0, 0, 0);
        }
        for (int n = 0; n < NACC; ++n)
            for (int r = 0; r < 4; ++r) sum += acc[n][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 256] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int SHAPE, int NACC>
void run(int wps, const unsigned *seed, float *out, unsigned long long *cyc) {
    const int blocks = 256 * wps;       // 4 waves per block: wps blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<SHAPE, NACC><<<blocks, 256>>>(seed, out, cyc);
    hipEventRecord(e0);
    k<SHAPE, NACC><<<blocks, 256>>>(seed, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 2.0 * (SHAPE == 32 ? 32 * 32 * 16 : 16 * 16 * 32) * (double)ITERS * NACC * blocks * 4;
    printf("%dx%d  acc/wave %d  waves/SIMD %d: %7.1f TFLOP/s  %5.1f memtime ticks per MFMA per wave  %.3f ms\n", SHAPE, SHAPE,
           NACC, wps, flop / ms / 1e9, (double)c / (ITERS * NACC), ms);
}
int main() {
    unsigned *seed;
    float *out;
    unsigned long long *cyc;
    const int n = 256 * 8 * 256;
    hipMalloc(&seed, n * 4);
    hipMalloc(&out, n * 4);
    hipMalloc(&cyc, 8);
    unsigned *h = (unsigned *)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = 12345u + 2654435761u * i;
    hipMemcpy(seed, h, n * 4, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
        run<32, 1>(wps, seed, out, cyc);
        run<32, 2>(wps, seed, out, cyc);
        run<32, 4>(wps, seed, out, cyc);
        run<16, 1>(wps, seed, out, cyc);
        run<16, 4>(wps, seed, out, cyc);
        run<16, 8>(wps, seed, out, cyc);
    }
    return 0;
}
