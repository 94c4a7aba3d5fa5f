// Device functions shared by the projection kernels and the fused disparity kernel.
//
// Arithmetic follows the reference's numpy code operation by operation so that rounded pixel
// coordinates and visibility flags are bit-exact:
//   pt2render.py:6-13   P @ [p;1] as a k-sequential fma chain (what dgemm does), divide, rint
//   pt2render.py:18-38  crop: subtract, multiply, add pad, rint -- separate roundings (the
//                       library is compiled with -ffp-contract=off; every fused op is explicit)
//   pt2render.py:43-55  (A) and classifier_code/.../utils/pt2render.py:43-53 (B) visibility
#pragma once
#include "common.h"

namespace ispa_geom {

constexpr double kCropSide = 224.0;

struct CropXform {
    double top, bottom, left, right;  // crop box as doubles (exact: they are int32)
    double ratio, pad_x, pad_y;
};

__device__ __forceinline__ CropXform make_crop(const int32_t *c) {
    CropXform x;
    x.top = (double)c[0];
    x.bottom = (double)c[1];
    x.left = (double)c[2];
    x.right = (double)c[3];
    const int y_len = c[1] - c[0];
    const int x_len = c[3] - c[2];
    x.pad_x = 0.0;
    x.pad_y = 0.0;
    if (y_len >= x_len) {
        x.ratio = kCropSide / (double)y_len;
        x.pad_x = (kCropSide - (double)x_len * x.ratio) / 2.0;
    } else {
        x.ratio = kCropSide / (double)x_len;
        x.pad_y = (kCropSide - (double)y_len * x.ratio) / 2.0;
    }
    return x;
}

// Strict crop test on the un-cropped coordinates, then the in-place remap.
__device__ __forceinline__ bool crop_apply(const CropXform &c, int y_ge_x, int resize, double &u,
                                           double &v) {
    const bool inside = (v > c.top) && (v < c.bottom) && (u > c.left) && (u < c.right);
    u = u - c.left;
    v = v - c.top;
    if (resize) {
        u = u * c.ratio;
        v = v * c.ratio;
        if (y_ge_x)
            u = u + c.pad_x;
        else
            v = v + c.pad_y;
        u = rint(u);
        v = rint(v);
    }
    return inside;
}

// Variant A in two halves, so that a caller can issue the gathers of several (point, pose) pairs
// before it needs any of them: depth_gate_index returns the texel index to read (0 for a point
// outside the gate, whose texel is discarded), depth_decide_a the flag.
__device__ __forceinline__ int64_t depth_gate_index(double u, double v, int W, int ylim, int xlim, bool &gate) {
    const double yt = trunc(v), xt = trunc(u);
    gate = yt > 0.0 && yt < (double)ylim && xt > 0.0 && xt < (double)xlim;
    const int y = gate ? (int)yt : 0, x = gate ? (int)xt : 0;
    return (int64_t)y * W + x;
}
__device__ __forceinline__ bool depth_decide_a(bool gate, double z, float texel, double relax) {
    return gate && fabs(z - (double)texel) <= relax;
}

template <int VARIANT>
__device__ __forceinline__ bool depth_test(double u, double v, double z,
                                           const float *__restrict__ depth, int H, int W, int ylim,
                                           int xlim, double relax) {
    // int() truncation of the already rounded coordinate; NaN/inf fail every comparison.
    const double yt = trunc(v), xt = trunc(u);
    const bool gate = yt > 0.0 && yt < (double)ylim && xt > 0.0 && xt < (double)xlim;
    if (VARIANT == ISPA_VIS_A) {
        // branch-free: a lane outside the gate reads texel 0 and discards it, so that the gathers
        // of several poses can be in flight at once
        const int y = gate ? (int)yt : 0, x = gate ? (int)xt : 0;
        const double d = (double)depth[(int64_t)y * W + x];
        return gate && fabs(z - d) <= relax;
    } else {
        if (!gate) return false;
        const int y = (int)yt, x = (int)xt;
        const float r32 = (float)relax;
        float best = __builtin_inff();
        const int y1 = min(y + 2, H), x1 = min(x + 2, W);
        for (int yy = y - 1; yy < y1; ++yy)
            for (int xx = x - 1; xx < x1; ++xx) best = fminf(best, depth[(int64_t)yy * W + xx] + r32);
        return z <= (double)best;
    }
}

// double -> int32 with the hardware's own rule for every input: truncation, saturation at the int32 range, 0 for
// NaN.  A C++ cast is undefined for values outside the range (the optimiser may assume they never occur), and
// the callers convert lanes whose coordinates are garbage before they mask them out.
__device__ __forceinline__ int cvt_i32_sat(double x) {
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// rint(a / b) with the IEEE division's result, at a fraction of its cost: a reciprocal refined by
// two Newton steps gives the quotient to a few ulp, which rounds to the same integer as the
// correctly rounded quotient unless it lies within 1e-5 of a half-integer (or is huge / not
// finite); only those lanes take the exact division.  Bit-identical to rint(a / b) by construction.
__device__ __forceinline__ double refined_rcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ double rint_div(double a, double b) {
    const double y = a * refined_rcp(b);
    double n = rint(y);
    const bool safe = fabs(y) < 1.0e9 && fabs(y - n) < 0.5 - 1.0e-5;
    if (__builtin_expect(!safe, 0)) n = rint(a / b);
    return n;
}
// Two quotients over one denominator (the two pixel coordinates of a projected point): one reciprocal, one
// branch for the lanes that need an exact division.
__device__ __forceinline__ void rint_div2(double a0, double a1, double b, double &n0, double &n1) {
    const double r = refined_rcp(b);
    const double y0 = a0 * r, y1 = a1 * r;
    n0 = rint(y0);
    n1 = rint(y1);
    const bool safe = fabs(y0) < 1.0e9 && fabs(y0 - n0) < 0.5 - 1.0e-5 && fabs(y1) < 1.0e9 && fabs(y1 - n1) < 0.5 - 1.0e-5;
    if (__builtin_expect(!safe, 0)) {
        n0 = rint(a0 / b);
        n1 = rint(a1 / b);
    }
}

__device__ __forceinline__ void project_point(const double *__restrict__ P, double px, double py,
                                              double pz, double &u, double &v, double &z) {
    double a[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = P[4 * r + 0] * px;
        acc = fma(P[4 * r + 1], py, acc);
        acc = fma(P[4 * r + 2], pz, acc);
        acc = fma(P[4 * r + 3], 1.0, acc);
        a[r] = acc;
    }
    z = a[2];
    rint_div2(a[0], a[1], a[2], u, v);
}

}  // namespace ispa_geom
