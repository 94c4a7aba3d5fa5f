/* CPU oracle, C part -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * orc_project      : data_creator/render_wt_pt_proj/pt2render.py:6-13 with the
 *                    accumulation order of the dgemm that numpy's matmul runs
 *                    (k-sequential fused multiply-add chain starting from 0).
 * orc_nn_exact     : contract of the Caffe-UCN `KNN` layer as used at
 *                    ucn_code/nets/train.prototxt:2128-2141 (external code, parity
 *                    unpinned): squared-L2 nearest neighbours, exact in float64.
 * orc_round_bf16   : round-to-nearest-even float32 -> bfloat16 -> float32.
 *
 * Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* pts (N,3) row-major, P (12) row-major 3x4, out (3,N): rows round(u), round(v), z. */
void orc_project(const double *pts, int64_t n, const double *P, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        double acc[3];
        for (int r = 0; r < 3; ++r) {
            double a = 0.0;
            a = fma(P[4 * r + 0], x, a);
            a = fma(P[4 * r + 1], y, a);
            a = fma(P[4 * r + 2], z, a);
            a = fma(P[4 * r + 3], 1.0, a);
            acc[r] = a;
        }
        out[i] = rint(acc[0] / acc[2]);
        out[n + i] = rint(acc[1] / acc[2]);
        out[2 * n + i] = acc[2];
    }
}

/* Canonical squared distance:
 *   dot(i,j) = sum_c (double)q[c][i] * (double)r[c][j], accumulated c = 0..C-1 in double
 *              (each product of two floats is exact in double),
 *   qq(i), rr(j) accumulated the same way,
 *   d2(i,j) = (qq(i) + rr(j)) - 2*dot(i,j).
 * Neighbours are ordered by (d2, j) ascending; ties go to the lowest j.
 * q is (C, Nq) and r is (C, Nr), the per-image slice of an NCHW feature map.
 * idx_out (k, Nq) int32 gets j + index_base, d2_out (k, Nq) double.
 */
void orc_nn_exact(const float *q, const float *r, int C, int64_t nq, int64_t nr, int k,
                  int index_base, int32_t *idx_out, double *d2_out)
{
    double *rr = (double *)calloc((size_t)nr, sizeof(double));
    for (int c = 0; c < C; ++c)
        for (int64_t j = 0; j < nr; ++j) {
            const double v = r[(int64_t)c * nr + j];
            rr[j] += v * v;
        }
#pragma omp parallel
    {
        double *dot = (double *)malloc((size_t)nr * sizeof(double));
        double *bd = (double *)malloc((size_t)k * sizeof(double));
        int64_t *bj = (int64_t *)malloc((size_t)k * sizeof(int64_t));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nq; ++i) {
            double qq = 0.0;
            memset(dot, 0, (size_t)nr * sizeof(double));
            for (int c = 0; c < C; ++c) {
                const double qv = q[(int64_t)c * nq + i];
                const float *rrow = r + (int64_t)c * nr;
                qq += qv * qv;
                for (int64_t j = 0; j < nr; ++j)
                    dot[j] += qv * (double)rrow[j];
            }
            int have = 0;
            for (int64_t j = 0; j < nr; ++j) {
                const double d = (qq + rr[j]) - 2.0 * dot[j];
                if (have == k && !(d < bd[k - 1]))
                    continue; /* equal distance, higher index: not better */
                int pos = have < k ? have : k - 1;
                while (pos > 0 && d < bd[pos - 1]) {
                    bd[pos] = bd[pos - 1];
                    bj[pos] = bj[pos - 1];
                    --pos;
                }
                bd[pos] = d;
                bj[pos] = j;
                if (have < k)
                    ++have;
            }
            for (int t = 0; t < k; ++t) {
                idx_out[(int64_t)t * nq + i] = t < have ? (int32_t)(bj[t] + index_base) : -1;
                d2_out[(int64_t)t * nq + i] = t < have ? bd[t] : INFINITY;
            }
        }
        free(dot);
        free(bd);
        free(bj);
    }
    free(rr);
}

void orc_round_bf16(float *x, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, &x[i], 4);
        if ((u & 0x7fffffffu) <= 0x7f800000u) /* not NaN */
            u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        memcpy(&x[i], &u, 4);
    }
}

int orc_version(void) { return 1; }
