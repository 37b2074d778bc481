/* Float32-ACCUMULATING matmul for the oracle's accumulation envelope (TEST INFRASTRUCTURE; see oracle/__init__.py and
 * oracle/numerics.py:set_accum).
 *
 * The oracle proper sums every dot product exactly (float64, one rounding: oracle/numerics.py:matmul_nt), which is what
 * "fp32 accumulate" means up to the summation order.  MLX -- like any GPU library -- accumulates in float32 in SOME
 * order (mx.matmul / mx.quantized_matmul at mlx_parallm/models/llama.py:64-67,93,143,160-165).  To bound what the
 * order alone can do at production widths this file computes the same products with float32 accumulators in two
 * different orders:
 *
 *   mode 0  "seq32"     y = (((p_0 + p_1) + p_2) + ...)           p_c = float32(exact sum of the 32 products of chunk c)
 *   mode 1  "pairwise"  y = balanced tree over the same p_c       (binary-counter stack: equal-sized blocks are merged)
 *
 * i.e. a 32-deep matrix-core step feeding a float32 accumulator, walked along K or reduced as a tree (split-K).
 * x (M, K) float32, w (N, K) float32 (values representable in the model dtype), y (M, N) float32.
 * Only tests/golden/make_golden_wide.py and the oracle's unit tests call this; the product never does.
 */
#include <stdlib.h>
#include <string.h>

#define NB 16          /* output columns per block */
#define CH 32          /* products per float32 rounding */
#define MAXLEV 40

typedef double v8d __attribute__((vector_size(64)));
typedef float v16f __attribute__((vector_size(64)));

static inline v16f to_f32(v8d lo, v8d hi) {
  v16f r;
  for (int i = 0; i < 8; ++i) { r[i] = (float)lo[i]; r[8 + i] = (float)hi[i]; }
  return r;
}

#define MR 4           /* rows of x per pass over a weight block (the block's loads are shared) */

void accum_gemm_nt(const float* x, const float* w, float* y, long M, long N, long K, int mode) {
  const long nblk = (N + NB - 1) / NB;
  const long nch = (K + CH - 1) / CH;
  const long Kp = (K + 7) / 8 * 8;
#pragma omp parallel
  {
    double* wt = (double*)aligned_alloc(64, (size_t)K * NB * sizeof(double));      /* [k][NB] */
    double* xr = (double*)aligned_alloc(64, (size_t)MR * Kp * sizeof(double));
#pragma omp for schedule(dynamic, 1)
    for (long nb = 0; nb < nblk; ++nb) {
      const long n0 = nb * NB;
      for (long k = 0; k < K; ++k)
        for (int j = 0; j < NB; ++j) wt[k * NB + j] = (n0 + j < N) ? (double)w[(n0 + j) * K + k] : 0.0;
      for (long m0 = 0; m0 < M; m0 += MR) {
        const int mr = (int)((M - m0 < MR) ? M - m0 : MR);
        for (int r = 0; r < MR; ++r) {
          const float* xm = x + (m0 + (r < mr ? r : 0)) * K;
          for (long k = 0; k < K; ++k) xr[r * Kp + k] = (double)xm[k];
        }
        v16f acc[MR];
        v16f stack[MR][MAXLEV];
        int depth = 0;
        for (int r = 0; r < MR; ++r) acc[r] = (v16f){0};
        for (long c = 0; c < nch; ++c) {
          const long k0 = c * CH, k1 = (k0 + CH < K) ? k0 + CH : K;
          v8d lo[MR], hi[MR];
          for (int r = 0; r < MR; ++r) { lo[r] = (v8d){0}; hi[r] = (v8d){0}; }
          for (long k = k0; k < k1; ++k) {
            const v8d* wv = (const v8d*)(wt + k * NB);
            const v8d w0 = wv[0], w1 = wv[1];
            for (int r = 0; r < MR; ++r) {          /* each row: its own sequential double chain, as in the 1-row form */
              lo[r] += xr[r * Kp + k] * w0;
              hi[r] += xr[r * Kp + k] * w1;
            }
          }
          if (mode == 0) {
            for (int r = 0; r < MR; ++r) acc[r] = acc[r] + to_f32(lo[r], hi[r]);
          } else {
            /* binary counter: after chunk c (0-based) merge once per trailing one bit of c */
            v16f p[MR];
            for (int r = 0; r < MR; ++r) p[r] = to_f32(lo[r], hi[r]);
            long t = c;
            while (t & 1) { --depth; for (int r = 0; r < MR; ++r) p[r] = stack[r][depth] + p[r]; t >>= 1; }
            for (int r = 0; r < MR; ++r) stack[r][depth] = p[r];
            ++depth;
          }
        }
        if (mode != 0) {
          for (int r = 0; r < MR; ++r) {
            acc[r] = stack[r][depth - 1];
            for (int d = depth - 2; d >= 0; --d) acc[r] = stack[r][d] + acc[r];      /* leftovers: smaller (later) blocks first */
          }
        }
        for (int r = 0; r < mr; ++r)
          for (int j = 0; j < NB && n0 + j < N; ++j) y[(m0 + r) * N + n0 + j] = acc[r][j];
      }
    }
    free(wt);
    free(xr);
  }
}
