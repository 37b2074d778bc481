/* Exactly-summed matmuls over COMPACT weights for the oracle at full depth (TEST INFRASTRUCTURE; see oracle/__init__.py).
 *
 * oracle/numerics.py:matmul_nt widens both operands to float64 and lets NumPy multiply: fine for the 2-block goldens, but a
 * 32- / 40-block model at production width (Mistral-7B: 7.2e9 weights, Qwen3-14B: 14.8e9) does not fit this container as
 * float64 (or even float32) arrays.  These loops compute the same thing -- every product exact in double, the sum kept in
 * double, ONE rounding to float32 -- straight from the checkpoint's own storage:
 *
 *   exact_gemm_nt_w16   nn.Linear on a 16-bit weight matrix (mlx_parallm/models/llama.py:64-67,93,143,160-165): w as bf16 /
 *                       f16 bit patterns
 *   exact_qgemm_nt      nn.QuantizedLinear (mlx_parallm/utils.py:679-690; mx.quantized_matmul): w_hat = float32(q * scale) +
 *                       bias in float32 with two roundings, exactly as oracle/ref_quant.py:dequantize forms it, never stored
 *   dequant_q_f32       the same w_hat written out as float32 (for the float32-ACCUMULATING envelope variants, which take a
 *                       dense matrix: accum_gemm.c)
 *
 * A double sum of K <= 17408 exact products is not associative either, but its error (~1e-16 relative) changes the float32
 * rounding of about one output in 1e7 -- the same caveat as NumPy's own BLAS order (tests/test_oracle_units.py compares
 * the two paths).  Only tests/golden/make_golden_wide.py and the oracle's unit tests reach this file; the product never does.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MMAX 16

typedef double v8d __attribute__((vector_size(64)));

/* sum_k a[k] * b[k] over [0, n) in double: eight interleaved partial sums (vector lanes), then their sum in lane order */
static inline double dot_d(const float* a, const double* b, long n) {
  v8d s = {0};
  long k = 0;
  for (; k + 8 <= n; k += 8) {
    v8d av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (double)a[k + j]; bv[j] = b[k + j]; }
    s += av * bv;
  }
  double t = 0.0;
  for (int j = 0; j < 8; ++j) t += s[j];
  for (; k < n; ++k) t += (double)a[k] * b[k];
  return t;
}

static inline float bf16_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float f16_f(uint16_t v) {      /* IEEE half -> float, subnormals included (this gcc has no _Float16) */
  const uint32_t sgn = (uint32_t)(v & 0x8000u) << 16, e = (v >> 10) & 31u, m = v & 1023u;
  uint32_t u;
  if (e == 31u) u = sgn | 0x7f800000u | (m << 13);
  else if (e != 0u) u = sgn | ((e + 112u) << 23) | (m << 13);
  else if (m == 0u) u = sgn;
  else { float f = (float)m * 5.9604644775390625e-08f; memcpy(&u, &f, 4); u |= sgn; }
  float f; memcpy(&f, &u, 4); return f;
}

/* y (M, N) = x (M, K) . w (N, K)^T ; kind 0: w holds bfloat16 bits, 1: float16 bits */
void exact_gemm_nt_w16(const float* x, const uint16_t* w, int kind, float* y, long M, long N, long K) {
  for (long m0 = 0; m0 < M; m0 += MMAX) {
    const int mr = (int)((M - m0 < MMAX) ? M - m0 : MMAX);
#pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
      const uint16_t* wn = w + (size_t)n * K;
      double acc[MMAX];
      for (int r = 0; r < mr; ++r) acc[r] = 0.0;
      for (long k0 = 0; k0 < K; k0 += 256) {
        const long k1 = (k0 + 256 < K) ? k0 + 256 : K;
        double wd[256];
        if (kind == 0) for (long k = k0; k < k1; ++k) wd[k - k0] = (double)bf16_f(wn[k]);
        else for (long k = k0; k < k1; ++k) wd[k - k0] = (double)f16_f(wn[k]);
        for (int r = 0; r < mr; ++r) {
          const float* xm = x + (size_t)(m0 + r) * K;
          acc[r] += dot_d(xm + k0, wd, k1 - k0);
        }
      }
      for (int r = 0; r < mr; ++r) y[(size_t)(m0 + r) * N + n] = (float)acc[r];
    }
  }
}

/* codes little-endian in uint32 words (element j of a word at bits [bits j, bits (j + 1))), scales / biases (N, K / group)
 * float32 arrays holding values of the model dtype */
static inline void dequant_row(const uint32_t* pk, const float* sc, const float* bi, int bits, int group, long K, float* out) {
  const int per = 32 / bits;
  const uint32_t mask = (1u << bits) - 1u;
  for (long k = 0; k < K; ++k) {
    const uint32_t q = (pk[k / per] >> (bits * (int)(k % per))) & mask;
    const volatile float p = (float)q * sc[k / group];     /* volatile: two roundings, never one fused multiply-add */
    out[k] = p + bi[k / group];
  }
}

void exact_qgemm_nt(const float* x, const uint32_t* packed, const float* scales, const float* biases, int bits, int group,
                    float* y, long M, long N, long K) {
  const long wpr = K * bits / 32, gpr = K / group;
  for (long m0 = 0; m0 < M; m0 += MMAX) {
    const int mr = (int)((M - m0 < MMAX) ? M - m0 : MMAX);
#pragma omp parallel
    {
      float* wf = (float*)malloc((size_t)K * sizeof(float));
      double* wd = (double*)malloc((size_t)K * sizeof(double));
#pragma omp for schedule(static)
      for (long n = 0; n < N; ++n) {
        dequant_row(packed + (size_t)n * wpr, scales + (size_t)n * gpr, biases + (size_t)n * gpr, bits, group, K, wf);
        for (long k = 0; k < K; ++k) wd[k] = (double)wf[k];
        for (int r = 0; r < mr; ++r) {
          const float* xm = x + (size_t)(m0 + r) * K;
          double acc = 0.0;
          for (long k0 = 0; k0 < K; k0 += 256) acc += dot_d(xm + k0, wd + k0, ((k0 + 256 < K) ? k0 + 256 : K) - k0);
          y[(size_t)(m0 + r) * N + n] = (float)acc;
        }
      }
      free(wf);
      free(wd);
    }
  }
}

void dequant_q_f32(const uint32_t* packed, const float* scales, const float* biases, int bits, int group, float* out, long N,
                   long K) {
  const long wpr = K * bits / 32, gpr = K / group;
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; ++n)
    dequant_row(packed + (size_t)n * wpr, scales + (size_t)n * gpr, biases + (size_t)n * gpr, bits, group, K, out + (size_t)n * K);
}

/* 16-bit patterns -> float32 (kind as above) */
void widen_w16_f32(const uint16_t* w, int kind, float* out, long n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) out[i] = kind == 0 ? bf16_f(w[i]) : f16_f(w[i]);
}
