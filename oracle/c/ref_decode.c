/*
 * ref_decode.c -- plain-C restatement of ONE decoder block's decode step (L = 1) and of the
 * lm_head, used (a) to cross-check oracle/ref_model.py and (b) as the timed CPU baseline of
 * bench.py ("cpu_baseline.kind" = "port").  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py);
 * nothing in the product links or calls this file.
 *
 * Follows the reference as text: TransformerBlock.__call__ (mlx_parallm/models/llama.py:181-191),
 * Attention.__call__ (llama.py:84-146; qwen3.py:55-115 for the q/k norms), MLP (llama.py:164-165),
 * BatchedKVCache.update_and_fetch (models/base.py:66-85), with the MLX op semantics of SURVEY.md
 * App. A: activations/weights/KV in bfloat16 (round-to-nearest-even after every op), fp32
 * accumulation, half-split RoPE, GQA by head broadcast, fp32 softmax.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC ref_decode.c -o ../_build/libref_decode.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint16_t bf16;

static inline float bf2f(bf16 v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16 f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16)((u >> 16) | 0x40);   /* NaN stays NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16)(u >> 16);
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

/* y[n] = bf16( sum_k W[n][k] * x[k] ) for M rows of x; W row-major bf16 */
static void linear_bf16(const bf16* W, int N, int K, const float* x, int M, float* y) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    const bf16* w = W + (size_t)n * K;
    float acc[16];
    for (int m = 0; m < M; ++m) acc[m] = 0.f;
    for (int k = 0; k < K; k += 8) {
      float wf[8];
      for (int j = 0; j < 8; ++j) wf[j] = bf2f(w[k + j]);
      for (int m = 0; m < M; ++m) {
        const float* xm = x + (size_t)m * K + k;
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += wf[j] * xm[j];
        acc[m] += s;
      }
    }
    for (int m = 0; m < M; ++m) y[(size_t)m * N + n] = rbf(acc[m]);
  }
}

static void rmsnorm(const float* x, const bf16* w, int n, float eps, float* out) {
  double ss = 0.0;
  for (int i = 0; i < n; ++i) ss += (double)x[i] * x[i];
  const float rs = (float)(1.0 / sqrt(ss / n + eps));
  for (int i = 0; i < n; ++i) out[i] = rbf(rbf(x[i] * rs) * bf2f(w[i]));
}

typedef struct {
  int H, Hq, Hkv, D, I;
  float eps, rope_theta;
  int qk_norm;
  const bf16 *wq, *wk, *wv, *wo, *wg, *wu, *wd;   /* [N][K] row-major */
  const bf16 *in_norm, *post_norm, *q_norm, *k_norm;
} ref_layer;

/* One decode step of one block for B rows.  h [B][H] (in/out, bf16-valued floats);
 * kc/vc [B][Hkv][cap][D] bf16; pos[b] = tokens already cached for row b. */
void ref_layer_step(const ref_layer* L, float* h, bf16* kc, bf16* vc, int cap, const int* pos, int B, float* scratch) {
  const int H = L->H, Hq = L->Hq, Hkv = L->Hkv, D = L->D, I = L->I, G = Hq / Hkv;
  float* xn = scratch;                    /* [B][H] */
  float* q = xn + (size_t)B * H;          /* [B][Hq*D] */
  float* k = q + (size_t)B * Hq * D;      /* [B][Hkv*D] */
  float* v = k + (size_t)B * Hkv * D;
  float* att = v + (size_t)B * Hkv * D;   /* [B][Hq*D] */
  float* r = att + (size_t)B * Hq * D;    /* [B][H] */
  float* g = r + (size_t)B * H;           /* [B][I] */
  float* u = g + (size_t)B * I;           /* [B][I] */
  for (int b = 0; b < B; ++b) rmsnorm(h + (size_t)b * H, L->in_norm, H, L->eps, xn + (size_t)b * H);
  linear_bf16(L->wq, Hq * D, H, xn, B, q);
  linear_bf16(L->wk, Hkv * D, H, xn, B, k);
  linear_bf16(L->wv, Hkv * D, H, xn, B, v);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int hh = 0; hh < Hq + Hkv; ++hh) {
      float* x = hh < Hq ? q + ((size_t)b * Hq + hh) * D : k + ((size_t)b * Hkv + (hh - Hq)) * D;
      if (L->qk_norm) {
        float tmp[512];
        rmsnorm(x, hh < Hq ? L->q_norm : L->k_norm, D, L->eps, tmp);
        memcpy(x, tmp, sizeof(float) * D);
      }
      for (int i = 0; i < D / 2; ++i) {
        const double ang = (double)pos[b] * pow((double)L->rope_theta, -2.0 * i / D);
        const float c = (float)cos(ang), s = (float)sin(ang);
        const float x1 = x[i], x2 = x[i + D / 2];
        x[i] = rbf(x1 * c - x2 * s);
        x[i + D / 2] = rbf(x1 * s + x2 * c);
      }
    }
  for (int b = 0; b < B; ++b)
    for (int kh = 0; kh < Hkv; ++kh)
      for (int d = 0; d < D; ++d) {
        const size_t o = (((size_t)b * Hkv + kh) * cap + pos[b]) * D + d;
        kc[o] = f2bf(k[((size_t)b * Hkv + kh) * D + d]);
        vc[o] = f2bf(v[((size_t)b * Hkv + kh) * D + d]);
      }
  const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int hq = 0; hq < Hq; ++hq) {
      const int kh = hq / G, n = pos[b] + 1;
      const float* qq = q + ((size_t)b * Hq + hq) * D;
      const bf16* kk = kc + ((size_t)b * Hkv + kh) * cap * D;
      const bf16* vv = vc + ((size_t)b * Hkv + kh) * cap * D;
      float m = -INFINITY, l = 0.f, o[512];
      for (int d = 0; d < D; ++d) o[d] = 0.f;
      for (int s = 0; s < n; ++s) {
        float dot = 0.f;
        for (int d = 0; d < D; ++d) dot += qq[d] * bf2f(kk[(size_t)s * D + d]);
        dot *= scale;
        const float mn = dot > m ? dot : m;
        const float corr = expf(m - mn), p = expf(dot - mn);
        l = l * corr + p;
        for (int d = 0; d < D; ++d) o[d] = o[d] * corr + p * bf2f(vv[(size_t)s * D + d]);
        m = mn;
      }
      for (int d = 0; d < D; ++d) att[((size_t)b * Hq + hq) * D + d] = rbf(o[d] / l);
    }
  linear_bf16(L->wo, H, Hq * D, att, B, r);
  for (size_t i = 0; i < (size_t)B * H; ++i) h[i] = rbf(h[i] + r[i]);
  for (int b = 0; b < B; ++b) rmsnorm(h + (size_t)b * H, L->post_norm, H, L->eps, xn + (size_t)b * H);
  linear_bf16(L->wg, I, H, xn, B, g);
  linear_bf16(L->wu, I, H, xn, B, u);
  for (size_t i = 0; i < (size_t)B * I; ++i) {
    const float sg = rbf(1.0f / (1.0f + expf(-g[i])));
    g[i] = rbf(rbf(g[i] * sg) * u[i]);
  }
  linear_bf16(L->wd, H, I, g, B, r);
  for (size_t i = 0; i < (size_t)B * H; ++i) h[i] = rbf(h[i] + r[i]);
}

size_t ref_layer_scratch_floats(const ref_layer* L, int B) {
  return (size_t)B * (2 * (size_t)L->H + 2 * (size_t)L->Hq * L->D + 2 * (size_t)L->Hkv * L->D + 2 * (size_t)L->I);
}

/* final norm + lm_head logits [B][V] (llama.py:231,249-252) */
void ref_head(const bf16* norm_w, const bf16* w_head, int V, int H, float eps, const float* h, int B, float* logits, float* scratch) {
  for (int b = 0; b < B; ++b) rmsnorm(h + (size_t)b * H, norm_w, H, eps, scratch + (size_t)b * H);
  linear_bf16(w_head, V, H, scratch, B, logits);
}

/* ---- timing helper for bench.py: synthetic weights for `nl` blocks of the given shape, KV prefilled
 * with `ctx` synthetic tokens; runs `steps` decode steps over those blocks (+ the lm_head when V > 0) and
 * returns the wall seconds of the timed steps.  Values are a xorshift stream -- only the timing matters. */
static uint64_t xs_state = 0x9E3779B97F4A7C15ull;
static inline float xs_uniform(void) {
  xs_state ^= xs_state << 13; xs_state ^= xs_state >> 7; xs_state ^= xs_state << 17;
  return (float)((xs_state >> 40) * (1.0 / 16777216.0)) - 0.5f;
}
static bf16* rand_bf16(size_t n, float scale) {
  bf16* p = (bf16*)malloc(n * sizeof(bf16));
  if (!p) return NULL;
  for (size_t i = 0; i < n; ++i) p[i] = f2bf(xs_uniform() * scale);
  return p;
}

double ref_bench_decode(int H, int Hq, int Hkv, int D, int I, int V, int nl, int B, int ctx, int steps, int* threads_out,
                        double* head_seconds_out) {
  ref_layer* Ls = (ref_layer*)calloc(nl, sizeof(ref_layer));
  bf16** kcs = (bf16**)calloc(nl, sizeof(bf16*));
  bf16** vcs = (bf16**)calloc(nl, sizeof(bf16*));
  const int cap = ctx + steps + 1;
  for (int i = 0; i < nl; ++i) {
    ref_layer* L = &Ls[i];
    L->H = H; L->Hq = Hq; L->Hkv = Hkv; L->D = D; L->I = I; L->eps = 1e-5f; L->rope_theta = 10000.f; L->qk_norm = 0;
    L->wq = rand_bf16((size_t)Hq * D * H, 0.07f); L->wk = rand_bf16((size_t)Hkv * D * H, 0.07f);
    L->wv = rand_bf16((size_t)Hkv * D * H, 0.07f); L->wo = rand_bf16((size_t)H * Hq * D, 0.07f);
    L->wg = rand_bf16((size_t)I * H, 0.07f); L->wu = rand_bf16((size_t)I * H, 0.07f); L->wd = rand_bf16((size_t)H * I, 0.07f);
    L->in_norm = rand_bf16(H, 0.f); L->post_norm = rand_bf16(H, 0.f);
    for (int j = 0; j < H; ++j) { ((bf16*)L->in_norm)[j] = f2bf(1.f); ((bf16*)L->post_norm)[j] = f2bf(1.f); }
    kcs[i] = rand_bf16((size_t)B * Hkv * cap * D, 1.f);
    vcs[i] = rand_bf16((size_t)B * Hkv * cap * D, 1.f);
  }
  bf16* head = V > 0 ? rand_bf16((size_t)V * H, 0.07f) : NULL;
  bf16* fn = rand_bf16(H, 0.f);
  for (int j = 0; j < H; ++j) fn[j] = f2bf(1.f);
  float* h = (float*)malloc((size_t)B * H * sizeof(float));
  float* scratch = (float*)malloc(ref_layer_scratch_floats(&Ls[0], B) * sizeof(float));
  float* logits = V > 0 ? (float*)malloc((size_t)B * V * sizeof(float)) : NULL;
  int* pos = (int*)malloc(B * sizeof(int));
  int nth = 1;
#ifdef _OPENMP
  nth = omp_get_max_threads();
#endif
  if (threads_out) *threads_out = nth;
  double t0 = 0.0, t1 = 0.0, th = 0.0;
  for (int s = -1; s < steps; ++s) {   /* one untimed warm-up step */
#ifdef _OPENMP
    if (s == 0) t0 = omp_get_wtime();
#endif
    for (int b = 0; b < B; ++b) pos[b] = ctx + (s < 0 ? 0 : s);
    for (size_t i = 0; i < (size_t)B * H; ++i) h[i] = rbf(xs_uniform());
    for (int i = 0; i < nl; ++i) ref_layer_step(&Ls[i], h, kcs[i], vcs[i], cap, pos, B, scratch);
    if (V > 0) {
      double a = 0.0;
#ifdef _OPENMP
      a = omp_get_wtime();
#endif
      ref_head(fn, head, V, H, 1e-5f, h, B, logits, scratch);
#ifdef _OPENMP
      if (s >= 0) th += omp_get_wtime() - a;
#endif
    }
  }
#ifdef _OPENMP
  t1 = omp_get_wtime();
#endif
  if (head_seconds_out) *head_seconds_out = th;
  for (int i = 0; i < nl; ++i) {
    free((void*)Ls[i].wq); free((void*)Ls[i].wk); free((void*)Ls[i].wv); free((void*)Ls[i].wo);
    free((void*)Ls[i].wg); free((void*)Ls[i].wu); free((void*)Ls[i].wd);
    free((void*)Ls[i].in_norm); free((void*)Ls[i].post_norm); free(kcs[i]); free(vcs[i]);
  }
  free(Ls); free(kcs); free(vcs); free(head); free(fn); free(h); free(scratch); free(logits); free(pos);
  return t1 - t0;
}
