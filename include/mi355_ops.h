/*
 * mi355_ops.h -- kernel-level entry points of libmi355_decode.so.
 *
 * These expose the individual HIP kernels behind mi355_decode.h on caller-owned DEVICE
 * buffers (e.g. torch tensors) so that each fused stage can be checked against the oracle at
 * its own scale and timed on its own (bench.py roofline leg).  They are not needed to use the
 * engine.  Every call runs on the HIP null stream of the current device and synchronises
 * before returning.  Error convention as in mi355_decode.h.
 *
 * Reference op each one replaces (MLX call sites in the reference):
 *   mi_op_gemv        nn.Linear / nn.QuantizedLinear (+ fused RMSNorm / residual / SwiGLU):
 *                     llama.py:64-67,93,143,160-165,175-177,188-190,250-252; qwen3.py:37-40,63,115
 *   mi_op_embed       nn.Embedding / QuantizedEmbedding: llama.py:212; qwen3.py:166
 *   mi_op_rope_append q_norm/k_norm + nn.RoPE + cache.update_and_fetch:
 *                     qwen3.py:65-70; llama.py:107-125; base.py:66-85,119-140
 *   mi_op_attention   mx.fast.scaled_dot_product_attention + causal mask: llama.py:139-141; base.py:17-40
 *   mi_op_sample      sample closure + top_p_sampling: utils.py:345-364; sample_utils.py:3-38
 */
#ifndef MI355_OPS_H
#define MI355_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* weight kinds */
#define MI_WK_F32 0
#define MI_WK_BF16 1
#define MI_WK_F16 2
#define MI_WK_Q4_F32 3
#define MI_WK_Q4_BF16 4
#define MI_WK_Q4_F16 5
#define MI_WK_Q8_F32 6
#define MI_WK_Q8_BF16 7
#define MI_WK_Q8_F16 8

/* run-time logical rounding on top of the storage dtype */
#define MI_RND_NONE 0
#define MI_RND_BF16 1
#define MI_RND_F16 2

#define MI_PRO_NONE 0
#define MI_PRO_NORM 1

#define MI_EPI_STORE 0
#define MI_EPI_STORE_F32 1
#define MI_EPI_RESID 2
#define MI_EPI_SWIGLU 3

typedef struct mi_op_linear {
  int32_t wk;            /* MI_WK_* */
  int32_t N, K, group;
  const void* w;         /* [N][K] dense, or MLX-packed [N][K*bits/32] */
  const void* scales;    /* [N][K/group] */
  const void* biases;
  int32_t layout;        /* 0 = row-major (checkpoint order), 1 = tile-major (mi_op_repack_tiled; then w is
                            the tiled buffer and scales/biases are unused) */
} mi_op_linear;

typedef struct mi_op_gemv_args {
  const void* x;         /* [M][ldx] */
  int32_t ldx, M;
  int32_t act, rnd;      /* MI_F32/BF16/F16 storage, MI_RND_* */
  int32_t pro;           /* MI_PRO_* */
  int32_t epi;           /* MI_EPI_* */
  const void* norm_w;    /* [K] */
  float eps;
  int32_t ldo;
  void* out;
  void* resid;
  int32_t pair_offset;
  int32_t force_generic; /* 1 = never take the MFMA path */
} mi_op_gemv_args;

typedef struct mi_op_attn_shape {
  int32_t B, L, Hq, Hkv, D;
  int32_t act, kv, rnd, cap;
} mi_op_attn_shape;

int mi_op_gemv(const mi_op_linear* w, const mi_op_gemv_args* a);
int mi_op_gemv_uses_mfma(const mi_op_linear* w, const mi_op_gemv_args* a);
/* launches the same call `iters` times back to back and returns the mean launch time (HIP events) */
int mi_op_gemv_bench(const mi_op_linear* w, const mi_op_gemv_args* a, int iters, float* avg_ms);
/* the split-K weight-streaming GEMM the engine uses for decode steps of 9..128 rows (int4 / int8 weights: 1..128), on its own:
 * a->M in that range, a->pro = MI_PRO_NONE, tile-major 16-bit, int4 or int8 (group 64) weights.  ksplit 0 = the library's
 * cost model (returned in *ksplit_used); iters >= 1 also times that many back-to-back launches into *avg_ms.
 * int4 weights above 16 rows with ksplit <= 0 run the round-4 kernel (gemm_q4.hip: x prepared once per launch, K split
 * over the waves of a workgroup; a->pro may then be MI_PRO_NORM); ksplit < 0 forces its plan for tests and A/B runs:
 * -(row_tiles | tile_units << 3 | k_lanes << 7 | ksplit << 11 | staging_waves << 15), any field 0 = the cost model's choice. */
int mi_op_gemm_skinny(const mi_op_linear* w, const mi_op_gemv_args* a, int ksplit, int* ksplit_used, int iters,
                      float* avg_ms);
/* gemm_prefill.hip on its own: the tile GEMM of the prefill call (generate_step's first model call, utils.py:243-262: every
 * nn.Linear over B x L rows at once).  a->M rows of 16-bit activations, tile-major dense 16-bit weights, a->pro =
 * MI_PRO_NONE; plain / residual / SwiGLU epilogues.  iters >= 1 also times that many back-to-back launches into *avg_ms. */
int mi_op_gemm_prefill(const mi_op_linear* w, const mi_op_gemv_args* a, int iters, float* avg_ms);
/* tile-major weight layout of the streaming kernels (what mi_engine_finalize applies to eligible
 * matrices): returns the size of the tiled buffer (0 if the matrix is not eligible) / fills `dst`. */
uint64_t mi_op_tiled_bytes(const mi_op_linear* row_major);
int mi_op_repack_tiled(const mi_op_linear* row_major, void* dst);
int mi_op_embed(const mi_op_linear* w, const int32_t* tokens, int rows, int act, int rnd, void* out);
int mi_op_rope_tables(float* cos_tab, float* sin_tab, int max_pos, int head_dim, float base, float scale);
int mi_op_rope_append(const mi_op_attn_shape* s, const void* qkv, void* q_out, void* kcache, void* vcache,
                      const int32_t* offsets, const void* q_norm_w, const void* k_norm_w, float eps,
                      const float* cos_tab, const float* sin_tab, int max_pos);
int mi_op_attention(const mi_op_attn_shape* s, const void* q, const void* kcache, const void* vcache,
                    const int32_t* offsets, void* out, float scale, int nsplit, float* partial);
/* fused decode attention (L == 1): q/k norm + RoPE + append + split-KV attention + combine.
 * counters: [B*Hkv] zero-initialised ints.  variant 0: MFMA kernel where it applies (16-bit caches,
 * head_dim % 32 == 0; float32 caches, head_dim 64 / 128), 1: VALU kernel (2: the MFMA kernel in its twelve-wave form,
 * only in a library built with -DMI_ATTN_WIDE; otherwise the same as 0).  iters > 1 repeats the launch (timing; avg_ms may be NULL). */
int mi_op_attention_decode(const mi_op_attn_shape* s, const void* qkv, void* kcache, void* vcache,
                           const int32_t* offsets, const void* q_norm_w, const void* k_norm_w, float eps,
                           const float* cos_tab, const float* sin_tab, void* out, float scale, int rnd_out,
                           int nsplit, float* partial, int32_t* counters, int variant, int iters, float* avg_ms);
int mi_op_sample(float* logits, int B, int V, float temperature, float top_p, const float* uniforms,
                 int top_logprobs, int32_t* tokens_out, float* logprob_out, float* prob_row0_out,
                 int32_t* topk_ids, float* topk_logprobs, float* row_stats);

#ifdef __cplusplus
}
#endif
#endif /* MI355_OPS_H */
