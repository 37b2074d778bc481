// ops_api.hip -- kernel-level C entry points (include/mi355_ops.h) on caller-owned device buffers.
#include <algorithm>

#include "../../include/mi355_ops.h"
#include "kernels.h"

namespace mi {
int launch_gemv(const LinearW& W, const GemvCall& c, hipStream_t st);
bool gemv_mfma_supported(const LinearW& W, const GemvCall& c);
}  // namespace mi

using namespace mi;

namespace {

int ready() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MI_ERR_RUNTIME, "no HIP device available (this library has no CPU backend)");
  return MI_OK;
}

int finish() {
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(nullptr));
  return MI_OK;
}

LinearW to_linear(const mi_op_linear* w) {
  LinearW W;
  W.wk = w->wk; W.w = w->w; W.scales = w->scales; W.biases = w->biases; W.N = w->N; W.K = w->K;
  W.group = w->group > 0 ? w->group : 64;
  W.layout = w->layout;
  return W;
}

GemvCall to_call(const mi_op_gemv_args* a) {
  GemvCall c;
  c.x = a->x; c.ldx = a->ldx; c.M = a->M; c.act = a->act; c.rnd = a->rnd; c.pro = a->pro; c.norm_w = a->norm_w;
  c.eps = a->eps; c.epi = a->epi; c.out = a->out; c.ldo = a->ldo; c.resid = a->resid; c.pair_offset = a->pair_offset;
  c.force_v1 = a->force_generic;
  return c;
}

AttnShape to_shape(const mi_op_attn_shape* s) {
  return AttnShape{s->B, s->L, s->Hq, s->Hkv, s->D, s->act, s->kv, s->rnd, s->cap};
}

}  // namespace

extern "C" {

int mi_op_gemv(const mi_op_linear* w, const mi_op_gemv_args* a) {
  if (!w || !a) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  const LinearW W = to_linear(w);
  const GemvCall c = to_call(a);
  const int maxm = gemv_mfma_supported(W, c) ? 16 : 8;
  if (c.M < 1 || c.M > maxm) return fail(MI_ERR_INVALID, "mi_op_gemv: M out of range for this kernel");
  MI_TRY(launch_gemv(W, c, nullptr));
  return finish();
}

int mi_op_gemv_bench(const mi_op_linear* w, const mi_op_gemv_args* a, int iters, float* avg_ms) {
  if (!w || !a || !avg_ms || iters < 1) return fail(MI_ERR_INVALID, "bad argument");
  MI_TRY(ready());
  const LinearW W = to_linear(w);
  const GemvCall c = to_call(a);
  hipEvent_t e0, e1;
  MI_HIP(hipEventCreate(&e0)); MI_HIP(hipEventCreate(&e1));
  MI_TRY(launch_gemv(W, c, nullptr));                   // warm-up
  MI_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) MI_TRY(launch_gemv(W, c, nullptr));
  MI_HIP(hipEventRecord(e1, nullptr));
  MI_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *avg_ms = ms / iters;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return finish();
}

// gemm_skinny.hip on its own: a->M in 9..128 (int4 / int8: 1..128), a->pro must be MI_PRO_NONE; ksplit 0 = the cost model's choice
// (*ksplit_used returns it); iters >= 1 additionally times `iters` back-to-back launches.  int4 weights above 16 rows with
// ksplit <= 0 run gemm_q4.hip (a->pro may then be MI_PRO_NORM); ksplit < 0 forces its plan: -(mt | TW << 3 | KW << 7 | ksplit << 11 | NS << 15).
int mi_op_gemm_skinny(const mi_op_linear* w, const mi_op_gemv_args* a, int ksplit, int* ksplit_used, int iters, float* avg_ms) {
  if (!w || !a) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  const LinearW W = to_linear(w);
  const GemvCall c = to_call(a);
  struct ForceGuard { bool on; ~ForceGuard() { if (on) gemm_q4_force(0); } } guard{ksplit < 0};
  if (ksplit < 0) {
    gemm_q4_force(-ksplit);
    if (!gemm_q4_supported(W, c, (size_t)c.M)) return fail(MI_ERR_UNSUPPORTED, "mi_op_gemm_skinny: ksplit < 0 forces gemm_q4, which does not take this call");
  }
  if (!gemm_skinny_supported(W, c, (size_t)c.M) && !gemm_q4_supported(W, c, (size_t)c.M))
    return fail(MI_ERR_UNSUPPORTED, "mi_op_gemm_skinny: call not supported by this kernel");
  const int groups = std::max(1, gemm_skinny_groups(W, c, (size_t)c.M));
  void* ws = nullptr; unsigned* ctr = nullptr;
  // 16 slices x N x 128 rows x 4 B: any ksplit of gemm_skinny; gemm_q4: its preparation buffers + partial tiles
  MI_HIP(hipMalloc(&ws, std::max((size_t)W.N * 8192 + 1024, gemm_skinny_ws_bytes(W, c, (size_t)c.M))));
  MI_HIP(hipMalloc(&ctr, (size_t)groups * sizeof(unsigned)));
  MI_HIP(hipMemset(ctr, 0, (size_t)groups * sizeof(unsigned)));
  if (ksplit_used) *ksplit_used = ksplit > 0 ? ksplit : gemm_skinny_ksplit(W, c, (size_t)c.M);
  if (ksplit < 0) ksplit = 0;
  int rc = launch_gemm_skinny(W, c, (size_t)c.M, nullptr, ws, ctr, ksplit);
  if (rc == MI_OK && iters >= 1 && avg_ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && rc == MI_OK; ++i) rc = launch_gemm_skinny(W, c, (size_t)c.M, nullptr, ws, ctr, ksplit);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *avg_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  const int rc2 = finish();
  hipFree(ws); hipFree(ctr);
  return rc != MI_OK ? rc : rc2;
}

int mi_op_gemm_prefill(const mi_op_linear* w, const mi_op_gemv_args* a, int iters, float* avg_ms) {
  if (!w || !a) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  const LinearW W = to_linear(w);
  const GemvCall c = to_call(a);
  if (c.M < 1 || c.pro != PRO_NONE || W.layout != 1 || wk_is_quant(W.wk) || c.act == MI_F32)
    return fail(MI_ERR_UNSUPPORTED, "mi_op_gemm_prefill: tile-major dense 16-bit weights, 16-bit activations, no prologue");
  // K-split workspace as the engine holds it (float32 partial tiles; prompts below 4096 rows)
  void* ws = nullptr; size_t cap = 0;
  if (c.M < 4096) { cap = (size_t)128 << 20; MI_HIP(hipMalloc(&ws, cap)); }
  int rc = launch_gemm_prefill(W, c, (size_t)c.M, nullptr, nullptr, ws, cap);
  if (rc == MI_OK && iters >= 1 && avg_ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && rc == MI_OK; ++i) rc = launch_gemm_prefill(W, c, (size_t)c.M, nullptr, nullptr, ws, cap);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *avg_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  const int rc2 = finish();
  if (ws) hipFree(ws);
  return rc != MI_OK ? rc : rc2;
}

int mi_op_gemv_uses_mfma(const mi_op_linear* w, const mi_op_gemv_args* a) {
  if (!w || !a) return 0;
  return gemv_mfma_supported(to_linear(w), to_call(a)) ? 1 : 0;
}

uint64_t mi_op_tiled_bytes(const mi_op_linear* w) {
  if (!w || !tiled_supported(w->wk, w->N, w->K, w->group > 0 ? w->group : 64)) return 0;
  return (uint64_t)tiled_bytes(w->wk, w->N, w->K);
}

int mi_op_repack_tiled(const mi_op_linear* w, void* dst) {
  if (!w || !dst) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  MI_TRY(launch_repack_tiled(to_linear(w), dst, nullptr));
  return finish();
}

int mi_op_embed(const mi_op_linear* w, const int32_t* tokens, int rows, int act, int rnd, void* out) {
  if (!w || !tokens || !out) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  EmbedCall ec{tokens, rows, act, rnd, out};
  MI_TRY(launch_embed(to_linear(w), ec, nullptr));
  return finish();
}

int mi_op_rope_tables(float* cos_tab, float* sin_tab, int max_pos, int head_dim, float base, float scale) {
  MI_TRY(ready());
  MI_TRY(launch_rope_tables(cos_tab, sin_tab, max_pos, head_dim, base, scale, nullptr));
  return finish();
}

int mi_op_rope_append(const mi_op_attn_shape* s, const void* qkv, void* q_out, void* kcache, void* vcache,
                      const int32_t* offsets, const void* q_norm_w, const void* k_norm_w, float eps,
                      const float* cos_tab, const float* sin_tab, int max_pos) {
  if (!s) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  RopeAppendCall rc{to_shape(s), qkv, q_out, kcache, vcache, offsets, q_norm_w, k_norm_w, eps, cos_tab, sin_tab, max_pos};
  MI_TRY(launch_rope_append(rc, nullptr));
  return finish();
}

int mi_op_attention(const mi_op_attn_shape* s, const void* q, const void* kcache, const void* vcache,
                    const int32_t* offsets, void* out, float scale, int nsplit, float* partial) {
  if (!s) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  AttnCall ac{to_shape(s), q, kcache, vcache, offsets, out, scale, nsplit, partial};
  MI_TRY(launch_attention(ac, nullptr));
  return finish();
}

int mi_op_attention_decode(const mi_op_attn_shape* s, const void* qkv, void* kcache, void* vcache,
                           const int32_t* offsets, const void* q_norm_w, const void* k_norm_w, float eps,
                           const float* cos_tab, const float* sin_tab, void* out, float scale, int rnd_out,
                           int nsplit, float* partial, int32_t* counters, int variant, int iters, float* avg_ms) {
  if (!s) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  AttnDecodeCall ac{to_shape(s), qkv, kcache, vcache, offsets, q_norm_w, k_norm_w, eps, cos_tab, sin_tab,
                    out, scale, rnd_out, nsplit, partial, (int*)counters, variant};
  if (iters <= 1) {
    MI_TRY(launch_attention_decode(ac, nullptr));
    return finish();
  }
  hipEvent_t e0, e1;
  MI_HIP(hipEventCreate(&e0)); MI_HIP(hipEventCreate(&e1));
  MI_TRY(launch_attention_decode(ac, nullptr));
  MI_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) MI_TRY(launch_attention_decode(ac, nullptr));
  MI_HIP(hipEventRecord(e1, nullptr));
  MI_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MI_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (avg_ms) *avg_ms = ms / iters;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return finish();
}

int mi_op_sample(float* logits, int B, int V, float temperature, float top_p, const float* uniforms,
                 int top_logprobs, int32_t* tokens_out, float* logprob_out, float* prob_row0_out,
                 int32_t* topk_ids, float* topk_logprobs, float* row_stats) {
  if (!logits || !tokens_out || !row_stats) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(ready());
  SampleCall sc{};
  sc.logits = logits; sc.B = B; sc.V = V; sc.rnd = RND_NONE; sc.temperature = temperature; sc.top_p = top_p;
  sc.uniforms = uniforms; sc.seed = 0; sc.step = 0; sc.top_logprobs = top_logprobs; sc.lp_temp = 0;
  sc.tokens_out = tokens_out; sc.logprob_out = logprob_out; sc.prob_row0_out = prob_row0_out;
  sc.topk_ids = topk_ids; sc.topk_logprobs = topk_logprobs; sc.row_stats = row_stats;
  MI_TRY(launch_sample(sc, nullptr));
  return finish();
}

}  // extern "C"
