/*
 * mi355_decode.h -- C ABI of libmi355_decode.so, the MI355X (gfx950) batched-decode engine.
 *
 * The reference (misanthropic-ai/mlx_parallm) has NO FFI / plugin interface: its boundary is
 * the Python API in mlx_parallm/utils.py and everything below it is the third-party MLX
 * runtime.  This ABI sits where MLX sits today; each entry point names the reference
 * interface it replaces (file:line in the reference checkout).  The Python side that binds
 * it is mlx_parallm_amd/_lib.py (ctypes); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - return 0 = OK, <0 = error class; message via mi_last_error() (thread local).
 *       MI_ERR_INVALID  (-1)  -> Python ValueError
 *       MI_ERR_NOTFOUND (-2)  -> Python FileNotFoundError / KeyError
 *       MI_ERR_UNSUPPORTED(-3)-> Python NotImplementedError
 *       MI_ERR_RUNTIME  (-4)  -> Python RuntimeError (HIP failure, out of memory, no device)
 *   - the caller owns every host buffer it passes, for the duration of the call only;
 *     the library owns all device memory behind its handles; every *_create has a *_destroy.
 *   - one mi_engine per device; calls on one engine are NOT re-entrant (the reference is
 *     single-flight too: utils.py:1345, server/main.py:1076-1107).  Different engines may be
 *     driven from different host threads / processes concurrently.
 *   - there is NO CPU backend: without a HIP device mi_engine_create fails with
 *     MI_ERR_RUNTIME.
 */
#ifndef MI355_DECODE_H
#define MI355_DECODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI guard.  mi_abi_version() returns MI_ABI_VERSION of the library that was loaded; both parameter structs begin with
 * `struct_size`, which the caller sets to sizeof() of ITS definition -- a binding generated from another version of this
 * header (a shorter mi_sample_params would make the library read pointers past its end) is refused with MI_ERR_INVALID
 * by mi_engine_create / mi_decode_sample / mi_step_enqueue(_rows) / mi_score_tokens instead of being read. */
#define MI_ABI_VERSION 2

#define MI_OK 0
#define MI_ERR_INVALID (-1)
#define MI_ERR_NOTFOUND (-2)
#define MI_ERR_UNSUPPORTED (-3)
#define MI_ERR_RUNTIME (-4)

/* element types of tensors handed to mi_engine_set_tensor / activation + KV storage */
#define MI_F32 0
#define MI_BF16 1
#define MI_F16 2
#define MI_U32 3 /* MLX-packed quantised weights: 32/bits codes per word, little-endian */

#define MI_ARCH_LLAMA 0 /* mlx_parallm/models/llama.py (also model_type "mistral": utils.py:33-36) */
#define MI_ARCH_QWEN3 1 /* mlx_parallm/models/qwen3.py: + per-head q_norm/k_norm (qwen3.py:65-70) */

/* KV-cache element type selector for mi_kv_create */
#define MI_KV_MODEL (-1) /* BatchedKVCache semantics: KV in the model dtype (models/base.py:66-85) */
/* MI_F32 on a 16-bit model = PagedKVCache semantics: float32 KV and everything after the
 * layer-0 attention promoted to float32 (models/base.py:104-117). */

typedef struct mi_engine mi_engine; /* opaque; replaces the nn.Module returned by utils.load (utils.py:711-747) */
typedef struct mi_kv mi_kv;         /* opaque; replaces List[PagedKVCache] from _KVPool.get (utils.py:199-223) */

/* ModelArgs (llama.py:15-46 / mlx-lm qwen3 ModelArgs) + config.json["quantization"] (utils.py:679-690) */
typedef struct mi_model_desc {
  uint32_t struct_size;      /* = sizeof(mi_model_desc) of the caller's header (ABI guard) */
  int32_t arch;              /* MI_ARCH_* */
  int32_t hidden_size;
  int32_t num_layers;
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;
  int32_t intermediate_size;
  int32_t vocab_size;
  float rms_norm_eps;
  float rope_theta;
  float rope_scale;          /* 1/factor for linear rope_scaling, else 1 (llama.py:69-76) */
  int32_t tie_word_embeddings;
  int32_t act_dtype;         /* MI_F32 | MI_BF16 | MI_F16: dtype of the weights (dense) or scales (quantised) */
  int32_t quant_bits;        /* 0 = no quantised tensors; else 4 or 8 */
  int32_t quant_group_size;  /* 64 (32 / 128 also accepted) */
  int32_t max_positions;     /* RoPE table length = largest KV capacity a mi_kv may have */
} mi_model_desc;

/* arguments of the `sample` closure (utils.py:345-364) + top_p_sampling (sample_utils.py:3-38) */
typedef struct mi_sample_params {
  uint32_t struct_size;      /* = sizeof(mi_sample_params) of the caller's header (ABI guard) */
  float temperature;         /* 0 -> greedy argmax (utils.py:352-353) */
  float top_p;               /* 0<top_p<1 -> nucleus (utils.py:355-356), else plain categorical */
  int32_t n_logit_bias;      /* logit_bias dict (utils.py:346-349) */
  const int32_t* logit_bias_ids;
  const float* logit_bias_values;
  const float* uniforms;     /* [B] caller-supplied U[0,1) noise, or NULL -> library Philox stream */
  uint64_t seed;             /* Philox key when uniforms == NULL */
  int32_t top_logprobs;      /* 0..MI_MAX_TOP_LOGPROBS: also return the k most likely tokens */
  int32_t logprobs_at_temperature; /* 0: logprob / top-k logprobs are log_softmax(logits) (generate_step);
                                    * 1 and temperature > 0: log_softmax(logits / temperature), the distribution
                                    * the server's logprobs path reports (server/main.py:571-584) */
  const float* row_temperature;    /* both NULL: `temperature` / `top_p` apply to every row.  Both [B]: per-row */
  const float* row_top_p;          /* values (continuous batching, where requests with different settings share a step) */
  int64_t stream_position;         /* Philox counter of this step when uniforms == NULL: >= 0 = the caller's own step index
                                    * (generate_step passes 0, 1, 2, ...: the same seed then reproduces the same tokens,
                                    * like re-seeding mx.random before a call); < 0 = the engine's running step counter */
} mi_sample_params;

#define MI_MAX_TOP_LOGPROBS 20

/* ---- engine life cycle: replaces utils.load_model (utils.py:630-708) ------------------- */
int mi_engine_create(const mi_model_desc* desc, int device, mi_engine** out);
void mi_engine_destroy(mi_engine* e);

/* One call per checkpoint tensor, named as in the safetensors file (e.g.
 * "model.layers.3.self_attn.q_proj.weight" / ".scales" / ".biases", "model.norm.weight",
 * "lm_head.weight"); replaces model.load_weights (utils.py:693-702).  `data` is a host
 * pointer (on_device = 0) or a device pointer on the engine's device (on_device = 1, e.g. a
 * torch tensor that was just filled by an RCCL broadcast); the bytes are COPIED.
 * Unknown names -> MI_ERR_NOTFOUND (the reference filters them, utils.py:693-698). */
int mi_engine_set_tensor(mi_engine* e, const char* name, const void* data, const int64_t* shape,
                         int ndim, int dtype, int on_device);

/* LoRALinear for one projection (mlx-lm load_adapters, utils.py:742-744; file layout
 * rl_training/lora_init.py:140-153).  proj: "self_attn.q_proj" etc.  A is (K, r), B is (r, N),
 * row-major, dtype MI_F32/BF16/F16.  y += (scale * ((x A) B)).astype(x.dtype). */
int mi_engine_set_lora(mi_engine* e, int layer, const char* proj, const void* A, const void* B,
                       int rank, float scale, int dtype, int on_device);

/* checks that every tensor was set, fuses q|k|v and gate|up buffers, builds RoPE tables */
int mi_engine_finalize(mi_engine* e);

/* ---- KV cache: replaces _KVPool.get / PagedKVCache (utils.py:199-223, base.py:93-150) -- */
int mi_kv_create(mi_engine* e, int batch, int capacity_tokens, int kv_dtype, mi_kv** out);
/* Block-paged form for serving (the reference's continuous scheduler rebuilds its caches per batch and keeps a
 * process-global prefix cache, server/main.py:1404-1726, utils.py:231-287): `slots` rows share an arena of `n_blocks`
 * blocks of `block_tokens` tokens (a power of two >= 16; block 0 is reserved); a row's tokens live in the blocks its
 * table names, so a row grows without copying, costs memory only for what it holds, and rows with a common prompt
 * prefix can share blocks (mi_kv_prefix_*).  A row holds at most max_tokens_per_row tokens.  Every entry point that
 * takes a mi_kv accepts either form; mi_kv_reserve is a no-op up to that limit.  MI_ERR_RUNTIME when a step needs a
 * block and every block is held by a live sequence. */
int mi_kv_create_paged(mi_engine* e, int slots, int block_tokens, int n_blocks, int max_tokens_per_row, int kv_dtype,
                       mi_kv** out);
/* Prefix reuse on a paged cache.  attach: `row` is empty; the longest run of FULL blocks at the start of tokens[0..n)
 * that an earlier prompt has published is mapped into the row (shared, read-only), the row's length becomes
 * *n_reused (a multiple of block_tokens, < n: at least one token is left to run) and the caller prefills
 * tokens[*n_reused..n).  Identical tokens at identical positions give identical K / V, so this is exact up to the
 * summation-order noise between prefill shapes; hits are verified token by token, never by hash alone.
 * publish: the full blocks of tokens[0..n) (already in the row's cache, i.e. after its prefill was enqueued) become
 * reusable; they stay alive after the row is reset, until evicted (least recently used first) when a step needs a
 * block.  Eviction takes the deepest block of a chain before its parents (a chain is only reachable from its root).
 * clear: forget everything (after a weight or adapter update).
 * stats: out[0..n) = free blocks, usable blocks, cached (published) blocks, reused tokens, looked-up tokens, evictions,
 * evictable blocks (published blocks no live row maps: only these can be given back to a step that needs a block). */
int mi_kv_prefix_attach(mi_kv* kv, int row, const int32_t* tokens, int n, int* n_reused);
int mi_kv_prefix_publish(mi_kv* kv, int row, const int32_t* tokens, int n);
int mi_kv_prefix_clear(mi_kv* kv);
int mi_kv_stats(const mi_kv* kv, int64_t* out, int n);
void mi_kv_destroy(mi_kv* kv);
int mi_kv_reset(mi_kv* kv, int batch);               /* base.py:146-149 */
int mi_kv_reserve(mi_kv* kv, int capacity_tokens);   /* base.py:104-117 growth, contents kept */
int mi_kv_offsets(const mi_kv* kv, int32_t* out);    /* base.py:142-144: per-row lengths [B] */
int mi_kv_capacity(const mi_kv* kv);

/* ---- forward: replaces `model(y, cache=cache)` (utils.py:403; llama.py:243-253) --------- */
/* tokens [B,L] row-major.  L>1 = prefill with the additive causal mask of base.py:17-40
 * (left pads ARE attended, quirk Q1); L==1 = decode.  logits_out: NULL, or [B,V] floats of
 * the last position (utils.py:404), or [B,L,V] if all_pos.  Synchronous. */
int mi_forward(mi_engine* e, mi_kv* kv, const int32_t* tokens, int B, int L, float* logits_out,
               int all_pos);

/* ---- one generate_step iteration: `_step` (utils.py:401-418), synchronous -------------- */
/* tokens_in [B,L].  Outputs (any may be NULL): tokens_out [B]; logprob_out [B] =
 * log_softmax(logits)[b, token_b]; prob_row0_out [B] = softmax(logits)[0, token_b] (the
 * `probs` the reference yields, quirk Q6); topk_ids/topk_logprobs [B,top_logprobs]. */
int mi_decode_sample(mi_engine* e, mi_kv* kv, const int32_t* tokens_in, int B, int L,
                     const mi_sample_params* sp, int32_t* tokens_out, float* logprob_out,
                     float* prob_row0_out, int32_t* topk_ids, float* topk_logprobs);

/* Teacher-forced scoring: the all-position logits + log-softmax gather of the server's echo /
 * perplexity paths (server/main.py:530-557, 646-654), through the KV cache instead of a re-run per
 * step.  Feeds tokens [B,L] (they are appended to `kv`), then for every position (b,i) reports
 * log softmax(logits[b,i] (+ logit_bias) (/ temperature if sp->logprobs_at_temperature))[targets[b,i]]
 * into logprob_out [B*L] (targets < 0: position skipped, 0.0 reported) and, if sp->top_logprobs = k > 0,
 * the k most likely ids / logprobs of that position into topk_ids / topk_logprobs [B*L,k].
 * Nothing is sampled; sp->top_p, uniforms and seed are ignored. */
int mi_score_tokens(mi_engine* e, mi_kv* kv, const int32_t* tokens, const int32_t* targets, int B, int L,
                    const mi_sample_params* sp, float* logprob_out, int32_t* topk_ids, float* topk_logprobs);

/* ---- pipelined form of the same step: the reference's one-step-ahead
 *      mx.async_eval(next_y); mx.eval(y) (utils.py:420-427) ----------------------------- */
/* Enqueue one step on the engine's stream and return a ticket.  tokens_in == NULL feeds the
 * tokens sampled by the previous enqueued step (they never leave the device). */
int mi_step_enqueue(mi_engine* e, mi_kv* kv, const int32_t* tokens_in, int B, int L,
                    const mi_sample_params* sp, int64_t* ticket);
/* Continuous batching (server/main.py:1404-1726 admits requests between steps): the same step on a SUBSET of
 * the cache's rows.  rows[n] are distinct row indices of `kv` (its batch = the number of slots); tokens_in is
 * [n, L] in that order, or NULL to feed the tokens the previous step sampled -- then the previous step must have
 * had the same n (and, for meaningful results, the same rows in the same order).  Rows not named are untouched,
 * so a finished sequence's slot can be reset (mi_kv_reset_row) and prefilled with a new prompt (n = 1, L = its
 * length) while the other slots keep decoding in their own steps.  Results come back in `rows` order. */
int mi_step_enqueue_rows(mi_engine* e, mi_kv* kv, const int32_t* rows, int n, const int32_t* tokens_in, int L,
                         const mi_sample_params* sp, int64_t* ticket);
/* Chunked prefill co-scheduled with the live decode rows (the reference admits a request only between batches and
 * prefills it alone, server/main.py:1404-1726): ONE pass over the weights for n segments that advance by different
 * numbers of tokens.  Segment i appends lens[i] >= 1 tokens to cache row rows[i] (distinct rows); the one-token
 * segments (decode rows) must come first; `tokens` is the concatenation (sum of lens).  want[i] != 0: sample a token
 * from the logits after the segment's last position (every decode row, and the LAST chunk of a prompt); want[i] == 0:
 * an inner chunk, no logits.  mi_step_wait then returns the results of the wanted segments, in segment order (the
 * ticket's batch = their count; row_temperature / row_top_p, if given, have one entry per wanted segment).  Explicit
 * tokens only: the device-resident token feed of mi_step_enqueue(_rows) restarts after a mixed step. */
int mi_step_enqueue_mixed(mi_engine* e, mi_kv* kv, const int32_t* rows, const int32_t* lens, const int32_t* want, int n,
                          const int32_t* tokens, const mi_sample_params* sp, int64_t* ticket);
/* Forget the contents of one row (its length becomes 0); ordered behind the steps already enqueued. */
int mi_kv_reset_row(mi_kv* kv, int row);
/* Block until `ticket` has finished; copy out its results (same meaning as mi_decode_sample). */
int mi_step_wait(mi_engine* e, int64_t ticket, int32_t* tokens_out, float* logprob_out,
                 float* prob_row0_out, int32_t* topk_ids, float* topk_logprobs);

/* ---- measurement hooks (bench.py) ------------------------------------------------------- */
/* Bracket every launch of the kernel family `name` ("gemv_gate_up", "gemv_qkv", "gemv_o",
 * "gemv_down", "gemv_head", "attn_decode", ...; NULL or "" = off) with HIP events on the
 * engine's own stream. */
int mi_profile_select(mi_engine* e, const char* name);
/* Drains the recorded events: number of launches and their summed duration in ms. */
int mi_profile_read(mi_engine* e, int64_t* n_launches, double* total_ms);
/* Engine switches (A/B measurement and tests); unknown key -> MI_ERR_NOTFOUND.
 *   "force_generic_gemv"      1: every linear layer on the generic VALU kernel (default 0)
 *   "fused_decode_attention"  0: decode attention as rope/append + attention + combine launches (default 1)
 *   "prefill_gemm"            0: prefill through the chunked <= 16-row kernels instead of the tile GEMM (default 1)
 *   "fused_gemv_pairs"        bit 0: o_proj -> gate|up, bit 1: down_proj -> next layer's q|k|v as ONE launch
 *                             each with an in-launch seam (default 0: measured no faster than two launches)
 *   "decode_attention_mfma"   0: the VALU form of the fused decode attention also for 16-bit caches (default 1)
 *   "skinny_gemm"             0: decode steps of 9..128 rows (int4 / int8 weights: of any size) through <= 16-row
 *                             launches of the M <= 16 kernels instead of the split-K streaming GEMM (default 1)
 *   "norm_handover"           0: decode steps of <= 16 rows through the split-K kernel run the RMSNorm as its own launch
 *                             instead of handing its row statistics from the residual epilogue of one launch to the
 *                             staging of the next (default 1)
 *   "defer_norm"              0: float32-activation (PagedKVCache mode) decode steps run the RMSNorm as its own launch
 *                             instead of applying its row scale in the epilogue of the linear behind it (default 1)
 *   "prefill_x_terms"         3: float32-activation (PagedKVCache mode) prefill multiplies an EXACT three-term 16-bit split of x
 *                             by dense bf16 weights (three walks of W); default 2: hi + lo, 16+ mantissa bits of x, two walks
 *                             (the CPU float32-accumulating variants do not move under it: DESIGN 8d); int4 always 3
 *   "short_prefill_skinny"    0: prefill calls of <= 128 rows in all through the tile GEMM like longer ones (default 1: the
 *                             weight-streaming split-K kernel of the decode steps, ~2x faster at that size)
 *   "tile_weights"            0: keep weights row-major (before mi_engine_finalize only; default 1)
 * Environment switches read once by the library (A/B runs only): MI_SKINNY_MIN_ROWS (hand-over row count for 16-bit
 * weights, default 9), MI_SKINNY_Q4_MIN_ROWS (set: int4 hands over like 16-bit), MI_SKINNY_NO_F32 (set: float32-KV mode
 * back on the generic VALU kernel), MI_SKINNY_NO_RAGGED_K, MI_GEMM_TILE128 (prefill: always the
 * 128 x 128 tile), MI_GEMM_B_DIRECT (prefill: W fragments straight from global memory), MI_Q4 (0: int4 decode steps above
 * 16 rows keep the split-K kernel also on the wide matrices), MI_Q4_NARROW (1: gemm_q4 also for the normed narrow matrix),
 * MI_ATTN_PREFILL_F32_EXACT (1: float32-KV prefill attention with exact float32 products instead of two-term bf16 operands;
 * read per call). */
int mi_engine_set_option(mi_engine* e, const char* key, int64_t value);
/* Blocks until the engine's stream is idle. */
int mi_engine_sync(mi_engine* e);

const char* mi_last_error(void);
const char* mi_version(void);
int mi_abi_version(void);                  /* MI_ABI_VERSION of the loaded library */

#ifdef __cplusplus
}
#endif
#endif /* MI355_DECODE_H */
