// kernels.h -- host-side launch interface of the HIP kernels (internal to the library).
#pragma once
#include "common.h"

namespace mi {

// One (possibly fused) weight matrix W[N][K] as the kernels see it.
struct LinearW {
  int wk = WK_F32;             // weight kind (common.h)
  const void* w = nullptr;     // dense [N][K] or MLX-packed [N][K*bits/32] uint32
  const void* scales = nullptr;  // [N][K/group] in the scale dtype
  const void* biases = nullptr;
  int N = 0, K = 0, group = 64;
  int layout = 0;              // 0 = row-major (checkpoint order); 1 = tile-major (repack.hip); then
                               // `w` is the single tiled buffer (int4: scales/biases live inside it)
  // LoRA (mlx-lm LoRALinear): rows [lora_row0, lora_row0+lora_n) of this (fused) matrix get
  // + scale * ((x A) B); A [K][r], B [r][lora_n] stored as float32.
  const float* lora_a[2] = {nullptr, nullptr};
  const float* lora_b[2] = {nullptr, nullptr};
  int lora_row0[2] = {0, 0}, lora_n[2] = {0, 0}, lora_rank[2] = {0, 0};
  float lora_scale[2] = {0.f, 0.f};
};

enum : int { PRO_NONE = 0, PRO_NORM = 1 };
enum : int {
  EPI_STORE = 0,      // out[m][n] = T(acc)
  EPI_STORE_F32 = 1,  // out_f32[m][n] = float(T(acc))            (logits for the sampler)
  EPI_RESID = 2,      // resid[m][n] = T(resid[m][n] + T(acc))     (h = x + r, llama.py:188,190)
  EPI_SWIGLU = 3,     // out[m][n] = T(T(silu(g)) * u), g/u = rows n, n+pair_offset (llama.py:165)
  EPI_SWIGLU_GU8 = 4, // the same on the row-interleaved copy of a gate|up matrix (tile t = gate rows 8t..8t+7, then up rows
                      // 8t..8t+7; launch_gate_up_interleave): a PLAIN one-stream GEMV over 2 x pair_offset rows whose
                      // epilogue pairs columns j and j + 8 of every tile -- gemv_mfma only (decode steps of <= 16 rows)
};

struct GemvCall {
  const void* x = nullptr;  // [M][ldx] activations (storage type `act`)
  int ldx = 0;
  int M = 0;
  int act = MI_F32;         // storage type of x / out / resid / norm_w
  int rnd = RND_NONE;       // logical rounding applied on top of the storage type
  int pro = PRO_NONE;
  const void* norm_w = nullptr;  // [K] RMSNorm weight (PRO_NORM)
  float eps = 0.f;
  int epi = EPI_STORE;
  void* out = nullptr;
  int ldo = 0;
  void* resid = nullptr;
  int pair_offset = 0;
  const float* lora_t = nullptr;  // [M][2][max_rank] = round(x A) for the (up to 2) adapted row ranges
  int lora_t_ld = 0;
  int force_v1 = 0;
  int kx = 0;                     // > 0: W is a [hi | lo] matrix of 2 kx columns (an f16 model in the float32-activation mode): the
                                  // activations have kx columns and are walked twice (k mod kx); 0: W.K columns
  // RMSNorm hand-over between two launches of gemm_skinny.hip's 16-row instantiation: a residual epilogue leaves
  // sum(h^2) per (tile group, row) in sq_out; the next linear (pro = PRO_NORM) reads them as sq_in[sq_parts][16]
  float* sq_out = nullptr;
  const float* sq_in = nullptr;
  int sq_parts = 0;
  void* ev_start = nullptr;       // measurement: hipEvent_t pair stamped with this kernel's own begin / end
  void* ev_stop = nullptr;        // (hipExtLaunchKernelGGL); MFMA path only
};

int launch_gemv(const LinearW& W, const GemvCall& c, hipStream_t st);

// Two dependent decode GEMVs as ONE launch (gemv_mfma.hip, gemv_pair_kernel): B reads what A wrote.
// `counter` is a device word that every launch advances by gemv_pair_grid(); `base` is its value before
// this launch (the host keeps the running total); `error` is set to 1 if a workgroup gave up waiting after `spin_limit`
// polls -- the engine reads it back with every step's results and fails the call (engine.hip: seam_record / seam_check_sync).
struct GemvSeam {
  unsigned* counter;
  unsigned base;
  int* error;
  unsigned spin_limit;
};
bool gemv_pair_supported(const LinearW& WA, const GemvCall& a, const LinearW& WB, const GemvCall& b);
int launch_gemv_pair(const LinearW& WA, const GemvCall& a, const LinearW& WB, const GemvCall& b, const GemvSeam& s,
                     hipStream_t st);
int gemv_pair_grid();

// prefill (many rows): MFMA tile GEMM + row-wise RMSNorm (gemm_prefill.hip)
bool gemm_prefill_supported(const LinearW& W, const GemvCall& c, size_t rows);
// splitk_ws / splitk_cap: float32 workspace for the K-split form of the 128 x 128 tile (few hundred rows against a narrow
// matrix); null = never split
int launch_gemm_prefill(const LinearW& W, const GemvCall& c, size_t rows, hipStream_t st, void* scratch = nullptr,
                        void* splitk_ws = nullptr, size_t splitk_cap = 0);
// float32 activations (PagedKVCache mode) in front of launch_gemm_prefill: rows of [hi | mid | lo] bf16, 3 K wide, with the
// RMSNorm (norm_w = float32 weights, or null) applied first; then call launch_gemm_prefill with c.x = out, c.ldx = 3 K
size_t split3_bytes(size_t rows, int K);
int launch_split3_rows(const void* x, int ldx, const void* norm_w, float eps, void* out, int rows, int K, hipStream_t st,
                       int terms = 3);
// int4 (tile-major) -> 16-bit tile-major [hi | lo], K' = 2K: the operand of the prefill GEMM for quantised weights
size_t dequant_hilo_bytes(int N, int K);
int launch_dequant_q4_hilo(const LinearW& src, void* dst, hipStream_t st);
// block_per_row: one workgroup per row, row in registers -- for the few rows of a decode step (default: one wave
// per row, the prefill's kernel; the two add the squares in different orders)
// act = MI_F32: float32 rows and weights with the run-time logical rounding `rnd` (PagedKVCache mode)
int launch_rmsnorm_rows(const void* x, int ldx, const void* w, void* out, int ldo, int rows, int H, float eps, int act,
                        hipStream_t st, bool block_per_row = false, int rnd = 0);

// 9..64 rows, int8 weights 1..64 (decode steps of larger batches): weight-streaming split-K GEMM (gemm_skinny.hip).
// c.pro must be PRO_NONE; `ws` >= gemm_skinny_ws_bytes(), `ctr` >= gemm_skinny_groups() words that are zero between
// launches (the kernel leaves them zero); ksplit = 0 lets the cost model choose.
bool gemm_skinny_supported(const LinearW& W, const GemvCall& c, size_t rows);
size_t gemm_skinny_ws_bytes(const LinearW& W, const GemvCall& c, size_t rows);
int gemm_skinny_groups(const LinearW& W, const GemvCall& c, size_t rows);
int gemm_skinny_ksplit(const LinearW& W, const GemvCall& c, size_t rows);
int gemm_skinny_handover_ld(const LinearW& W, const GemvCall& c, size_t rows);     // > 0: the call can leave / take the RMSNorm hand-over
int gemm_skinny_tile_groups(const LinearW& W, const GemvCall& c, size_t rows);
int launch_gemm_skinny(const LinearW& W, const GemvCall& c, size_t rows, hipStream_t st, void* ws, unsigned* ctr,
                       int ksplit = 0);

// int4 (group 64) weights, 17..128 rows of 16-bit activations (gemm_q4.hip): x prepared once per launch (fragment-major,
// with the RMSNorm of c.pro = PRO_NORM applied there), K split over the waves of a workgroup.  launch_gemm_skinny routes
// to it; `ws` >= gemm_q4_ws_bytes() (preparation buffers + partial tiles), `ctr` >= gemm_q4_groups() zeroed words.
bool gemm_q4_supported(const LinearW& W, const GemvCall& c, size_t rows);
size_t gemm_q4_ws_bytes(const LinearW& W, const GemvCall& c, size_t rows);
int gemm_q4_groups(const LinearW& W, const GemvCall& c, size_t rows);
int gemm_q4_ksplit(const LinearW& W, const GemvCall& c, size_t rows);
void gemm_q4_force(int code);     // test / A-B hook: mt | TW << 3 | KW << 7 | ksplit << 11 | NS << 15 for this thread's next plans; 0 clears
int launch_gemm_q4(const LinearW& W, const GemvCall& c, size_t rows, hipStream_t st, void* ws, unsigned* ctr);

// tile-major weight layout (repack.hip)
bool tiled_supported(int wk, int N, int K, int group);
size_t tiled_bytes(int wk, int N, int K);
int launch_repack_tiled(const LinearW& src_row_major, void* dst, hipStream_t st);
// tile-major f16 (N x K) -> tile-major bf16 [hi | lo] (N x 2K), hi + lo == w exactly (repack.hip); dst: 2 x the source's bytes
int launch_f16_to_hilo(const LinearW& tiled_f16, void* dst, hipStream_t st);
// tile-major gate|up (rows [0, I) gate, [I, 2 I) up) -> tile-major with tile t = {gate rows 8t..8t+7, up rows 8t..8t+7}
int launch_gate_up_interleave(const LinearW& tiled, int pair_offset, void* dst, hipStream_t st);

// byte offset of the 16-byte piece holding W[row][k .. k+8) (k % 8 == 0) of a tile-major dense matrix
__host__ __device__ inline size_t tiled_piece_dense16(size_t row, int k, int K) {
  return (((row >> 4) * (size_t)(K / 32) + (size_t)(k >> 5)) * 64 + (size_t)(((k >> 3) & 3) * 16 + (int)(row & 15))) * 16;
}
// tile-major int4: byte offset of the block of (row, k), of the dword holding codes k..k+8, and of the scale / bias
__host__ __device__ inline size_t tiled_block_q4(size_t row, int k, int K) {
  return ((row >> 4) * (size_t)(K / 128) + (size_t)(k >> 7)) * 1152;
}
__host__ __device__ inline int tiled_q4_code_off(size_t row, int k) {   // within the block, bytes
  const int dd = (k >> 3) & 15;
  return (((dd >> 2) * 16 + (int)(row & 15)) * 4 + (dd & 3)) * 4;
}
__host__ __device__ inline int tiled_q4_scale_off(size_t row, int k) {  // + 64 for the bias
  return 1024 + (int)(row & 15) * 4 + ((k >> 6) & 1) * 2;
}
// tile-major int8 (group 64): block (row tile, k / 64) of 1088 bytes = [1024 codes: lane l = g*16 + r owns the 16
// codes W[16i + r][64j + 16g .. +16)] [32 B scales: row r -> one 16-bit value] [32 B biases]
__host__ __device__ inline size_t tiled_block_q8(size_t row, int k, int K) {
  return ((row >> 4) * (size_t)(K / 64) + (size_t)(k >> 6)) * 1088;
}
__host__ __device__ inline int tiled_q8_code_off(size_t row, int k) {   // within the block, bytes
  return (((k >> 4) & 3) * 16 + (int)(row & 15)) * 16 + (k & 15);
}
__host__ __device__ inline int tiled_q8_scale_off(size_t row) { return 1024 + (int)(row & 15) * 2; }   // + 32 for the bias
// t[m][slot][j] = round(sum_k x[m][k] A[k][j]) for the adapted ranges of W (same prologue as the gemv)
int launch_lora_down(const LinearW& W, const GemvCall& c, float* t, int t_ld, hipStream_t st);
// the LoRA term of the GEMV epilogues applied to an already stored c.out (EPI_STORE, c.M rows): after a tile GEMM
int launch_lora_up_add(const LinearW& W, const GemvCall& c, const float* t, int t_ld, hipStream_t st);

struct EmbedCall {
  const int32_t* tokens;  // device [rows]
  int rows;
  int act, rnd;
  void* out;  // [rows][H]
};
int launch_embed(const LinearW& W, const EmbedCall& c, hipStream_t st);

struct AttnShape {
  int B, L, Hq, Hkv, D;
  int act;      // storage type of qkv / q / out
  int kv;       // storage type of the caches
  int rnd;      // logical rounding of q, k (after norm / rope) and of the attention output
  int cap;      // cache capacity (tokens)
  const int32_t* rows;  // device [B] or null: batch entry b lives in cache row rows[b] (continuous batching:
                        // the call covers a subset of the cache's rows); offsets[] is indexed by cache row
  // Block-paged cache (mi_kv_create_paged): the cache arrays are arenas [block][Hkv][1 << bs_shift][D] and token `key` of
  // cache row r lives in block btab[r * bt_stride + (key >> bs_shift)].  btab == null: contiguous [row][Hkv][cap][D].
  const int32_t* btab;
  int bt_stride;
  int bs_shift;
};

// element index of (cache row kb, kv head kh, token key, d = 0) in a K or V array
template <bool PAGED>
__device__ __forceinline__ size_t kv_elem(const AttnShape& s, int kb, int kh, int key) {
  if constexpr (PAGED) {
    const int blk = s.btab[(size_t)kb * s.bt_stride + (key >> s.bs_shift)];
    return ((((size_t)blk * s.Hkv + kh) << s.bs_shift) + (size_t)(key & ((1 << s.bs_shift) - 1))) * s.D;
  } else {
    return (((size_t)kb * s.Hkv + kh) * s.cap + key) * s.D;
  }
}
struct RopeAppendCall {
  AttnShape s;
  const void* qkv;       // [B*L][(Hq+2Hkv)*D]
  void* q_out;           // [B*L][Hq*D]
  void* kcache;          // [B][Hkv][cap][D]
  void* vcache;
  const int32_t* offsets;  // device [B]: tokens already in the cache
  const void* q_norm_w;  // [D] or null (qwen3.py:65-70)
  const void* k_norm_w;
  float eps;
  const float* cos_tab;  // [max_pos][D/2]
  const float* sin_tab;
  int max_pos;
};
int launch_rope_append(const RopeAppendCall& c, hipStream_t st);

struct AttnCall {
  AttnShape s;
  const void* q;         // [B*L][Hq*D] (normed + roped)
  const void* kcache;
  const void* vcache;
  const int32_t* offsets;  // device [B]: tokens in the cache BEFORE this call's L were appended
  void* out;             // [B*L][Hq*D]
  float scale;
  int nsplit;            // >1 only for L == 1
  float* partial;        // [B*L*Hq][nsplit][D+2] when nsplit > 1
  int variant;           // 0: L > 1 goes to the MFMA prefill kernel where it applies; 1: always the VALU kernel
};
int launch_attention(const AttnCall& c, hipStream_t st);
// causal prefill attention on the matrix cores (attn_prefill.hip); launch_attention dispatches to it
bool attention_prefill_supported(const AttnShape& s);
int launch_attention_prefill(const AttnCall& c, hipStream_t st);

// fused decode step (L == 1): q/k norm + RoPE + KV append + split-KV attention + combine
struct AttnDecodeCall {
  AttnShape s;             // s.rnd = logical rounding of q / k after norm and RoPE
  const void* qkv;         // [B][(Hq+2Hkv)*D] raw projections
  void* kcache;
  void* vcache;
  const int32_t* offsets;  // device [B]: tokens already in the cache
  const void* q_norm_w;
  const void* k_norm_w;
  float eps;
  const float* cos_tab;
  const float* sin_tab;
  void* out;               // [B][Hq*D]
  float scale;
  int rnd_out;             // logical rounding of the attention output
  int nsplit;
  float* partial;          // [B*Hq][nsplit][D+2]
  int* counters;           // [B*Hkv] zero-initialised arrival tickets
  int variant;             // 0: MFMA kernel where it applies (16-bit caches, D % 32 == 0); 1: VALU kernel
  int n_host_off;          // > 0: host_off[b] = offsets[cache row of b], host_row[b] = that cache row (B <= 32)
  int host_off[32];
  int host_row[32];
};
int launch_attention_decode(const AttnDecodeCall& c, hipStream_t st);
bool attention_decode_supported(const AttnShape& s);
bool gemv_mfma_supported(const LinearW& W, const GemvCall& c);

struct SampleCall {
  float* logits;         // [B][V] float32 (modified in place by logit_bias)
  int B, V;
  int rnd;
  float temperature, top_p;
  int n_bias;
  const int32_t* bias_ids;   // device
  const float* bias_vals;    // device
  const float* uniforms;     // device [B] or null
  uint64_t seed, step;       // Philox key/counter when uniforms == null
  int top_logprobs;
  int lp_temp;               // report logprobs under softmax(logits / temperature) (temperature > 0)
  const int32_t* forced;     // device [B] or null: teacher forcing -- score these ids instead of sampling (< 0: skip row)
  const float* row_temp;     // device [B] or null: per-row temperature / top_p instead of the scalars
  const float* row_top_p;
  int32_t* tokens_out;       // device [B]
  float* logprob_out;        // device [B]
  float* prob_row0_out;      // device [B]
  int32_t* topk_ids;         // device [B][top_logprobs]
  float* topk_logprobs;
  float* row_stats;          // device [B][2] scratch: (max, logsumexp)
};
int launch_sample(const SampleCall& c, hipStream_t st);

int launch_advance_offsets(int32_t* offsets, const int32_t* rows, int B, int L, hipStream_t st);
// dst[i][:] = src[idx[i]][:] for n rows of row_bytes (a multiple of 16) each; idx on the device
int launch_gather_rows(const void* src, size_t row_bytes, const int32_t* idx, int n, void* dst, hipStream_t st);
int launch_rope_tables(float* cos_tab, float* sin_tab, int max_pos, int D, float base, float scale,
                       hipStream_t st);
int launch_convert(const void* src, int src_dt, void* dst, int dst_dt, size_t n, hipStream_t st);

}  // namespace mi
