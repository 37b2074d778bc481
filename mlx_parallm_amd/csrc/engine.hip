// engine.hip -- host runtime behind the C ABI of include/mi355_decode.h.
//
// One mi_engine = one model replica on one MI355X: device-resident weights (q|k|v and gate|up
// fused into single matrices), RoPE tables, an activation workspace, a private HIP stream and
// a small ring of pinned result slots for the one-step-ahead pipelining of generate_step
// (utils.py:420-427).  One mi_kv = the per-layer KV buffers of one batch, laid out
// [layer][B][Hkv][capacity][D], plus per-row lengths that live ON THE DEVICE so that a decode
// step needs no host data at all.
#include <algorithm>
#include <cstring>
#include <list>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

namespace mi {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(int code, const std::string& m) { g_err = m; return code; }

int launch_gemv_v1(const LinearW& W, const GemvCall& c, hipStream_t st);
int launch_gemv_mfma(const LinearW& W, const GemvCall& c, hipStream_t st);
bool gemv_mfma_supported(const LinearW& W, const GemvCall& c);

int launch_gemv(const LinearW& W, const GemvCall& c, hipStream_t st) {
  if (gemv_mfma_supported(W, c)) return launch_gemv_mfma(W, c, st);
  return launch_gemv_v1(W, c, st);
}

}  // namespace mi

using namespace mi;

namespace {

struct FusedLinear {
  LinearW W;
  int parts = 0;                 // bit mask of sub-tensors seen (weight / scales / biases per part)
  int n_parts = 1;
  int part_rows[3] = {0, 0, 0};
  bool quant = false;
  int dense_dtype = -1, scale_dtype = -1;
  size_t w_bytes = 0, s_bytes = 0;
  void* w = nullptr; void* scales = nullptr; void* biases = nullptr;
  // Persistent second copies of a matrix (made by mi_engine_finalize, i.e. BEFORE a scheduler sizes its KV arena from the
  // free device memory; a pointer is published only after its repack launch succeeded; when the allocation fails the
  // copy is marked unavailable and the linear keeps its plain path -- ensure_hilo / ensure_gu8)
  void* w_hilo = nullptr;        // f16 dense weights in the float32-activation (PagedKVCache) mode: [hi | lo] bf16 copy
  void* w_gu8 = nullptr;         // dense 16-bit gate|up: row-interleaved copy for the decode GEMV (EPI_SWIGLU_GU8)
  bool hilo_failed = false, gu8_failed = false;
  float* lora_a[2] = {nullptr, nullptr};
  float* lora_b[2] = {nullptr, nullptr};
  int seen_w[3] = {0, 0, 0}, seen_s[3] = {0, 0, 0}, seen_b[3] = {0, 0, 0};
};

struct LayerW {
  FusedLinear qkv, o, gate_up, down;
  void* in_norm = nullptr; void* post_norm = nullptr; void* q_norm = nullptr; void* k_norm = nullptr;
  // float32 copies for the float32-activation ("PagedKVCache quirk") mode
  void* in_norm32 = nullptr; void* post_norm32 = nullptr; void* q_norm32 = nullptr; void* k_norm32 = nullptr;
};

constexpr int NSLOT = 4;
struct Slot {
  hipEvent_t ev = nullptr;
  int32_t* tokens = nullptr; float* logprob = nullptr; float* prob0 = nullptr;
  int32_t* topk_ids = nullptr; float* topk_lp = nullptr;   // pinned host
  int B = 0, topk = 0; int64_t ticket = -1;
};

}  // namespace

struct mi_engine {
  mi_model_desc d{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<LayerW> layers;
  FusedLinear embed, lm_head;
  void* final_norm = nullptr; void* final_norm32 = nullptr;
  bool finalized = false;
  float* cos_tab = nullptr; float* sin_tab = nullptr;
  // workspace
  size_t ws_rows = 0; size_t ws_logit_rows = 0;
  void* h = nullptr; void* qkv = nullptr; void* q = nullptr; void* attn = nullptr; void* act = nullptr;
  float* logits = nullptr; float* lora_t = nullptr;
  int32_t* d_tokens = nullptr; size_t d_tokens_cap = 0;
  const int32_t* tok_src = nullptr;                       // device-fed step: the embedding reads the sampler's output in place
  int32_t* d_forced = nullptr; size_t d_forced_cap = 0;   // mi_score_tokens targets
  int32_t* d_gather = nullptr;                            // token positions whose hidden state feeds the head (mixed steps)
  float* d_rowpar = nullptr; size_t d_rowpar_cap = 0;     // per-row temperature | top_p of the current step
  void* deq_scratch = nullptr; size_t deq_cap = 0;        // [hi | lo] 16-bit copy of one int4 matrix (prefill GEMM)
  void* sk_ws = nullptr; size_t sk_ws_cap = 0;            // split-K partial tiles of gemm_skinny.hip
  unsigned* sk_ctr = nullptr; int sk_ctr_cap = 0;         // its per-tile-group arrival counters (zero between launches)
  int32_t* d_next = nullptr;      // tokens sampled by the last step [maxB]
  float* d_logprob = nullptr; float* d_prob0 = nullptr; float* d_rowstats = nullptr; float* d_uniforms = nullptr;
  int32_t* d_topk_ids = nullptr; float* d_topk_lp = nullptr;
  int32_t* d_bias_ids = nullptr; float* d_bias_vals = nullptr;
  int maxB = 0;
  int max_lora_rank = 0;
  Slot slots[NSLOT];
  int64_t next_ticket = 0;
  uint64_t step_counter = 0;
  // profiling
  std::string prof_name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  int opt_force_v1 = 0;
  int opt_fused_attn = 1;
  int opt_attn_mfma = 1;                 // decode attention on the matrix cores where the shape allows
  int opt_tile_weights = 1;
  int opt_prefill_gemm = 1;
  int opt_skinny_gemm = 1;      // decode steps of 9..128 rows (int4 / int8: 1..128): the split-K weight-streaming GEMM (gemm_skinny.hip)
  int cur_L = 0;                // tokens per sequence of the forward pass being enqueued
  // RMSNorm hand-over between gemm_skinny launches (GemvCall::sq_out / sq_in): the residual linear in front of a norm
  // leaves the rows' sums of squares, the normalised linear behind it uses them instead of an rmsnorm launch
  int opt_norm_handover = 1;    // round 2 (partial sums read in ONE round trip): Mistral-7B int4 +3.2 %, int8 +2.9 %, bf16 B = 16 +3.7 %, Qwen3-14B int4 +0.6 %
  int opt_prefill_x_terms = 2;  // float32 activations, prefill tile GEMM on dense bf16 weights: 16-bit terms of x (2 or 3 = exact)
  int opt_defer_norm = 1;       // float32 activations: RMSNorm row scale applied in the split-K kernel's epilogue (no norm launch)
  int opt_short_prefill_skinny = 1;   // short prefill / mixed calls on the weight-streaming kernel instead of the tile GEMM (gemv_rows)
  float* d_sq = nullptr;        // [4096 tile groups][16 rows]
  bool sq_valid = false; const void* sq_src = nullptr; int sq_parts = 0, sq_K = 0, sq_ld = 0;
  bool opt_gu8 = true;                   // decode GEMV of a dense gate|up matrix on its row-interleaved copy: 7 one-tile items per CU instead of 3.5 pairs (twice the matrix's bytes)
  bool opt_f16_hilo = true;              // f16 dense weights in the float32-activation mode through an exact [hi | lo] bf16 copy (matrix cores)
  int opt_fused_pairs = 0;               // bit 0: o_proj -> gate|up, bit 1: down_proj -> next q|k|v as one launch each.
                                         // Off: measured on Mistral-7B bf16 B=8 the in-launch seam costs what the kernel
                                         // boundary it replaces costs (bit 0: +-0 %, bit 1: -2 %), DESIGN.md section 5
  unsigned* d_seam_counter = nullptr;    // arrival counter of the in-launch seams (monotonic)
  int* d_seam_error = nullptr;           // set by a workgroup that gave up waiting at a seam
  int* h_seam_err = nullptr;             // pinned: [slot] = *d_seam_error behind that step's launches, [NSLOT] = synchronous reads
  unsigned seam_spin_limit = 1u << 20;   // polls before a workgroup gives up (option "seam_spin_limit"; tests force 0)
  unsigned seam_base = 0;
  int last_n = 0;                        // rows of the last enqueued step (device-resident token feed)                // value of *d_seam_counter once every enqueued launch has run
  void* xn = nullptr;            // [rows][max(H, I)] normalised activations of the prefill GEMMs
  void* xs = nullptr; size_t xs_cap = 0;   // [rows][3 K] bf16: float32 activations split three ways (float32-KV prefill)
  void* gk_ws = nullptr; size_t gk_cap = 0;   // float32 partial tiles of the K-split 128 x 128 tile GEMM
};

struct mi_kv {
  mi_engine* e = nullptr;
  int B = 0, cap = 0, dtype = MI_F32;
  bool quirk = false;            // float32 "PagedKVCache" semantics on a 16-bit model
  void* k = nullptr; void* v = nullptr;
  int32_t* d_off = nullptr;
  std::vector<int32_t> h_off;
  int32_t* d_rows = nullptr;     // [B] scratch: cache rows of the current call (continuous batching)
  float* partial = nullptr; int partial_splits = 0;
  int* counters = nullptr;       // [B*Hkv] split-arrival tickets of the fused decode attention
  // ---- block-paged form (mi_kv_create_paged): k / v are arenas [layer][block][Hkv][1 << bs_shift][D]; a row owns the
  // blocks its table names, in order; block 0 is never handed out (unassigned table entries point at it, so a clamped
  // address is always mapped memory); cap = bt_stride << bs_shift is the longest sequence a row can hold
  bool paged = false;
  int bs_shift = 0, nblocks = 0, bt_stride = 0;
  int32_t* d_btab = nullptr;
  std::vector<int32_t> h_btab;   // [B][bt_stride]
  std::vector<int> row_blocks;   // blocks in use per row
  std::vector<int> refcnt;       // per block: rows that map it + 1 if the prefix cache holds it
  std::vector<int> free_blocks;
  bool btab_dirty = false;
  // prefix cache: full blocks of prompts, keyed by the hash of ALL tokens up to and including the block; an entry keeps
  // the block's own tokens and its parent's key, so a hit is verified token by token (no reliance on the hash alone)
  struct PrefixEntry { int block; uint64_t parent; std::vector<int32_t> tokens; std::list<uint64_t>::iterator lru_it; };
  std::unordered_map<uint64_t, PrefixEntry> prefix;
  std::list<uint64_t> lru;       // front = most recently used
  int64_t stat_hit_tokens = 0, stat_lookup_tokens = 0, stat_evictions = 0;
};

namespace {

struct Prof {
  mi_engine* e; bool on; hipEvent_t a = nullptr, b = nullptr;
  Prof(mi_engine* e_, const char* name) : e(e_), on(!e_->prof_name.empty() && e_->prof_name == name) {
    if (on) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, e->stream); }
  }
  ~Prof() { if (on) { hipEventRecord(b, e->stream); e->prof_events.emplace_back(a, b); } }
};

int kind_of(const FusedLinear& f, int bits) {
  if (!f.quant) return f.dense_dtype == MI_F32 ? WK_F32 : (f.dense_dtype == MI_BF16 ? WK_BF16 : WK_F16);
  const int base = bits == 8 ? WK_Q8_F32 : WK_Q4_F32;
  return base + (f.scale_dtype == MI_F32 ? 0 : (f.scale_dtype == MI_BF16 ? 1 : 2));
}

int copy_in(void* dst, const void* src, size_t bytes, int on_device, hipStream_t st) {
  MI_HIP(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  MI_HIP(hipStreamSynchronize(st));
  return MI_OK;
}

// sub-tensor `part` of a fused linear: kind 0 = weight, 1 = scales, 2 = biases
int set_linear(mi_engine* e, FusedLinear& f, int part, int K, int kind, const void* data, const int64_t* shape,
               int ndim, int dtype, int on_device, const std::string& name) {
  const mi_model_desc& d = e->d;
  if (ndim != 2) return fail(MI_ERR_INVALID, name + ": expected a 2-D tensor");
  const int rows = f.part_rows[part];
  int total = 0, row0 = 0;
  for (int i = 0; i < f.n_parts; ++i) { if (i < part) row0 += f.part_rows[i]; total += f.part_rows[i]; }
  if (shape[0] != rows) return fail(MI_ERR_INVALID, name + ": wrong number of rows");
  if (kind == 0) {
    const bool packed = dtype == MI_U32;
    if (packed && d.quant_bits == 0) return fail(MI_ERR_INVALID, name + ": packed weights but desc.quant_bits == 0");
    const int64_t cols = packed ? (int64_t)K * d.quant_bits / 32 : K;
    if (shape[1] != cols) return fail(MI_ERR_INVALID, name + ": wrong number of columns");
    if (!packed && dtype != d.act_dtype)
      return fail(MI_ERR_UNSUPPORTED, name + ": dense weight dtype differs from the model dtype");
    if (f.w != nullptr && f.quant != packed) return fail(MI_ERR_UNSUPPORTED, name + ": fused parts mix dense and quantised weights");
    const size_t row_bytes = (size_t)cols * dtype_size(dtype);
    if (f.w == nullptr) {
      f.quant = packed; f.dense_dtype = packed ? -1 : dtype; f.w_bytes = row_bytes * total;
      MI_HIP(hipMalloc(&f.w, f.w_bytes));
    }
    MI_TRY(copy_in((char*)f.w + (size_t)row0 * row_bytes, data, row_bytes * rows, on_device, e->stream));
    f.seen_w[part] = 1;
  } else {
    if (d.quant_bits == 0) return fail(MI_ERR_INVALID, name + ": scales/biases but desc.quant_bits == 0");
    if (K % d.quant_group_size != 0) return fail(MI_ERR_INVALID, name + ": K not divisible by the group size");
    const int64_t cols = K / d.quant_group_size;
    if (shape[1] != cols) return fail(MI_ERR_INVALID, name + ": wrong number of groups");
    if (dtype != MI_F32 && dtype != MI_BF16 && dtype != MI_F16) return fail(MI_ERR_INVALID, name + ": bad dtype");
    if (f.scale_dtype >= 0 && f.scale_dtype != dtype) return fail(MI_ERR_UNSUPPORTED, name + ": mixed scale dtypes");
    f.scale_dtype = dtype;
    const size_t row_bytes = (size_t)cols * dtype_size(dtype);
    void*& dst = kind == 1 ? f.scales : f.biases;
    if (dst == nullptr) { f.s_bytes = row_bytes * total; MI_HIP(hipMalloc(&dst, f.s_bytes)); }
    MI_TRY(copy_in((char*)dst + (size_t)row0 * row_bytes, data, row_bytes * rows, on_device, e->stream));
    (kind == 1 ? f.seen_s : f.seen_b)[part] = 1;
  }
  return MI_OK;
}

int set_vector(mi_engine* e, void*& dst, int n, const void* data, const int64_t* shape, int ndim, int dtype,
               int on_device, const std::string& name) {
  if (ndim != 1 || shape[0] != n) return fail(MI_ERR_INVALID, name + ": wrong shape");
  if (dtype != e->d.act_dtype) return fail(MI_ERR_UNSUPPORTED, name + ": norm weight dtype differs from the model dtype");
  if (dst == nullptr) MI_HIP(hipMalloc(&dst, (size_t)n * dtype_size(dtype)));
  return copy_in(dst, data, (size_t)n * dtype_size(dtype), on_device, e->stream);
}

int finalize_linear(mi_engine* e, FusedLinear& f, int K, const std::string& name) {
  int total = 0;
  for (int i = 0; i < f.n_parts; ++i) {
    total += f.part_rows[i];
    if (!f.seen_w[i]) return fail(MI_ERR_NOTFOUND, name + ": weight not set");
    if (f.quant && (!f.seen_s[i] || !f.seen_b[i])) return fail(MI_ERR_NOTFOUND, name + ": scales/biases not set");
  }
  if (f.quant && f.scale_dtype != e->d.act_dtype)
    return fail(MI_ERR_UNSUPPORTED, name + ": scale dtype differs from the model dtype");
  f.W.wk = kind_of(f, e->d.quant_bits);
  f.W.w = f.w; f.W.scales = f.scales; f.W.biases = f.biases;
  f.W.N = total; f.W.K = K; f.W.group = e->d.quant_group_size > 0 ? e->d.quant_group_size : 64;
  // "repack weights": tile-major layout for the matrices the streaming kernels read (repack.hip)
  if (e->opt_tile_weights && tiled_supported(f.W.wk, f.W.N, f.W.K, f.W.group)) {
    void* dst = nullptr;
    if (hipMalloc(&dst, tiled_bytes(f.W.wk, f.W.N, f.W.K)) != hipSuccess)
      return fail(MI_ERR_RUNTIME, name + ": out of device memory repacking weights");
    MI_TRY(launch_repack_tiled(f.W, dst, e->stream));
    MI_HIP(hipStreamSynchronize(e->stream));
    hipFree(f.w); hipFree(f.scales); hipFree(f.biases);
    f.w = dst; f.scales = nullptr; f.biases = nullptr;
    f.W.w = dst; f.W.scales = nullptr; f.W.biases = nullptr; f.W.layout = 1;
  }
  return MI_OK;
}

int make_f32_copy(mi_engine* e, const void* src, int n, void** dst) {
  if (src == nullptr) { *dst = nullptr; return MI_OK; }
  if (e->d.act_dtype == MI_F32) { *dst = nullptr; return MI_OK; }
  MI_HIP(hipMalloc(dst, (size_t)n * sizeof(float)));
  return launch_convert(src, e->d.act_dtype, *dst, MI_F32, (size_t)n, e->stream);
}

void free_linear(FusedLinear& f) {
  hipFree(f.w); hipFree(f.scales); hipFree(f.biases); hipFree(f.w_hilo); hipFree(f.w_gu8);
  for (int i = 0; i < 2; ++i) { hipFree(f.lora_a[i]); hipFree(f.lora_b[i]); }
}

int ensure_workspace(mi_engine* e, size_t rows, size_t logit_rows, int B) {
  const mi_model_desc& d = e->d;
  if (rows > e->ws_rows) {
    MI_HIP(hipStreamSynchronize(e->stream));
    hipFree(e->h); hipFree(e->qkv); hipFree(e->q); hipFree(e->attn); hipFree(e->act); hipFree(e->lora_t);
    const size_t es = 4;  // sized for float32 activations (the widest mode)
    const size_t nqkv = (size_t)(d.num_heads + 2 * d.num_kv_heads) * d.head_dim;
    MI_HIP(hipMalloc(&e->h, rows * d.hidden_size * es));
    MI_HIP(hipMalloc(&e->qkv, rows * nqkv * es));
    MI_HIP(hipMalloc(&e->q, rows * (size_t)d.num_heads * d.head_dim * es));
    MI_HIP(hipMalloc(&e->attn, rows * (size_t)d.num_heads * d.head_dim * es));
    MI_HIP(hipMalloc(&e->act, rows * (size_t)d.intermediate_size * es));
    MI_HIP(hipMalloc(&e->lora_t, rows * 2 * 64 * sizeof(float)));
    hipFree(e->xn);
    MI_HIP(hipMalloc(&e->xn, rows * (size_t)d.hidden_size * es));
    e->ws_rows = rows;
  }
  if (logit_rows > e->ws_logit_rows) {
    MI_HIP(hipStreamSynchronize(e->stream));
    hipFree(e->logits);
    MI_HIP(hipMalloc(&e->logits, logit_rows * (size_t)d.vocab_size * sizeof(float)));
    e->ws_logit_rows = logit_rows;
  }
  if (rows > e->d_tokens_cap) {
    MI_HIP(hipStreamSynchronize(e->stream));
    hipFree(e->d_tokens);
    MI_HIP(hipMalloc(&e->d_tokens, rows * sizeof(int32_t)));
    e->d_tokens_cap = rows;
  }
  if (B > e->maxB) {
    MI_HIP(hipStreamSynchronize(e->stream));
    hipFree(e->d_next); hipFree(e->d_rowstats); hipFree(e->d_uniforms);      // (d_logprob / d_prob0 live in d_next's block)
    hipFree(e->d_topk_ids); hipFree(e->d_topk_lp);
    // tokens | logprobs | row-0 probabilities of a step in ONE block: one device-to-host copy per step instead of three
    MI_HIP(hipMalloc(&e->d_next, (size_t)3 * B * sizeof(int32_t)));
    e->d_logprob = (float*)(e->d_next + B);
    e->d_prob0 = (float*)(e->d_next + 2 * B);
    MI_HIP(hipMalloc(&e->d_rowstats, 2 * B * sizeof(float)));
    MI_HIP(hipMalloc(&e->d_uniforms, B * sizeof(float)));
    MI_HIP(hipMalloc(&e->d_topk_ids, (size_t)B * MI_MAX_TOP_LOGPROBS * sizeof(int32_t)));
    MI_HIP(hipMalloc(&e->d_topk_lp, (size_t)B * MI_MAX_TOP_LOGPROBS * sizeof(float)));
    for (auto& s : e->slots) {
      hipHostFree(s.tokens); hipHostFree(s.topk_ids); hipHostFree(s.topk_lp);       // (logprob / prob0 live in tokens' block)
      MI_HIP(hipHostMalloc(&s.tokens, (size_t)3 * B * sizeof(int32_t)));
      s.logprob = (float*)(s.tokens + B);
      s.prob0 = (float*)(s.tokens + 2 * B);
      MI_HIP(hipHostMalloc(&s.topk_ids, (size_t)B * MI_MAX_TOP_LOGPROBS * sizeof(int32_t)));
      MI_HIP(hipHostMalloc(&s.topk_lp, (size_t)B * MI_MAX_TOP_LOGPROBS * sizeof(float)));
      if (!s.ev) MI_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    }
    e->maxB = B;
  }
  return MI_OK;
}

// ---- persistent weight copies.  Both return true when the copy exists afterwards.  A failed allocation is not an error of
// the call: the runtime's error state is cleared, the copy is marked unavailable for this linear and the caller uses the
// path that reads the matrix as loaded.  The pointer is published only once the repack kernel has been enqueued successfully
// (same stream as every consumer).
bool hilo_wanted(const mi_engine* e, const FusedLinear& f) {
  return f.W.wk == WK_F16 && f.W.layout == 1 && e->opt_f16_hilo && !e->opt_force_v1 && f.W.K % 256 == 0;
}
bool ensure_hilo(mi_engine* e, FusedLinear& f) {
  if (f.w_hilo != nullptr) return true;
  if (f.hilo_failed || !hilo_wanted(e, f)) return false;
  void* p = nullptr;
  if (hipMalloc(&p, 2 * (size_t)f.W.N * f.W.K * sizeof(uint16_t)) != hipSuccess) { (void)hipGetLastError(); f.hilo_failed = true; return false; }
  if (launch_f16_to_hilo(f.W, p, e->stream) != MI_OK) { hipFree(p); f.hilo_failed = true; return false; }
  f.w_hilo = p;
  return true;
}
bool gu8_wanted(const mi_engine* e, const FusedLinear& f, int pair_offset) {
  const bool has_lora = f.W.lora_b[0] != nullptr || f.W.lora_b[1] != nullptr;
  return e->opt_gu8 && !e->opt_force_v1 && !has_lora && f.W.layout == 1 && (f.W.wk == WK_BF16 || f.W.wk == WK_F16) &&
         f.W.N == 2 * pair_offset && pair_offset % 16 == 0;
}
bool ensure_gu8(mi_engine* e, FusedLinear& f, int pair_offset) {
  if (f.w_gu8 != nullptr) return true;
  if (f.gu8_failed || !gu8_wanted(e, f, pair_offset)) return false;
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)f.W.N * f.W.K * sizeof(uint16_t)) != hipSuccess) { (void)hipGetLastError(); f.gu8_failed = true; return false; }
  if (launch_gate_up_interleave(f.W, pair_offset, p, e->stream) != MI_OK) { hipFree(p); f.gu8_failed = true; return false; }
  f.w_gu8 = p;
  return true;
}

// y = W x for `rows` rows, split into launches of at most 16 (MFMA) / 8 (generic) rows
int gemv_rows_on(mi_engine* e, const FusedLinear& f, const FusedLinear& f0, GemvCall c, size_t rows, size_t es_in, size_t es_out,
                 const char* prof, bool sq_was_valid);
int gemv_rows(mi_engine* e, const FusedLinear& f0, GemvCall c, size_t rows, size_t es_in, size_t es_out,
              const char* prof) {
  c.force_v1 = e->opt_force_v1;
  const bool sq_was_valid = e->sq_valid;      // whatever runs now consumes or invalidates the hand-over
  e->sq_valid = false;
  // An f16 model in the PagedKVCache mode (float32 activations): the matrix-core kernels of that mode multiply an exact
  // three-way bf16 split of x by bf16 weights, so an f16 matrix is used through its exact [hi | lo] bf16 copy (2 K
  // columns, x walked twice: GemvCall::kx) -- made by mi_engine_finalize (here only if the option was switched on later),
  // twice the matrix's bytes; without it (allocation failed) the exact VALU kernel reads the f16 matrix.  Everything that is
  // about x (norms, LoRA down-projection, the split) keeps the true K of f0.
  if (c.act == MI_F32 && hilo_wanted(e, f0) && ensure_hilo(e, const_cast<FusedLinear&>(f0))) {
    FusedLinear fh = f0;                       // (a view: the pointers stay owned by f0)
    fh.W.wk = WK_BF16; fh.W.w = f0.w_hilo; fh.W.K = 2 * f0.W.K;
    c.kx = f0.W.K;
    return gemv_rows_on(e, fh, f0, c, rows, es_in, es_out, prof, sq_was_valid);
  }
  return gemv_rows_on(e, f0, f0, c, rows, es_in, es_out, prof, sq_was_valid);
}

// f: the matrix the matmul kernels stream (f0 itself, or its [hi | lo] view); f0: the linear as loaded (true K, LoRA)
int gemv_rows_on(mi_engine* e, const FusedLinear& f, const FusedLinear& f0, GemvCall c, size_t rows, size_t es_in, size_t es_out,
                 const char* prof, bool sq_was_valid) {
  const int KT = f0.W.K;                      // the true K: columns of x
  // Which kernel for a call that is not a pure decode step (prefill, mixed step)?  Measured on Mistral-7B shapes, one prompt
  // of L tokens (tools/debug/prefill_sweep.py, ms for the whole call): dense 16-bit weights -- streaming kernel 4.2 / 5.1 /
  // 6.1 / 6.9 at L = 32 / 64 / 96 / 128 against 5.2 / 5.5 / 5.6 / 5.8 on the K-split 128 x 128 tile: the hand-over is at ~80 rows;
  // int4 -- 3.0 / 5.0 / 6.9 (L <= 96) against 15.5 on the tile path, which first writes a [hi | lo] 16-bit copy of every
  // matrix (4 x 54 us per layer): the streaming kernel keeps the call, in two row slabs up to twice its row limit.
  static const int dense_short = getenv("MI_SHORT_PREFILL_ROWS") ? atoi(getenv("MI_SHORT_PREFILL_ROWS")) : 64;
  const bool quant = wk_is_quant(f.W.wk);
  const size_t short_rows = e->opt_short_prefill_skinny ? (quant ? 128 : (size_t)dense_short) : 0;
  if (e->opt_skinny_gemm && e->cur_L != 1 && quant && e->opt_short_prefill_skinny && rows > 64 && rows <= 256 &&
      !gemm_skinny_supported(f.W, c, rows)) {
    // quantised weights, a call just above the streaming kernel's row limit: two slabs of rows, two reads of W
    const size_t half = (rows + 1) / 2;
    GemvCall probe = c; probe.pro = PRO_NONE;
    if (gemm_skinny_supported(f.W, probe, half) && f.W.lora_b[0] == nullptr && f.W.lora_b[1] == nullptr) {
      Prof pr(e, prof);
      if (c.pro == PRO_NORM) {
        MI_TRY(launch_rmsnorm_rows(c.x, c.ldx, c.norm_w, e->xn, KT, (int)rows, KT, c.eps, c.act, e->stream, true, c.rnd));
        c.x = e->xn; c.ldx = KT; c.pro = PRO_NONE;
      }
      const char* x0 = (const char*)c.x; char* o0 = (char*)c.out; char* r0 = (char*)c.resid;
      for (size_t r = 0; r < rows; r += half) {
        GemvCall cc = c;
        cc.M = (int)std::min(half, rows - r);
        cc.x = x0 + r * (size_t)c.ldx * es_in;
        if (o0) cc.out = o0 + r * (size_t)c.ldo * es_out;
        if (r0) cc.resid = r0 + r * (size_t)c.ldo * es_in;
        const size_t need = gemm_skinny_ws_bytes(f.W, cc, (size_t)cc.M);
        const int groups = gemm_skinny_groups(f.W, cc, (size_t)cc.M);
        if (need > e->sk_ws_cap || groups > e->sk_ctr_cap) {
          MI_HIP(hipStreamSynchronize(e->stream));
          if (need > e->sk_ws_cap) {
            hipFree(e->sk_ws); e->sk_ws = nullptr; e->sk_ws_cap = 0;
            MI_HIP(hipMalloc(&e->sk_ws, need));
            e->sk_ws_cap = need;
          }
          if (groups > e->sk_ctr_cap) {
            hipFree(e->sk_ctr); e->sk_ctr = nullptr; e->sk_ctr_cap = 0;
            const int cap = std::max(groups, 4096);
            MI_HIP(hipMalloc(&e->sk_ctr, (size_t)cap * sizeof(unsigned)));
            MI_HIP(hipMemsetAsync(e->sk_ctr, 0, (size_t)cap * sizeof(unsigned), e->stream));
            e->sk_ctr_cap = cap;
          }
        }
        MI_TRY(launch_gemm_skinny(f.W, cc, (size_t)cc.M, e->stream, e->sk_ws, e->sk_ctr));
      }
      return MI_OK;
    }
  }
  // decode steps: the streaming kernel up to 96 rows for dense weights -- above that the K-split tile GEMM is ahead
  // (Mistral-7B bf16, KV 512: 96 rows 7.18 vs 7.14 ms / step, 128 rows 8.71 vs 7.96)
  static const int decode_max_env = getenv("MI_SKINNY_DECODE_MAX") ? atoi(getenv("MI_SKINNY_DECODE_MAX")) : 0;   // A/B
  const size_t decode_max = decode_max_env > 0 ? (size_t)decode_max_env : (quant ? 128 : 96);
  if (e->opt_skinny_gemm && ((e->cur_L == 1 && rows <= decode_max) || rows <= short_rows) && gemm_skinny_supported(f.W, c, rows)) {
    // the decode step of a batch of 9..128 sequences (int4 / int8 weights: any batch up to 128): W is streamed once, K split over workgroups (gemm_skinny.hip).
    // Also a prefill of up to 128 rows in all (a short prompt, a few short prompts): at that size the op is a weight
    // stream, not a GEMM -- measured on Mistral-7B bf16, one prompt of 64 tokens: 13.0 ms through the 128 x 128 tile GEMM
    // (32 workgroups for N = 4096) against ~6 ms here.  The price: a sequence's prefill arithmetic (summation order) now
    // depends on whether the whole call is above or below 128 rows; inside one regime rows stay bit-independent of
    // their neighbours (tests/test_gpu_fullsize.py).
    Prof pr(e, prof);
    // (float32 activations: the split-K kernel neither leaves nor takes the row statistics -- always the norm launch)
    c.M = (int)rows;
    const int take_ld = (c.act != MI_F32 && c.pro == PRO_NORM) ? gemm_skinny_handover_ld(f.W, c, rows) : 0;
    const bool handed = sq_was_valid && e->opt_norm_handover && take_ld > 0 && take_ld == e->sq_ld &&
                        e->sq_src == c.x && e->sq_K == KT && c.ldx == KT;
    c.M = (int)rows;
    if (f.W.lora_b[0] != nullptr || f.W.lora_b[1] != nullptr) {     // (normalises on its own when c.pro says so)
      c.lora_t = e->lora_t; c.lora_t_ld = 128;
      MI_TRY(launch_lora_down(f0.W, c, e->lora_t, 128, e->stream));
    }
    // float32 activations without logical rounding (PagedKVCache mode after layer 0): the kernel applies the row scale
    // of the RMSNorm in its epilogue (gemm_skinny.hip "defer_norm") -- no norm launch, nothing waits for row statistics
    const bool defer = e->opt_defer_norm && c.pro == PRO_NORM && c.act == MI_F32 && c.rnd == RND_NONE && !handed;
    const bool q4_prep = gemm_q4_supported(f.W, c, rows);      // gemm_q4.hip: its preparation pass over x applies the RMSNorm
    if (handed) { c.sq_in = e->d_sq; c.sq_parts = e->sq_parts; }
    else if (c.pro == PRO_NORM && !defer && !q4_prep) {
      MI_TRY(launch_rmsnorm_rows(c.x, c.ldx, c.norm_w, e->xn, KT, (int)rows, KT, c.eps, c.act, e->stream, true, c.rnd));
      c.x = e->xn; c.ldx = KT; c.pro = PRO_NONE;
    }
    const size_t need = gemm_skinny_ws_bytes(f.W, c, rows);
    const int groups = gemm_skinny_groups(f.W, c, rows);
    const int tgroups = gemm_skinny_tile_groups(f.W, c, rows);
    const int leave_ld = (c.act != MI_F32 && c.epi == EPI_RESID) ? gemm_skinny_handover_ld(f.W, c, rows) : 0;
    const bool produce = e->opt_norm_handover && leave_ld > 0 && tgroups <= 64 && c.ldo == f.W.N;
    if (produce) {
      if (!e->d_sq) MI_HIP(hipMalloc(&e->d_sq, (size_t)4096 * 16 * sizeof(float)));     // (64 tile groups x <= 128 rows fit eight times)
      c.sq_out = e->d_sq;
    }
    if (need > e->sk_ws_cap || groups > e->sk_ctr_cap) {
      MI_HIP(hipStreamSynchronize(e->stream));
      if (need > e->sk_ws_cap) {
        hipFree(e->sk_ws); e->sk_ws = nullptr; e->sk_ws_cap = 0;
        MI_HIP(hipMalloc(&e->sk_ws, need));
        e->sk_ws_cap = need;
      }
      if (groups > e->sk_ctr_cap) {
        hipFree(e->sk_ctr); e->sk_ctr = nullptr; e->sk_ctr_cap = 0;
        const int cap = std::max(groups, 4096);
        MI_HIP(hipMalloc(&e->sk_ctr, (size_t)cap * sizeof(unsigned)));
        MI_HIP(hipMemsetAsync(e->sk_ctr, 0, (size_t)cap * sizeof(unsigned), e->stream));
        e->sk_ctr_cap = cap;
      }
    }
    MI_TRY(launch_gemm_skinny(f.W, c, rows, e->stream, e->sk_ws, e->sk_ctr));
    if (produce) { e->sq_valid = true; e->sq_src = c.resid; e->sq_parts = tgroups; e->sq_K = f.W.N; e->sq_ld = leave_ld; }
    return MI_OK;
  }
  if (e->opt_prefill_gemm && gemm_prefill_supported(f.W, c, rows)) {
    // prefill: one MFMA tile GEMM over all rows (the RMSNorm runs as its own row-wise kernel)
    Prof pr(e, prof);
    const GemvCall c_in = c;                 // (the LoRA down-projection reads the caller's x and normalises on its own)
    if (c.act == MI_F32) {      // PagedKVCache mode: (norm +) exact three-way split of x, then the same tile GEMM over 3 K
      const size_t need = split3_bytes(rows, KT);
      if (need > e->xs_cap) {
        MI_HIP(hipStreamSynchronize(e->stream));
        hipFree(e->xs); e->xs = nullptr; e->xs_cap = 0;
        MI_HIP(hipMalloc(&e->xs, need));
        e->xs_cap = need;
      }
      // dense bf16 weights: two terms (16+ mantissa bits of x, two walks of W; option "prefill_x_terms"); int4 / [hi | lo]
      // matrices keep the exact three (their own pair needs the 3 : 2 walk)
      const int terms = (e->opt_prefill_x_terms == 2 && f.W.wk == WK_BF16 && c.kx == 0 && c.rnd == RND_NONE) ? 2 : 3;
      MI_TRY(launch_split3_rows(c.x, c.ldx, c.pro == PRO_NORM ? c.norm_w : nullptr, c.eps, e->xs, (int)rows, KT, e->stream, terms));
      c.x = e->xs; c.ldx = terms * KT; c.pro = PRO_NONE;
    } else if (c.pro == PRO_NORM) {    // (one workgroup per row also here: one L2 round trip per row instead of a wave's 16)
      MI_TRY(launch_rmsnorm_rows(c.x, c.ldx, c.norm_w, e->xn, KT, (int)rows, KT, c.eps, c.act, e->stream, true));
      c.x = e->xn; c.ldx = KT; c.pro = PRO_NONE;
    }
    void* scratch = nullptr;
    if (wk_is_quant(f.W.wk)) {               // int4: the GEMM multiplies by a [hi | lo] 16-bit copy made on the fly
      const size_t need = dequant_hilo_bytes(f.W.N, KT);
      if (need > e->deq_cap) {
        MI_HIP(hipStreamSynchronize(e->stream));
        hipFree(e->deq_scratch);
        e->deq_scratch = nullptr; e->deq_cap = 0;
        MI_HIP(hipMalloc(&e->deq_scratch, need));
        e->deq_cap = need;
      }
      scratch = e->deq_scratch;
    }
    if (rows < 4096 && e->gk_ws == nullptr) {     // K-split partial tiles of the tile GEMMs (one prompt of a few hundred to a few thousand rows)
      e->gk_cap = (size_t)128 << 20;
      if (hipMalloc(&e->gk_ws, e->gk_cap) != hipSuccess) { e->gk_ws = nullptr; e->gk_cap = 0; }
    }
    MI_TRY(launch_gemm_prefill(f.W, c, rows, e->stream, scratch, rows < 4096 ? e->gk_ws : nullptr, e->gk_cap));
    if (f.W.lora_b[0] != nullptr || f.W.lora_b[1] != nullptr) {   // y = T(y + T(scale (x A) B)) on the adapted columns
      GemvCall cl = c.act == MI_F32 ? c_in : c;
      cl.M = (int)rows;
      MI_TRY(launch_lora_down(f0.W, cl, e->lora_t, 128, e->stream));
      MI_TRY(launch_lora_up_add(f0.W, cl, e->lora_t, 128, e->stream));
    }
    return MI_OK;
  }
  if (e->opt_skinny_gemm && c.act == MI_F32 && rows > 32 && gemm_skinny_supported(f.W, c, 32)) {
    // PagedKVCache mode, more than 32 rows (its prefill): 32 rows per launch of the float32-activation streaming kernel
    // (each row's arithmetic is that of a decode step, whatever the batch); the generic kernel took 8 rows per pass
    // over W.  A tile GEMM with the three-way split of x is the next step for this mode.
    Prof pr(e, prof);
    const bool lora32 = f.W.lora_b[0] != nullptr || f.W.lora_b[1] != nullptr;
    if (lora32) { GemvCall cl = c; cl.M = (int)rows; MI_TRY(launch_lora_down(f0.W, cl, e->lora_t, 128, e->stream)); }
    if (c.pro == PRO_NORM) {
      MI_TRY(launch_rmsnorm_rows(c.x, c.ldx, c.norm_w, e->xn, KT, (int)rows, KT, c.eps, c.act, e->stream, true, c.rnd));
      c.x = e->xn; c.ldx = KT; c.pro = PRO_NONE;
    }
    // workspace for every chunk size that is launched below: full 32-row chunks and the tail, whose plan (tile rows,
    // K split) is made for ITS row count and can need more partial-tile space than the 32-row plan
    GemvCall c16 = c; c16.M = 32;
    const size_t tail = rows % 32;
    size_t need = gemm_skinny_ws_bytes(f.W, c16, 32);
    int groups = gemm_skinny_groups(f.W, c16, 32);
    if (tail) {
      GemvCall ct = c; ct.M = (int)tail;
      need = std::max(need, gemm_skinny_ws_bytes(f.W, ct, tail));
      groups = std::max(groups, gemm_skinny_groups(f.W, ct, tail));
    }
    if (need > e->sk_ws_cap || groups > e->sk_ctr_cap) {
      MI_HIP(hipStreamSynchronize(e->stream));
      if (need > e->sk_ws_cap) {
        hipFree(e->sk_ws); e->sk_ws = nullptr; e->sk_ws_cap = 0;
        MI_HIP(hipMalloc(&e->sk_ws, need));
        e->sk_ws_cap = need;
      }
      if (groups > e->sk_ctr_cap) {
        hipFree(e->sk_ctr); e->sk_ctr = nullptr; e->sk_ctr_cap = 0;
        const int cap = std::max(groups, 4096);
        MI_HIP(hipMalloc(&e->sk_ctr, (size_t)cap * sizeof(unsigned)));
        MI_HIP(hipMemsetAsync(e->sk_ctr, 0, (size_t)cap * sizeof(unsigned), e->stream));
        e->sk_ctr_cap = cap;
      }
    }
    const char* x0 = (const char*)c.x; char* o0 = (char*)c.out; char* r0 = (char*)c.resid;
    for (size_t r = 0; r < rows; r += 32) {
      GemvCall cc = c;
      cc.M = (int)std::min<size_t>(32, rows - r);
      cc.x = x0 + r * (size_t)c.ldx * es_in;
      if (o0) cc.out = o0 + r * (size_t)c.ldo * es_out;
      if (r0) cc.resid = r0 + r * (size_t)c.ldo * es_in;
      if (lora32) { cc.lora_t = e->lora_t + r * 128; cc.lora_t_ld = 128; }
      MI_TRY(launch_gemm_skinny(f.W, cc, (size_t)cc.M, e->stream, e->sk_ws, e->sk_ctr));
    }
    return MI_OK;
  }
  // the generic per-8 / per-16-row kernels read the matrix as loaded (an f16 model falls back to its f16 weights, exact VALU)
  c.kx = 0;
  const bool has_lora = f0.W.lora_b[0] != nullptr || f0.W.lora_b[1] != nullptr;
  GemvCall probe = c; probe.M = (int)std::min<size_t>(rows, 16);
  const bool mfma = gemv_mfma_supported(f0.W, probe);
  const size_t step = mfma ? 16 : 8;
  // measurement: the MFMA kernel is stamped with its own dispatch begin/end (hipExtLaunchKernelGGL);
  // other paths are bracketed with events on the stream
  const bool selected = !e->prof_name.empty() && e->prof_name == prof;
  const bool stamp = selected && mfma && rows <= step;
  Prof pr(e, stamp ? "" : prof);
  if (stamp) {
    hipEvent_t a = nullptr, b = nullptr;
    hipEventCreate(&a); hipEventCreate(&b);
    c.ev_start = a; c.ev_stop = b;
    e->prof_events.emplace_back(a, b);
  }
  // SwiGLU on the matrix-core GEMV: the row-interleaved copy of the gate|up matrix (repack.hip; made by mi_engine_finalize,
  // here only if the option was switched on later; without it the paired-tile form below)
  const bool gu8 = mfma && c.epi == EPI_SWIGLU && gu8_wanted(e, f0, c.pair_offset) &&
                   ensure_gu8(e, const_cast<FusedLinear&>(f0), c.pair_offset);
  const char* x0 = (const char*)c.x; char* o0 = (char*)c.out; char* r0 = (char*)c.resid;
  for (size_t r = 0; r < rows; r += step) {
    GemvCall cc = c;
    cc.M = (int)std::min(step, rows - r);
    cc.x = x0 + r * (size_t)c.ldx * es_in;
    if (o0) cc.out = o0 + r * (size_t)c.ldo * es_out;
    if (r0) cc.resid = r0 + r * (size_t)c.ldo * es_in;
    if (has_lora) {
      cc.lora_t = e->lora_t + r * 128; cc.lora_t_ld = 128;
      MI_TRY(launch_lora_down(f0.W, cc, e->lora_t + r * 128, 128, e->stream));
    }
    if (gu8) {
      LinearW wv = f0.W;                       // (a view: the copy stays owned by f0)
      wv.w = f0.w_gu8;
      cc.epi = EPI_SWIGLU_GU8;
      MI_TRY(launch_gemv_mfma(wv, cc, e->stream));
      continue;
    }
    MI_TRY(launch_gemv(f0.W, cc, e->stream));
  }
  return MI_OK;
}

// decode: two dependent GEMVs (b reads a's output) in one launch when the MFMA pair kernel supports them
bool can_pair(mi_engine* e, const FusedLinear& fa, GemvCall a, const FusedLinear& fb, GemvCall b, size_t rows) {
  if (!e->opt_fused_pairs || rows > 8) return false;
  a.force_v1 = b.force_v1 = e->opt_force_v1;
  a.M = b.M = (int)rows;
  if (fa.W.lora_b[0] || fa.W.lora_b[1] || fb.W.lora_b[0] || fb.W.lora_b[1]) return false;
  return gemv_pair_supported(fa.W, a, fb.W, b);
}

int ensure_seam(mi_engine* e) {          // arrival counter + error flag of the in-launch seams
  if (e->d_seam_counter == nullptr) {
    MI_HIP(hipMalloc(&e->d_seam_counter, sizeof(unsigned)));
    MI_HIP(hipMalloc(&e->d_seam_error, sizeof(int)));
    MI_HIP(hipMemsetAsync(e->d_seam_counter, 0, sizeof(unsigned), e->stream));
    MI_HIP(hipMemsetAsync(e->d_seam_error, 0, sizeof(int), e->stream));
    MI_HIP(hipHostMalloc(&e->h_seam_err, (NSLOT + 1) * sizeof(int)));
    for (int i = 0; i <= NSLOT; ++i) e->h_seam_err[i] = 0;
    e->seam_base = 0;
  }
  return MI_OK;
}

// The seam's error flag travels with every step's results (one 4-byte copy behind the step's launches, only once a
// paired launch has ever run on this engine); a synchronous call reads it after its stream synchronisation.  A set
// flag means some workgroup stopped waiting and computed phase B from incomplete inputs: the call fails, the flag
// is cleared so that the engine stays usable.
int seam_record(mi_engine* e, int64_t ticket) {
  if (e->d_seam_counter == nullptr) return MI_OK;
  MI_HIP(hipMemcpyAsync(&e->h_seam_err[ticket % NSLOT], e->d_seam_error, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  return MI_OK;
}
int seam_fail(mi_engine* e) {
  hipMemsetAsync(e->d_seam_error, 0, sizeof(int), e->stream);
  return fail(MI_ERR_RUNTIME, "a workgroup gave up waiting at an in-launch seam (fused_gemv_pairs): the results of this call are invalid");
}
int seam_check_sync(mi_engine* e) {        // the stream is idle
  if (e->d_seam_counter == nullptr) return MI_OK;
  MI_HIP(hipMemcpyAsync(&e->h_seam_err[NSLOT], e->d_seam_error, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  MI_HIP(hipStreamSynchronize(e->stream));
  if (e->h_seam_err[NSLOT] == 0) return MI_OK;
  e->h_seam_err[NSLOT] = 0;
  return seam_fail(e);
}

int gemv_pair(mi_engine* e, const FusedLinear& fa, GemvCall a, const FusedLinear& fb, GemvCall b, size_t rows,
              const char* prof) {
  a.M = b.M = (int)rows;
  MI_TRY(ensure_seam(e));
  const bool selected = !e->prof_name.empty() && e->prof_name == prof;
  Prof pr(e, selected ? "" : prof);
  if (selected) {
    hipEvent_t x = nullptr, y = nullptr;
    hipEventCreate(&x); hipEventCreate(&y);
    a.ev_start = x; a.ev_stop = y;
    e->prof_events.emplace_back(x, y);
  }
  GemvSeam s{e->d_seam_counter, e->seam_base, e->d_seam_error, e->seam_spin_limit};
  MI_TRY(launch_gemv_pair(fa.W, a, fb.W, b, s, e->stream));
  e->seam_base += (unsigned)gemv_pair_grid();
  return MI_OK;
}


// ---- block-paged KV: host-side block management (the device only sees the table) ------------------------------------
uint64_t prefix_hash(uint64_t parent, const int32_t* toks, int n) {
  uint64_t h = parent ^ 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; ++i) { h ^= (uint64_t)(uint32_t)toks[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xD6E8FEB86659FD93ull; }
  return h ? h : 1;
}

void kv_release_block(mi_kv* kv, int blk) {
  if (blk <= 0) return;
  if (--kv->refcnt[blk] == 0) kv->free_blocks.push_back(blk);
}

// A prefix chain is only reachable from its first block (attach stops at the first miss), so a chain is made "recent"
// leaf first: afterwards its root is the most recently used entry and its deepest block the oldest of the chain --
// eviction (from the back) takes leaves before their parents and never strands children behind a missing parent.
void lru_touch_chain(mi_kv* kv, const std::vector<uint64_t>& chain) {
  for (auto h = chain.rbegin(); h != chain.rend(); ++h) {
    auto it = kv->prefix.find(*h);
    if (it == kv->prefix.end()) continue;
    kv->lru.erase(it->second.lru_it);
    kv->lru.push_front(*h);
    it->second.lru_it = kv->lru.begin();
  }
}

// a block nobody but the prefix cache holds can be taken back (least recently used first)
bool kv_evict_one(mi_kv* kv) {
  for (auto it = kv->lru.rbegin(); it != kv->lru.rend(); ++it) {
    auto pe = kv->prefix.find(*it);
    if (pe == kv->prefix.end()) continue;
    if (kv->refcnt[pe->second.block] == 1) {
      const int blk = pe->second.block;
      kv->lru.erase(pe->second.lru_it);
      kv->prefix.erase(pe);
      kv_release_block(kv, blk);
      ++kv->stat_evictions;
      return true;
    }
  }
  return false;
}

int kv_take_block(mi_kv* kv) {
  if (kv->free_blocks.empty() && !kv_evict_one(kv)) return -1;
  const int blk = kv->free_blocks.back();
  kv->free_blocks.pop_back();
  kv->refcnt[blk] = 1;
  return blk;
}

// the row's table covers `tokens` positions after this
int kv_ensure_blocks(mi_kv* kv, int row, int tokens) {
  if (!kv->paged) return MI_OK;
  const int need = (tokens + (1 << kv->bs_shift) - 1) >> kv->bs_shift;
  if (need > kv->bt_stride) return fail(MI_ERR_INVALID, "paged KV: sequence longer than the row's block table");
  while (kv->row_blocks[row] < need) {
    const int blk = kv_take_block(kv);
    if (blk < 0) return fail(MI_ERR_RUNTIME, "paged KV: the block arena is exhausted (every block is held by a live sequence)");
    kv->h_btab[(size_t)row * kv->bt_stride + kv->row_blocks[row]++] = blk;
    kv->btab_dirty = true;
  }
  return MI_OK;
}

void kv_release_row(mi_kv* kv, int row) {
  if (!kv->paged) return;
  for (int i = 0; i < kv->row_blocks[row]; ++i) {
    kv_release_block(kv, kv->h_btab[(size_t)row * kv->bt_stride + i]);
    kv->h_btab[(size_t)row * kv->bt_stride + i] = 0;
  }
  if (kv->row_blocks[row] > 0) kv->btab_dirty = true;
  kv->row_blocks[row] = 0;
}

int kv_upload_table(mi_kv* kv, hipStream_t st) {
  if (kv->paged && kv->btab_dirty) {
    // (pageable host memory: the copy is staged by the runtime before the call returns, so h_btab may change afterwards)
    MI_HIP(hipMemcpyAsync(kv->d_btab, kv->h_btab.data(), kv->h_btab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    kv->btab_dirty = false;
  }
  return MI_OK;
}

void kv_shape(const mi_kv* kv, AttnShape& s) {
  if (kv->paged) { s.btab = kv->d_btab; s.bt_stride = kv->bt_stride; s.bs_shift = kv->bs_shift; }
}

size_t kv_layer_elems(const mi_kv* kv, const mi_model_desc& d) {
  return kv->paged ? ((size_t)kv->nblocks * d.num_kv_heads << kv->bs_shift) * d.head_dim
                   : (size_t)kv->B * d.num_kv_heads * kv->cap * d.head_dim;
}

int choose_nsplit(const mi_kv* kv, int B, int Hkv, int L, const int32_t* rows = nullptr) {
  if (L != 1) return 1;
  int mx = 0;
  for (int b = 0; b < B; ++b) mx = std::max(mx, kv->h_off[rows ? rows[b] : b] + 1);
  int ns = (256 + B * Hkv - 1) / (B * Hkv);
  ns = std::min(ns, std::max(1, mx / 64));
  // long contexts: at most four 256-key rounds per workgroup -- eight when the batch alone fills the CUs (every extra
  // workgroup then costs a prologue and a merge on a CU that is already busy)
  const int per_wg = B * Hkv >= 256 ? 2048 : 1024;
  ns = std::max(ns, (mx + per_wg - 1) / per_wg);
  ns = std::max(1, std::min(ns, 16));
  // a workgroup walks its keys in rounds of 256 (float32 caches: 128): the smallest split count with the SAME number of
  // rounds has fewer workgroups to start and fewer partials to merge (KV length 1100, B = 8: 4 -> 3 splits, float32-KV
  // 1968 -> 1995 tok/s, bf16 2238 -> 2252; more splits are slower: 5 / 6 / 8 splits 2175 / 2173 / 2155)
  const int rk = kv->dtype == MI_F32 ? 128 : 256;
  auto rounds = [&](int n) { return ((mx - 1 + n - 1) / n + rk - 1) / rk; };     // (the cached keys; the new one is merged from registers)
  const int r0 = rounds(ns);
  while (ns > 1 && rounds(ns - 1) == r0) --ns;
  static const int ns_env = getenv("MI_ATTN_NSPLIT") ? atoi(getenv("MI_ATTN_NSPLIT")) : 0;   // A/B
  if (ns_env > 0) ns = std::min(ns_env, std::max(1, mx / 64));
  return std::max(1, std::min(ns, 16));
}

// the model forward for B*L tokens already in e->d_tokens; logits of the requested rows end
// up in e->logits (float32, [B][V] or [B*L][V]).
// `rows` (host, B distinct cache rows) = the call covers that subset of kv's rows, batch entry b <-> rows[b]
int forward_device(mi_engine* e, mi_kv* kv, int B, int L, bool all_pos, bool want_logits, const int32_t* rows = nullptr) {
  const mi_model_desc& d = e->d;
  const size_t R = (size_t)B * L;
  const bool quirk = kv->quirk;
  const int act = quirk ? MI_F32 : d.act_dtype;
  const size_t es = dtype_size(act);
  const int rndT = quirk ? (d.act_dtype == MI_BF16 ? RND_BF16 : RND_F16) : RND_NONE;
  const int H = d.hidden_size, D = d.head_dim, Hq = d.num_heads, Hkv = d.num_kv_heads, I = d.intermediate_size;
  const int nqkv = (Hq + 2 * Hkv) * D;
  hipStream_t st = e->stream;
  e->cur_L = L;
  e->sq_valid = false;

  auto row_of = [&](int b) { return rows ? rows[b] : b; };
  for (int b = 0; b < B; ++b)
    if (kv->h_off[row_of(b)] + L > kv->cap || kv->h_off[row_of(b)] + L > d.max_positions)
      return fail(MI_ERR_INVALID, "forward: KV capacity / max_positions exceeded (call mi_kv_reserve)");
  const int32_t* d_rows = nullptr;
  if (rows) {
    if (!kv->d_rows) MI_HIP(hipMalloc(&kv->d_rows, kv->B * sizeof(int32_t)));
    MI_HIP(hipMemcpyAsync(kv->d_rows, rows, B * sizeof(int32_t), hipMemcpyHostToDevice, st));
    d_rows = kv->d_rows;
  }
  for (int b = 0; b < B; ++b) MI_TRY(kv_ensure_blocks(kv, row_of(b), kv->h_off[row_of(b)] + L));
  MI_TRY(kv_upload_table(kv, st));

  { Prof pr(e, "embed");
    EmbedCall ec{e->tok_src ? e->tok_src : e->d_tokens, (int)R, act, rndT, e->h};
    MI_TRY(launch_embed(e->embed.W, ec, st)); }

  const size_t layer_elems = kv_layer_elems(kv, d);
  const size_t kes = dtype_size(kv->dtype);
  const int nsplit = choose_nsplit(kv, B, Hkv, L, rows);
  if (nsplit > 1) {
    const size_t need = R * Hq * nsplit * (D + 2);
    if (kv->partial == nullptr || kv->partial_splits < nsplit) {
      MI_HIP(hipStreamSynchronize(st));
      hipFree(kv->partial);
      MI_HIP(hipMalloc(&kv->partial, (size_t)kv->B * Hq * 16 * (D + 2) * sizeof(float)));
      kv->partial_splits = 16;
    }
    (void)need;
  }

  bool qkv_done = false;     // this layer's q|k|v already ran behind the previous layer's down_proj
  for (int li = 0; li < d.num_layers; ++li) {
    LayerW& lw = e->layers[li];
    const bool w32 = quirk && d.act_dtype != MI_F32;     // norm weights must match the activation storage type
    const void* in_norm = w32 ? lw.in_norm32 : lw.in_norm;
    const void* post_norm = w32 ? lw.post_norm32 : lw.post_norm;
    const void* q_norm = w32 ? lw.q_norm32 : lw.q_norm;
    const void* k_norm = w32 ? lw.k_norm32 : lw.k_norm;
    // everything up to the layer-0 attention still rounds to the model dtype in quirk mode
    const int rnd = (quirk && li == 0) ? rndT : RND_NONE;
    auto qkv_call = [&](int layer) {  // input_layernorm + q|k|v projections (llama.py:187,93)
      const LayerW& l2 = e->layers[layer];
      GemvCall c; c.x = e->h; c.ldx = H; c.act = act; c.rnd = (quirk && layer == 0) ? rndT : RND_NONE; c.pro = PRO_NORM;
      c.norm_w = w32 ? l2.in_norm32 : l2.in_norm;
      c.eps = d.rms_norm_eps; c.epi = EPI_STORE; c.out = e->qkv; c.ldo = nqkv;
      return c;
    };
    (void)in_norm;
    if (!qkv_done) MI_TRY(gemv_rows(e, lw.qkv, qkv_call(li), R, es, es, "gemv_qkv"));
    qkv_done = false;
    AttnShape s{B, L, Hq, Hkv, D, act, kv->dtype, rnd, kv->cap, d_rows};
    kv_shape(kv, s);
    void* kc = (char*)kv->k + (size_t)li * layer_elems * kes;
    void* vc = (char*)kv->v + (size_t)li * layer_elems * kes;
    // o_proj + residual (llama.py:143,188)
    GemvCall co; co.x = e->attn; co.ldx = Hq * D; co.act = act; co.rnd = RND_NONE; co.epi = EPI_RESID;
    co.resid = e->h; co.ldo = H;
    if (L == 1 && e->opt_fused_attn && attention_decode_supported(s)) {
      // decode: norm + RoPE + append + attention + split combine in one launch
      AttnDecodeCall ac{s, e->qkv, kc, vc, kv->d_off, q_norm, k_norm, d.rms_norm_eps, e->cos_tab, e->sin_tab,
                        e->attn, 1.0f / sqrtf((float)D), RND_NONE, nsplit, kv->partial, kv->counters,
                        e->opt_attn_mfma ? 0 : 1};
      if (B <= 32) {
        ac.n_host_off = B;
        for (int b = 0; b < B; ++b) { ac.host_row[b] = row_of(b); ac.host_off[b] = kv->h_off[row_of(b)]; }
      }
      Prof pr(e, "attn");
      MI_TRY(launch_attention_decode(ac, st));
    } else {
      { Prof pr(e, "rope_append");
        RopeAppendCall rc{s, e->qkv, e->q, kc, vc, kv->d_off, q_norm, k_norm, d.rms_norm_eps,
                          e->cos_tab, e->sin_tab, d.max_positions};
        MI_TRY(launch_rope_append(rc, st)); }
      { Prof pr(e, "attn");
        AttnShape sa = s; sa.rnd = RND_NONE;  // SDPA output dtype = promote(q, kv): float32 in quirk mode
        AttnCall ac{sa, e->q, kc, vc, kv->d_off, e->attn, 1.0f / sqrtf((float)D), nsplit, kv->partial,
                    e->opt_attn_mfma ? 0 : 1};
        MI_TRY(launch_attention(ac, st)); }
    }
    {
      // post_attention_layernorm + gate|up + SwiGLU (llama.py:189,165)
      GemvCall cg; cg.x = e->h; cg.ldx = H; cg.act = act; cg.rnd = RND_NONE; cg.pro = PRO_NORM; cg.norm_w = post_norm;
      cg.eps = d.rms_norm_eps; cg.epi = EPI_SWIGLU; cg.out = e->act; cg.ldo = I; cg.pair_offset = I;
      if ((e->opt_fused_pairs & 1) && can_pair(e, lw.o, co, lw.gate_up, cg, R)) {
        MI_TRY(gemv_pair(e, lw.o, co, lw.gate_up, cg, R, "gemv_o_gate_up"));
      } else {
        MI_TRY(gemv_rows(e, lw.o, co, R, es, es, "gemv_o"));
        MI_TRY(gemv_rows(e, lw.gate_up, cg, R, es, es, "gemv_gate_up"));
      }
    }
    {  // down_proj + residual (llama.py:165,190); with the next layer's q|k|v behind it in the same launch
      GemvCall c; c.x = e->act; c.ldx = I; c.act = act; c.rnd = RND_NONE; c.epi = EPI_RESID; c.resid = e->h; c.ldo = H;
      if ((e->opt_fused_pairs & 2) && li + 1 < d.num_layers &&
          can_pair(e, lw.down, c, e->layers[li + 1].qkv, qkv_call(li + 1), R)) {
        MI_TRY(gemv_pair(e, lw.down, c, e->layers[li + 1].qkv, qkv_call(li + 1), R, "gemv_down_qkv"));
        qkv_done = true;
      } else {
        MI_TRY(gemv_rows(e, lw.down, c, R, es, es, "gemv_down"));
      }
    }
  }
  if (want_logits) {  // final norm + lm_head / tied embedding (llama.py:231,249-252)
    const FusedLinear& head = d.tie_word_embeddings ? e->embed : e->lm_head;
    GemvCall c; c.act = act; c.rnd = RND_NONE; c.pro = PRO_NORM; c.norm_w = (quirk && d.act_dtype != MI_F32) ? e->final_norm32 : e->final_norm; c.eps = d.rms_norm_eps;
    c.epi = EPI_STORE_F32; c.out = e->logits; c.ldo = d.vocab_size;
    if (all_pos) { c.x = e->h; c.ldx = H; MI_TRY(gemv_rows(e, head, c, R, es, sizeof(float), "gemv_head")); }
    else { c.x = (char*)e->h + (size_t)(L - 1) * H * es; c.ldx = L * H; MI_TRY(gemv_rows(e, head, c, B, es, sizeof(float), "gemv_head")); }
  }
  MI_TRY(launch_advance_offsets(kv->d_off, d_rows, B, L, st));
  for (int b = 0; b < B; ++b) kv->h_off[row_of(b)] += L;
  return MI_OK;
}


// ---- one pass over the weights for rows that advance by DIFFERENT numbers of tokens (chunked prefill co-scheduled with
// the live decode rows): segment i = lens[i] tokens of cache row rows[i].  The decode rows (lens == 1) come first in
// the token buffer and share one decode-attention launch; every longer segment is one prefill-attention call on its
// own row.  All linear layers run ONCE over the concatenated tokens (tile GEMM from 32 tokens up), which is the point:
// the weights of a layer are read once for the 7 live sequences AND the 256-token chunk of an arriving prompt.
// Logits are produced for the last token of the segments with want[i] != 0, in segment order.
int forward_mixed(mi_engine* e, mi_kv* kv, const int32_t* rows, const int32_t* lens, const int32_t* want, int n, int n_out) {
  const mi_model_desc& d = e->d;
  const bool quirk = kv->quirk;
  const int act = quirk ? MI_F32 : d.act_dtype;
  const size_t es = dtype_size(act);
  const int rndT = quirk ? (d.act_dtype == MI_BF16 ? RND_BF16 : RND_F16) : RND_NONE;
  const int H = d.hidden_size, D = d.head_dim, Hq = d.num_heads, Hkv = d.num_kv_heads, I = d.intermediate_size;
  const int nqkv = (Hq + 2 * Hkv) * D;
  hipStream_t st = e->stream;
  int nd = 0;                                    // decode rows: the leading segments of length 1
  while (nd < n && lens[nd] == 1) ++nd;
  size_t R = 0;
  std::vector<size_t> tok0(n);
  for (int i = 0; i < n; ++i) {
    if (i >= nd && lens[i] == 1) return fail(MI_ERR_INVALID, "mixed step: one-token segments must come first");
    if (kv->h_off[rows[i]] + lens[i] > kv->cap || kv->h_off[rows[i]] + lens[i] > d.max_positions)
      return fail(MI_ERR_INVALID, "mixed step: KV capacity / max_positions exceeded (call mi_kv_reserve)");
    tok0[i] = R; R += (size_t)lens[i];
  }
  // Only decode rows: the decode step's kernels.  Otherwise the call is routed like a prefill of R rows (gemv_rows: the
  // weight-streaming kernel for short calls, the K-split 128 x 128 tile for a few hundred rows, plain tiles above).
  e->cur_L = R == (size_t)nd ? 1 : 2;
  e->sq_valid = false;
  if (!kv->d_rows) MI_HIP(hipMalloc(&kv->d_rows, kv->B * sizeof(int32_t)));
  MI_HIP(hipMemcpyAsync(kv->d_rows, rows, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  for (int i = 0; i < n; ++i) MI_TRY(kv_ensure_blocks(kv, rows[i], kv->h_off[rows[i]] + lens[i]));
  MI_TRY(kv_upload_table(kv, st));

  { Prof pr(e, "embed");
    EmbedCall ec{e->tok_src ? e->tok_src : e->d_tokens, (int)R, act, rndT, e->h};
    MI_TRY(launch_embed(e->embed.W, ec, st)); }

  const size_t layer_elems = kv_layer_elems(kv, d);
  const size_t kes = dtype_size(kv->dtype);
  const int nsplit = nd > 0 ? choose_nsplit(kv, nd, Hkv, 1, rows) : 1;
  if (nsplit > 1 && (kv->partial == nullptr || kv->partial_splits < nsplit)) {
    MI_HIP(hipStreamSynchronize(st));
    hipFree(kv->partial);
    MI_HIP(hipMalloc(&kv->partial, (size_t)kv->B * Hq * 16 * (D + 2) * sizeof(float)));
    kv->partial_splits = 16;
  }
  for (int li = 0; li < d.num_layers; ++li) {
    LayerW& lw = e->layers[li];
    const bool w32 = quirk && d.act_dtype != MI_F32;
    const void* post_norm = w32 ? lw.post_norm32 : lw.post_norm;
    const void* q_norm = w32 ? lw.q_norm32 : lw.q_norm;
    const void* k_norm = w32 ? lw.k_norm32 : lw.k_norm;
    const int rnd = (quirk && li == 0) ? rndT : RND_NONE;
    { GemvCall c; c.x = e->h; c.ldx = H; c.act = act; c.rnd = rnd; c.pro = PRO_NORM;
      c.norm_w = w32 ? lw.in_norm32 : lw.in_norm; c.eps = d.rms_norm_eps; c.epi = EPI_STORE; c.out = e->qkv; c.ldo = nqkv;
      MI_TRY(gemv_rows(e, lw.qkv, c, R, es, es, "gemv_qkv")); }
    void* kc = (char*)kv->k + (size_t)li * layer_elems * kes;
    void* vc = (char*)kv->v + (size_t)li * layer_elems * kes;
    if (nd > 0) {                                // the decode rows: tokens [0, nd)
      AttnShape s{nd, 1, Hq, Hkv, D, act, kv->dtype, rnd, kv->cap, kv->d_rows};
      kv_shape(kv, s);
      if (e->opt_fused_attn && attention_decode_supported(s)) {
        AttnDecodeCall ac{s, e->qkv, kc, vc, kv->d_off, q_norm, k_norm, d.rms_norm_eps, e->cos_tab, e->sin_tab,
                          e->attn, 1.0f / sqrtf((float)D), RND_NONE, nsplit, kv->partial, kv->counters,
                          e->opt_attn_mfma ? 0 : 1};
        if (nd <= 32) {
          ac.n_host_off = nd;
          for (int b = 0; b < nd; ++b) { ac.host_row[b] = rows[b]; ac.host_off[b] = kv->h_off[rows[b]]; }
        }
        Prof pr(e, "attn");
        MI_TRY(launch_attention_decode(ac, st));
      } else {
        RopeAppendCall rc{s, e->qkv, e->q, kc, vc, kv->d_off, q_norm, k_norm, d.rms_norm_eps, e->cos_tab, e->sin_tab, d.max_positions};
        MI_TRY(launch_rope_append(rc, st));
        AttnShape sa = s; sa.rnd = RND_NONE;
        AttnCall ac{sa, e->q, kc, vc, kv->d_off, e->attn, 1.0f / sqrtf((float)D), nsplit, kv->partial, e->opt_attn_mfma ? 0 : 1};
        MI_TRY(launch_attention(ac, st));
      }
    }
    for (int i = nd; i < n; ++i) {               // every chunk: rope + append + causal attention on its own row
      AttnShape s{1, lens[i], Hq, Hkv, D, act, kv->dtype, rnd, kv->cap, kv->d_rows + i};
      kv_shape(kv, s);
      const void* qkv_i = (const char*)e->qkv + tok0[i] * (size_t)nqkv * es;
      void* q_i = (char*)e->q + tok0[i] * (size_t)Hq * D * es;
      void* o_i = (char*)e->attn + tok0[i] * (size_t)Hq * D * es;
      RopeAppendCall rc{s, qkv_i, q_i, kc, vc, kv->d_off, q_norm, k_norm, d.rms_norm_eps, e->cos_tab, e->sin_tab, d.max_positions};
      MI_TRY(launch_rope_append(rc, st));
      AttnShape sa = s; sa.rnd = RND_NONE;
      AttnCall ac{sa, q_i, kc, vc, kv->d_off, o_i, 1.0f / sqrtf((float)D), 1, nullptr, e->opt_attn_mfma ? 0 : 1};
      MI_TRY(launch_attention(ac, st));
    }
    { GemvCall co; co.x = e->attn; co.ldx = Hq * D; co.act = act; co.rnd = RND_NONE; co.epi = EPI_RESID; co.resid = e->h; co.ldo = H;
      MI_TRY(gemv_rows(e, lw.o, co, R, es, es, "gemv_o"));
      GemvCall cg; cg.x = e->h; cg.ldx = H; cg.act = act; cg.rnd = RND_NONE; cg.pro = PRO_NORM; cg.norm_w = post_norm;
      cg.eps = d.rms_norm_eps; cg.epi = EPI_SWIGLU; cg.out = e->act; cg.ldo = I; cg.pair_offset = I;
      MI_TRY(gemv_rows(e, lw.gate_up, cg, R, es, es, "gemv_gate_up"));
      GemvCall c; c.x = e->act; c.ldx = I; c.act = act; c.rnd = RND_NONE; c.epi = EPI_RESID; c.resid = e->h; c.ldo = H;
      MI_TRY(gemv_rows(e, lw.down, c, R, es, es, "gemv_down")); }
  }
  if (n_out > 0) {                               // hidden states of the wanted positions -> compact rows -> head
    std::vector<int32_t> idx;
    for (int i = 0; i < n; ++i) if (want[i]) idx.push_back((int32_t)(tok0[i] + lens[i] - 1));
    if (!e->d_gather) MI_HIP(hipMalloc(&e->d_gather, 4096 * sizeof(int32_t)));
    if (idx.size() > 4096) return fail(MI_ERR_INVALID, "mixed step: too many sampled rows");
    MI_HIP(hipMemcpyAsync(e->d_gather, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MI_TRY(launch_gather_rows(e->h, (size_t)H * es, e->d_gather, (int)idx.size(), e->xn, st));
    const FusedLinear& head = d.tie_word_embeddings ? e->embed : e->lm_head;
    GemvCall c; c.act = act; c.rnd = RND_NONE; c.pro = PRO_NORM; c.norm_w = (quirk && d.act_dtype != MI_F32) ? e->final_norm32 : e->final_norm;
    c.eps = d.rms_norm_eps; c.epi = EPI_STORE_F32; c.out = e->logits; c.ldo = d.vocab_size; c.x = e->xn; c.ldx = H;
    const int keep_L = e->cur_L; e->cur_L = 1;   // the few sampled rows go through the decode step's head kernel
    const int rc_head = gemv_rows(e, head, c, idx.size(), es, sizeof(float), "gemv_head");
    e->cur_L = keep_L;
    MI_TRY(rc_head);
  }
  if (nd > 0) MI_TRY(launch_advance_offsets(kv->d_off, kv->d_rows, nd, 1, st));
  for (int i = nd; i < n; ++i) MI_TRY(launch_advance_offsets(kv->d_off, kv->d_rows + i, 1, lens[i], st));
  for (int i = 0; i < n; ++i) kv->h_off[rows[i]] += lens[i];
  return MI_OK;
}

int check_call(mi_engine* e, mi_kv* kv, int B, int L) {
  if (!e || !kv) return fail(MI_ERR_INVALID, "null handle");
  if (!e->finalized) return fail(MI_ERR_INVALID, "engine not finalized");
  if (kv->e != e) return fail(MI_ERR_INVALID, "kv belongs to another engine");
  if (B != kv->B) return fail(MI_ERR_INVALID, "batch size mismatch (PagedKVCache batch size mismatch, base.py:125)");
  if (L < 1) return fail(MI_ERR_INVALID, "L must be >= 1");
  MI_HIP(hipSetDevice(e->device));
  return MI_OK;
}

int upload_tokens(mi_engine* e, const int32_t* tokens, int B, int L) {
  const size_t R = (size_t)B * L;
  for (size_t i = 0; i < R; ++i)
    if (tokens[i] < 0 || tokens[i] >= e->d.vocab_size) return fail(MI_ERR_INVALID, "token id out of range");
  MI_HIP(hipMemcpyAsync(e->d_tokens, tokens, R * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  return MI_OK;
}

// ABI guard (mi355_decode.h): a binding built against another layout of the struct is refused, never read
int check_params(const mi_sample_params* sp) {
  if (sp != nullptr && sp->struct_size != sizeof(mi_sample_params))
    return fail(MI_ERR_INVALID, "mi_sample_params.struct_size = " + std::to_string(sp->struct_size) + ", this library (ABI " +
                                    std::to_string(MI_ABI_VERSION) + ") expects " + std::to_string(sizeof(mi_sample_params)) +
                                    ": the binding was generated from another version of mi355_decode.h");
  return MI_OK;
}

int run_sample(mi_engine* e, int B, const mi_sample_params* sp, const int32_t* forced = nullptr) {
  mi_sample_params def{}; def.struct_size = sizeof(def); def.temperature = 0.f; def.top_p = 1.f; def.stream_position = -1;
  if (!sp) sp = &def;
  hipStream_t st = e->stream;
  if (sp->n_logit_bias > 0) {
    if (sp->n_logit_bias > 4096) return fail(MI_ERR_INVALID, "too many logit_bias entries");
    if (!e->d_bias_ids) { MI_HIP(hipMalloc(&e->d_bias_ids, 4096 * sizeof(int32_t))); MI_HIP(hipMalloc(&e->d_bias_vals, 4096 * sizeof(float))); }
    MI_HIP(hipMemcpyAsync(e->d_bias_ids, sp->logit_bias_ids, sp->n_logit_bias * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(e->d_bias_vals, sp->logit_bias_values, sp->n_logit_bias * sizeof(float), hipMemcpyHostToDevice, st));
  }
  if (sp->uniforms && !forced) MI_HIP(hipMemcpyAsync(e->d_uniforms, sp->uniforms, B * sizeof(float), hipMemcpyHostToDevice, st));
  const bool per_row = sp->row_temperature != nullptr && sp->row_top_p != nullptr && !forced;
  if (per_row) {
    if ((size_t)B > e->d_rowpar_cap) {
      MI_HIP(hipStreamSynchronize(st));
      hipFree(e->d_rowpar);
      MI_HIP(hipMalloc(&e->d_rowpar, 2 * (size_t)B * sizeof(float)));
      e->d_rowpar_cap = B;
    }
    MI_HIP(hipMemcpyAsync(e->d_rowpar, sp->row_temperature, B * sizeof(float), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(e->d_rowpar + B, sp->row_top_p, B * sizeof(float), hipMemcpyHostToDevice, st));
  }
  Prof pr(e, "sample");
  SampleCall sc{};
  sc.logits = e->logits; sc.B = B; sc.V = e->d.vocab_size; sc.rnd = RND_NONE;
  sc.temperature = sp->temperature; sc.top_p = sp->top_p;
  sc.n_bias = sp->n_logit_bias; sc.bias_ids = e->d_bias_ids; sc.bias_vals = e->d_bias_vals;
  sc.forced = forced;
  sc.uniforms = (sp->uniforms && !forced) ? e->d_uniforms : nullptr; sc.seed = sp->seed; sc.step = sp->stream_position >= 0 ? (uint64_t)sp->stream_position : e->step_counter++;
  sc.top_logprobs = sp->top_logprobs;
  sc.lp_temp = sp->logprobs_at_temperature;
  sc.row_temp = per_row ? e->d_rowpar : nullptr;
  sc.row_top_p = per_row ? e->d_rowpar + B : nullptr;
  sc.tokens_out = e->d_next; sc.logprob_out = e->d_logprob; sc.prob_row0_out = forced ? nullptr : e->d_prob0;
  sc.topk_ids = e->d_topk_ids; sc.topk_logprobs = e->d_topk_lp; sc.row_stats = e->d_rowstats;
  return launch_sample(sc, st);
}

}  // namespace

// =========================================================================================
extern "C" {

const char* mi_last_error(void) { return g_err.c_str(); }
const char* mi_version(void) { return "mi355_decode 0.2 (gfx950)"; }
int mi_abi_version(void) { return MI_ABI_VERSION; }

int mi_engine_create(const mi_model_desc* desc, int device, mi_engine** out) {
  if (!desc || !out) return fail(MI_ERR_INVALID, "null argument");
  if (desc->struct_size != sizeof(mi_model_desc))
    return fail(MI_ERR_INVALID, "mi_model_desc.struct_size = " + std::to_string(desc->struct_size) + ", this library (ABI " +
                                    std::to_string(MI_ABI_VERSION) + ") expects " + std::to_string(sizeof(mi_model_desc)));
  const mi_model_desc& d = *desc;
  if (d.arch != MI_ARCH_LLAMA && d.arch != MI_ARCH_QWEN3) return fail(MI_ERR_UNSUPPORTED, "unsupported arch");
  if (d.hidden_size <= 0 || d.num_layers <= 0 || d.num_heads <= 0 || d.num_kv_heads <= 0 || d.head_dim <= 0 ||
      d.intermediate_size <= 0 || d.vocab_size <= 0 || d.max_positions <= 0)
    return fail(MI_ERR_INVALID, "model dimensions must be positive");
  if (d.num_heads % d.num_kv_heads != 0) return fail(MI_ERR_INVALID, "num_heads must be a multiple of num_kv_heads");
  if (d.act_dtype != MI_F32 && d.act_dtype != MI_BF16 && d.act_dtype != MI_F16) return fail(MI_ERR_INVALID, "bad act_dtype");
  if (d.quant_bits != 0 && d.quant_bits != 4 && d.quant_bits != 8) return fail(MI_ERR_UNSUPPORTED, "quant_bits must be 0, 4 or 8");
  if (d.hidden_size % 8 || d.intermediate_size % 8 || (d.num_heads * d.head_dim) % 8)
    return fail(MI_ERR_UNSUPPORTED, "hidden / intermediate sizes must be multiples of 8");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MI_ERR_RUNTIME, "no HIP device available (this library has no CPU backend)");
  if (device < 0 || device >= ndev) return fail(MI_ERR_INVALID, "bad device index");
  MI_HIP(hipSetDevice(device));
  mi_engine* e = new mi_engine();
  e->d = d; e->device = device;
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail(MI_ERR_RUNTIME, "hipStreamCreate failed"); }
  e->layers.resize(d.num_layers);
  const int D = d.head_dim;
  for (auto& l : e->layers) {
    l.qkv.n_parts = 3; l.qkv.part_rows[0] = d.num_heads * D; l.qkv.part_rows[1] = d.num_kv_heads * D; l.qkv.part_rows[2] = d.num_kv_heads * D;
    l.o.n_parts = 1; l.o.part_rows[0] = d.hidden_size;
    l.gate_up.n_parts = 2; l.gate_up.part_rows[0] = d.intermediate_size; l.gate_up.part_rows[1] = d.intermediate_size;
    l.down.n_parts = 1; l.down.part_rows[0] = d.hidden_size;
  }
  e->embed.n_parts = 1; e->embed.part_rows[0] = d.vocab_size;
  e->lm_head.n_parts = 1; e->lm_head.part_rows[0] = d.vocab_size;
  *out = e;
  return MI_OK;
}

void mi_engine_destroy(mi_engine* e) {
  if (!e) return;
  hipSetDevice(e->device);
  hipStreamSynchronize(e->stream);
  for (auto& l : e->layers) {
    free_linear(l.qkv); free_linear(l.o); free_linear(l.gate_up); free_linear(l.down);
    hipFree(l.in_norm); hipFree(l.post_norm); hipFree(l.q_norm); hipFree(l.k_norm);
    hipFree(l.in_norm32); hipFree(l.post_norm32); hipFree(l.q_norm32); hipFree(l.k_norm32);
  }
  hipFree(e->final_norm32); hipFree(e->xn); hipFree(e->xs); hipFree(e->gk_ws);
  free_linear(e->embed); free_linear(e->lm_head);
  hipFree(e->final_norm); hipFree(e->cos_tab); hipFree(e->sin_tab);
  hipFree(e->h); hipFree(e->qkv); hipFree(e->q); hipFree(e->attn); hipFree(e->act); hipFree(e->logits); hipFree(e->lora_t); hipFree(e->d_forced); hipFree(e->d_gather);
  hipFree(e->d_seam_counter); hipFree(e->d_seam_error); hipHostFree(e->h_seam_err); hipFree(e->d_rowpar); hipFree(e->deq_scratch);
  hipFree(e->sk_ws); hipFree(e->sk_ctr); hipFree(e->d_sq);
  hipFree(e->d_tokens); hipFree(e->d_next); hipFree(e->d_rowstats);
  hipFree(e->d_uniforms); hipFree(e->d_topk_ids); hipFree(e->d_topk_lp); hipFree(e->d_bias_ids); hipFree(e->d_bias_vals);
  for (auto& s : e->slots) {
    hipHostFree(s.tokens); hipHostFree(s.topk_ids); hipHostFree(s.topk_lp);
    if (s.ev) hipEventDestroy(s.ev);
  }
  for (auto& p : e->prof_events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  hipStreamDestroy(e->stream);
  delete e;
}

int mi_engine_set_tensor(mi_engine* e, const char* name_c, const void* data, const int64_t* shape, int ndim,
                         int dtype, int on_device) {
  if (!e || !name_c || !data || !shape) return fail(MI_ERR_INVALID, "null argument");
  if (e->finalized) return fail(MI_ERR_INVALID, "engine already finalized");
  MI_HIP(hipSetDevice(e->device));
  const mi_model_desc& d = e->d;
  std::string name(name_c);
  int kind = -1;
  std::string base;
  auto strip = [&](const char* suf, int k) {
    const size_t n = strlen(suf);
    if (name.size() > n && name.compare(name.size() - n, n, suf) == 0) { base = name.substr(0, name.size() - n); kind = k; }
  };
  strip(".weight", 0); strip(".scales", 1); strip(".biases", 2);
  if (kind < 0) return fail(MI_ERR_NOTFOUND, "unknown tensor: " + name);
  const int H = d.hidden_size, I = d.intermediate_size, QD = d.num_heads * d.head_dim;
  if (base == "model.embed_tokens") return set_linear(e, e->embed, 0, H, kind, data, shape, ndim, dtype, on_device, name);
  if (base == "lm_head") {
    if (d.tie_word_embeddings) return fail(MI_ERR_NOTFOUND, "lm_head given but tie_word_embeddings is set: " + name);
    return set_linear(e, e->lm_head, 0, H, kind, data, shape, ndim, dtype, on_device, name);
  }
  if (base == "model.norm" && kind == 0) return set_vector(e, e->final_norm, H, data, shape, ndim, dtype, on_device, name);
  const std::string pre = "model.layers.";
  if (base.compare(0, pre.size(), pre) != 0) return fail(MI_ERR_NOTFOUND, "unknown tensor: " + name);
  const size_t dot = base.find('.', pre.size());
  if (dot == std::string::npos) return fail(MI_ERR_NOTFOUND, "unknown tensor: " + name);
  int li = -1;
  try { li = std::stoi(base.substr(pre.size(), dot - pre.size())); } catch (...) { li = -1; }
  if (li < 0 || li >= d.num_layers) return fail(MI_ERR_NOTFOUND, "layer index out of range: " + name);
  const std::string sub = base.substr(dot + 1);
  LayerW& l = e->layers[li];
  if (sub == "self_attn.q_proj") return set_linear(e, l.qkv, 0, H, kind, data, shape, ndim, dtype, on_device, name);
  if (sub == "self_attn.k_proj") return set_linear(e, l.qkv, 1, H, kind, data, shape, ndim, dtype, on_device, name);
  if (sub == "self_attn.v_proj") return set_linear(e, l.qkv, 2, H, kind, data, shape, ndim, dtype, on_device, name);
  if (sub == "self_attn.o_proj") return set_linear(e, l.o, 0, QD, kind, data, shape, ndim, dtype, on_device, name);
  if (sub == "mlp.gate_proj") return set_linear(e, l.gate_up, 0, H, kind, data, shape, ndim, dtype, on_device, name);
  if (sub == "mlp.up_proj") return set_linear(e, l.gate_up, 1, H, kind, data, shape, ndim, dtype, on_device, name);
  if (sub == "mlp.down_proj") return set_linear(e, l.down, 0, I, kind, data, shape, ndim, dtype, on_device, name);
  if (kind == 0) {
    if (sub == "input_layernorm") return set_vector(e, l.in_norm, H, data, shape, ndim, dtype, on_device, name);
    if (sub == "post_attention_layernorm") return set_vector(e, l.post_norm, H, data, shape, ndim, dtype, on_device, name);
    if (d.arch == MI_ARCH_QWEN3 && sub == "self_attn.q_norm") return set_vector(e, l.q_norm, d.head_dim, data, shape, ndim, dtype, on_device, name);
    if (d.arch == MI_ARCH_QWEN3 && sub == "self_attn.k_norm") return set_vector(e, l.k_norm, d.head_dim, data, shape, ndim, dtype, on_device, name);
  }
  return fail(MI_ERR_NOTFOUND, "unknown tensor: " + name);
}

int mi_engine_set_lora(mi_engine* e, int layer, const char* proj, const void* A, const void* B, int rank, float scale,
                       int dtype, int on_device) {
  if (!e || !proj || !A || !B) return fail(MI_ERR_INVALID, "null argument");
  if (layer < 0 || layer >= e->d.num_layers) return fail(MI_ERR_INVALID, "layer out of range");
  if (rank < 1 || rank > 64) return fail(MI_ERR_UNSUPPORTED, "LoRA rank must be in [1,64]");
  if (dtype != MI_F32 && dtype != MI_BF16 && dtype != MI_F16) return fail(MI_ERR_INVALID, "bad LoRA dtype");
  if (dtype != MI_F32 && dtype != e->d.act_dtype) return fail(MI_ERR_UNSUPPORTED, "LoRA dtype must be float32 or the model dtype");
  if (dtype != MI_F32) return fail(MI_ERR_UNSUPPORTED, "16-bit LoRA factors are not supported yet (float32 only)");
  MI_HIP(hipSetDevice(e->device));
  const mi_model_desc& d = e->d;
  LayerW& l = e->layers[layer];
  const std::string p(proj);
  FusedLinear* f = nullptr; int row0 = 0, n = 0, K = d.hidden_size;
  const int QD = d.num_heads * d.head_dim, KD = d.num_kv_heads * d.head_dim;
  if (p == "self_attn.q_proj") { f = &l.qkv; row0 = 0; n = QD; }
  else if (p == "self_attn.k_proj") { f = &l.qkv; row0 = QD; n = KD; }
  else if (p == "self_attn.v_proj") { f = &l.qkv; row0 = QD + KD; n = KD; }
  else if (p == "self_attn.o_proj") { f = &l.o; row0 = 0; n = d.hidden_size; K = QD; }
  else if (p == "mlp.down_proj") { f = &l.down; row0 = 0; n = d.hidden_size; K = d.intermediate_size; }
  else return fail(MI_ERR_UNSUPPORTED, "LoRA on " + p + " is not supported (q/k/v/o/down only)");
  int slot = -1;
  for (int i = 0; i < 2; ++i) if (f->lora_b[i] != nullptr && f->W.lora_row0[i] == row0) slot = i;   // hot-swap
  if (slot < 0) for (int i = 0; i < 2; ++i) if (f->lora_b[i] == nullptr) { slot = i; break; }
  if (slot < 0) return fail(MI_ERR_UNSUPPORTED, "at most two adapted projections per fused matrix");
  MI_HIP(hipStreamSynchronize(e->stream));
  hipFree(f->lora_a[slot]); hipFree(f->lora_b[slot]);
  f->lora_a[slot] = f->lora_b[slot] = nullptr;
  MI_HIP(hipMalloc(&f->lora_a[slot], (size_t)K * rank * sizeof(float)));
  MI_HIP(hipMalloc(&f->lora_b[slot], (size_t)rank * n * sizeof(float)));
  MI_TRY(copy_in(f->lora_a[slot], A, (size_t)K * rank * sizeof(float), on_device, e->stream));
  MI_TRY(copy_in(f->lora_b[slot], B, (size_t)rank * n * sizeof(float), on_device, e->stream));
  f->W.lora_a[slot] = f->lora_a[slot]; f->W.lora_b[slot] = f->lora_b[slot];
  f->W.lora_row0[slot] = row0; f->W.lora_n[slot] = n; f->W.lora_rank[slot] = rank; f->W.lora_scale[slot] = scale;
  return MI_OK;
}

int mi_engine_finalize(mi_engine* e) {
  if (!e) return fail(MI_ERR_INVALID, "null engine");
  if (e->finalized) return MI_OK;
  MI_HIP(hipSetDevice(e->device));
  const mi_model_desc& d = e->d;
  const int H = d.hidden_size, QD = d.num_heads * d.head_dim;
  for (int i = 0; i < d.num_layers; ++i) {
    LayerW& l = e->layers[i];
    const std::string p = "model.layers." + std::to_string(i);
    MI_TRY(finalize_linear(e, l.qkv, H, p + ".self_attn.{q,k,v}_proj"));
    MI_TRY(finalize_linear(e, l.o, QD, p + ".self_attn.o_proj"));
    MI_TRY(finalize_linear(e, l.gate_up, H, p + ".mlp.{gate,up}_proj"));
    MI_TRY(finalize_linear(e, l.down, d.intermediate_size, p + ".mlp.down_proj"));
    if (!l.in_norm || !l.post_norm) return fail(MI_ERR_NOTFOUND, p + ": layernorm weights not set");
    if (d.arch == MI_ARCH_QWEN3 && (!l.q_norm || !l.k_norm)) return fail(MI_ERR_NOTFOUND, p + ": q_norm/k_norm not set");
    MI_TRY(make_f32_copy(e, l.in_norm, H, &l.in_norm32));
    MI_TRY(make_f32_copy(e, l.post_norm, H, &l.post_norm32));
    MI_TRY(make_f32_copy(e, l.q_norm, d.head_dim, &l.q_norm32));
    MI_TRY(make_f32_copy(e, l.k_norm, d.head_dim, &l.k_norm32));
    // persistent second copies, while mem_get_info still tells a scheduler the truth about what is left for its KV arena
    (void)ensure_gu8(e, l.gate_up, d.intermediate_size);
    if (d.act_dtype == MI_F16) { (void)ensure_hilo(e, l.qkv); (void)ensure_hilo(e, l.o); (void)ensure_hilo(e, l.gate_up); (void)ensure_hilo(e, l.down); }
  }
  MI_TRY(finalize_linear(e, e->embed, H, "model.embed_tokens"));
  if (!d.tie_word_embeddings) MI_TRY(finalize_linear(e, e->lm_head, H, "lm_head"));
  if (d.act_dtype == MI_F16) (void)ensure_hilo(e, d.tie_word_embeddings ? e->embed : e->lm_head);
  if (!e->final_norm) return fail(MI_ERR_NOTFOUND, "model.norm.weight not set");
  MI_TRY(make_f32_copy(e, e->final_norm, H, &e->final_norm32));
  const size_t n = (size_t)d.max_positions * (d.head_dim / 2);
  MI_HIP(hipMalloc(&e->cos_tab, n * sizeof(float)));
  MI_HIP(hipMalloc(&e->sin_tab, n * sizeof(float)));
  MI_TRY(launch_rope_tables(e->cos_tab, e->sin_tab, d.max_positions, d.head_dim, d.rope_theta, d.rope_scale, e->stream));
  MI_HIP(hipStreamSynchronize(e->stream));
  e->finalized = true;
  return MI_OK;
}

// ---- KV ---------------------------------------------------------------------------------
int mi_kv_create(mi_engine* e, int batch, int capacity_tokens, int kv_dtype, mi_kv** out) {
  if (!e || !out) return fail(MI_ERR_INVALID, "null argument");
  if (batch < 1 || capacity_tokens < 1) return fail(MI_ERR_INVALID, "batch and capacity must be positive");
  if (capacity_tokens > e->d.max_positions) return fail(MI_ERR_INVALID, "capacity exceeds desc.max_positions");
  MI_HIP(hipSetDevice(e->device));
  mi_kv* kv = new mi_kv();
  kv->e = e; kv->B = batch; kv->cap = capacity_tokens;
  if (kv_dtype == MI_KV_MODEL || kv_dtype == e->d.act_dtype) { kv->dtype = e->d.act_dtype; kv->quirk = false; }
  else if (kv_dtype == MI_F32) { kv->dtype = MI_F32; kv->quirk = true; }
  else { delete kv; return fail(MI_ERR_UNSUPPORTED, "kv dtype must be the model dtype or float32"); }
  const size_t bytes = (size_t)e->d.num_layers * batch * e->d.num_kv_heads * capacity_tokens * e->d.head_dim * dtype_size(kv->dtype);
  const size_t nctr = (size_t)batch * e->d.num_kv_heads;
  if (hipMalloc(&kv->k, bytes) != hipSuccess || hipMalloc(&kv->v, bytes) != hipSuccess ||
      hipMalloc(&kv->d_off, batch * sizeof(int32_t)) != hipSuccess ||
      hipMalloc(&kv->counters, nctr * sizeof(int)) != hipSuccess) {
    hipFree(kv->k); hipFree(kv->v); hipFree(kv->d_off); hipFree(kv->counters); delete kv;
    return fail(MI_ERR_RUNTIME, "out of device memory allocating the KV cache");
  }
  MI_HIP(hipMemsetAsync(kv->counters, 0, nctr * sizeof(int), e->stream));
  // zero-filled like the reference's mx.zeros (base.py:71-72,111-112)
  MI_HIP(hipMemsetAsync(kv->k, 0, bytes, e->stream));
  MI_HIP(hipMemsetAsync(kv->v, 0, bytes, e->stream));
  MI_HIP(hipMemsetAsync(kv->d_off, 0, batch * sizeof(int32_t), e->stream));
  MI_HIP(hipStreamSynchronize(e->stream));
  kv->h_off.assign(batch, 0);
  *out = kv;
  return MI_OK;
}

int mi_kv_create_paged(mi_engine* e, int slots, int block_tokens, int n_blocks, int max_tokens_per_row, int kv_dtype, mi_kv** out) {
  if (!e || !out) return fail(MI_ERR_INVALID, "null argument");
  if (slots < 1 || n_blocks < 2 || max_tokens_per_row < 1) return fail(MI_ERR_INVALID, "slots, blocks and tokens per row must be positive (block 0 is reserved)");
  if (block_tokens < 16 || (block_tokens & (block_tokens - 1)) != 0) return fail(MI_ERR_INVALID, "block_tokens must be a power of two >= 16");
  if (max_tokens_per_row > e->d.max_positions) return fail(MI_ERR_INVALID, "max_tokens_per_row exceeds desc.max_positions");
  MI_HIP(hipSetDevice(e->device));
  mi_kv* kv = new mi_kv();
  kv->e = e; kv->B = slots; kv->paged = true;
  while ((1 << kv->bs_shift) < block_tokens) ++kv->bs_shift;
  kv->nblocks = n_blocks;
  kv->bt_stride = (max_tokens_per_row + block_tokens - 1) / block_tokens;
  kv->cap = std::min(kv->bt_stride * block_tokens, e->d.max_positions);
  if (kv_dtype == MI_KV_MODEL || kv_dtype == e->d.act_dtype) { kv->dtype = e->d.act_dtype; kv->quirk = false; }
  else if (kv_dtype == MI_F32) { kv->dtype = MI_F32; kv->quirk = true; }
  else { delete kv; return fail(MI_ERR_UNSUPPORTED, "kv dtype must be the model dtype or float32"); }
  const size_t bytes = (size_t)e->d.num_layers * n_blocks * e->d.num_kv_heads * block_tokens * e->d.head_dim * dtype_size(kv->dtype);
  const size_t nctr = (size_t)slots * e->d.num_kv_heads;
  const size_t ntab = (size_t)slots * kv->bt_stride;
  if (hipMalloc(&kv->k, bytes) != hipSuccess || hipMalloc(&kv->v, bytes) != hipSuccess ||
      hipMalloc(&kv->d_off, slots * sizeof(int32_t)) != hipSuccess || hipMalloc(&kv->counters, nctr * sizeof(int)) != hipSuccess ||
      hipMalloc(&kv->d_btab, ntab * sizeof(int32_t)) != hipSuccess) {
    hipFree(kv->k); hipFree(kv->v); hipFree(kv->d_off); hipFree(kv->counters); hipFree(kv->d_btab); delete kv;
    return fail(MI_ERR_RUNTIME, "out of device memory allocating the paged KV arena");
  }
  MI_HIP(hipMemsetAsync(kv->counters, 0, nctr * sizeof(int), e->stream));
  MI_HIP(hipMemsetAsync(kv->k, 0, bytes, e->stream));
  MI_HIP(hipMemsetAsync(kv->v, 0, bytes, e->stream));
  MI_HIP(hipMemsetAsync(kv->d_off, 0, slots * sizeof(int32_t), e->stream));
  MI_HIP(hipMemsetAsync(kv->d_btab, 0, ntab * sizeof(int32_t), e->stream));
  MI_HIP(hipStreamSynchronize(e->stream));
  kv->h_off.assign(slots, 0);
  kv->h_btab.assign(ntab, 0);
  kv->row_blocks.assign(slots, 0);
  kv->refcnt.assign(n_blocks, 0);
  for (int b = n_blocks - 1; b >= 1; --b) kv->free_blocks.push_back(b);     // block 0 stays out of circulation
  *out = kv;
  return MI_OK;
}

int mi_kv_prefix_attach(mi_kv* kv, int row, const int32_t* tokens, int n, int* n_reused) {
  if (!kv || !tokens || !n_reused) return fail(MI_ERR_INVALID, "null argument");
  if (!kv->paged) return fail(MI_ERR_INVALID, "prefix reuse needs a paged KV (mi_kv_create_paged)");
  if (row < 0 || row >= kv->B) return fail(MI_ERR_INVALID, "row out of range");
  if (kv->h_off[row] != 0 || kv->row_blocks[row] != 0) return fail(MI_ERR_INVALID, "prefix attach: the row must be empty (mi_kv_reset_row first)");
  const int bs = 1 << kv->bs_shift;
  uint64_t parent = 0;
  int reused = 0;
  std::vector<uint64_t> chain;
  // full blocks only, and at least one token of the prompt is left to run (its logits are needed)
  while (reused + bs <= n - 1 && kv->row_blocks[row] < kv->bt_stride) {
    const uint64_t h = prefix_hash(parent, tokens + reused, bs);
    auto it = kv->prefix.find(h);
    if (it == kv->prefix.end() || it->second.parent != parent ||
        memcmp(it->second.tokens.data(), tokens + reused, bs * sizeof(int32_t)) != 0) break;
    kv->h_btab[(size_t)row * kv->bt_stride + kv->row_blocks[row]++] = it->second.block;
    ++kv->refcnt[it->second.block];
    chain.push_back(h);
    parent = h;
    reused += bs;
  }
  lru_touch_chain(kv, chain);
  kv->stat_lookup_tokens += n;
  kv->stat_hit_tokens += reused;
  if (reused > 0) {
    kv->btab_dirty = true;
    kv->h_off[row] = reused;
    MI_HIP(hipSetDevice(kv->e->device));
    MI_HIP(hipMemsetD32Async((hipDeviceptr_t)(kv->d_off + row), reused, 1, kv->e->stream));
  }
  *n_reused = reused;
  return MI_OK;
}

int mi_kv_prefix_publish(mi_kv* kv, int row, const int32_t* tokens, int n) {
  if (!kv || !tokens) return fail(MI_ERR_INVALID, "null argument");
  if (!kv->paged) return fail(MI_ERR_INVALID, "prefix reuse needs a paged KV (mi_kv_create_paged)");
  if (row < 0 || row >= kv->B) return fail(MI_ERR_INVALID, "row out of range");
  if (n > kv->h_off[row]) return fail(MI_ERR_INVALID, "prefix publish: those tokens are not in the row's cache yet");
  const int bs = 1 << kv->bs_shift;
  uint64_t parent = 0;
  std::vector<uint64_t> chain;
  for (int i = 0; (i + 1) * bs <= n; ++i) {
    const uint64_t h = prefix_hash(parent, tokens + i * bs, bs);
    auto it = kv->prefix.find(h);
    if (it == kv->prefix.end()) {
      const int blk = kv->h_btab[(size_t)row * kv->bt_stride + i];
      mi_kv::PrefixEntry pe;
      pe.block = blk; pe.parent = parent; pe.tokens.assign(tokens + i * bs, tokens + (i + 1) * bs);
      kv->lru.push_front(h);
      pe.lru_it = kv->lru.begin();
      kv->prefix.emplace(h, std::move(pe));
      ++kv->refcnt[blk];                      // the cache's own reference: the block outlives the row
    } else if (it->second.parent != parent || memcmp(it->second.tokens.data(), tokens + i * bs, bs * sizeof(int32_t)) != 0) {
      break;                                  // a different prefix owns this key: leave it (and everything behind it) alone
    }
    chain.push_back(h);
    parent = h;
  }
  lru_touch_chain(kv, chain);
  return MI_OK;
}

int mi_kv_prefix_clear(mi_kv* kv) {
  if (!kv) return fail(MI_ERR_INVALID, "null kv");
  for (auto& p : kv->prefix) kv_release_block(kv, p.second.block);
  kv->prefix.clear();
  kv->lru.clear();
  return MI_OK;
}

int mi_kv_stats(const mi_kv* kv, int64_t* out, int n) {
  if (!kv || !out) return fail(MI_ERR_INVALID, "null argument");
  int64_t evictable = 0;           // published blocks that only the cache holds: what kv_evict_one can actually give back
  for (const auto& pe : kv->prefix) evictable += kv->refcnt[pe.second.block] == 1 ? 1 : 0;
  const int64_t v[7] = {kv->paged ? (int64_t)kv->free_blocks.size() : -1, kv->paged ? (int64_t)kv->nblocks - 1 : -1,
                        (int64_t)kv->prefix.size(), kv->stat_hit_tokens, kv->stat_lookup_tokens, kv->stat_evictions, evictable};
  for (int i = 0; i < n && i < 7; ++i) out[i] = v[i];
  return MI_OK;
}

void mi_kv_destroy(mi_kv* kv) {
  if (!kv) return;
  hipSetDevice(kv->e->device);
  hipStreamSynchronize(kv->e->stream);
  hipFree(kv->k); hipFree(kv->v); hipFree(kv->d_off); hipFree(kv->partial); hipFree(kv->counters); hipFree(kv->d_rows);
  hipFree(kv->d_btab);
  delete kv;
}

int mi_kv_reset(mi_kv* kv, int batch) {
  if (!kv) return fail(MI_ERR_INVALID, "null kv");
  if (batch != kv->B) return fail(MI_ERR_INVALID, "mi_kv_reset: batch differs from the allocated batch (create a new kv)");
  MI_HIP(hipSetDevice(kv->e->device));
  MI_HIP(hipMemsetAsync(kv->d_off, 0, kv->B * sizeof(int32_t), kv->e->stream));
  MI_HIP(hipStreamSynchronize(kv->e->stream));
  std::fill(kv->h_off.begin(), kv->h_off.end(), 0);
  for (int r = 0; r < kv->B; ++r) kv_release_row(kv, r);
  return MI_OK;
}

int mi_kv_reset_row(mi_kv* kv, int row) {
  if (!kv) return fail(MI_ERR_INVALID, "null kv");
  if (row < 0 || row >= kv->B) return fail(MI_ERR_INVALID, "mi_kv_reset_row: row out of range");
  MI_HIP(hipSetDevice(kv->e->device));
  MI_HIP(hipMemsetAsync(kv->d_off + row, 0, sizeof(int32_t), kv->e->stream));   // ordered behind the steps already enqueued
  kv->h_off[row] = 0;
  kv_release_row(kv, row);      // (paged: the blocks go back to the pool; whoever gets them next writes behind the steps in flight)
  return MI_OK;
}

int mi_kv_reserve(mi_kv* kv, int capacity_tokens) {
  if (!kv) return fail(MI_ERR_INVALID, "null kv");
  if (capacity_tokens <= kv->cap) return MI_OK;
  mi_engine* e = kv->e;
  if (kv->paged) return fail(MI_ERR_INVALID, "paged KV: a row holds at most block_tokens x table length tokens (create a larger table)");
  if (capacity_tokens > e->d.max_positions) return fail(MI_ERR_INVALID, "capacity exceeds desc.max_positions");
  MI_HIP(hipSetDevice(e->device));
  const size_t es = dtype_size(kv->dtype);
  const size_t slabs = (size_t)e->d.num_layers * kv->B * e->d.num_kv_heads;
  const size_t old_pitch = (size_t)kv->cap * e->d.head_dim * es, new_pitch = (size_t)capacity_tokens * e->d.head_dim * es;
  void* nk = nullptr; void* nv = nullptr;
  if (hipMalloc(&nk, slabs * new_pitch) != hipSuccess || hipMalloc(&nv, slabs * new_pitch) != hipSuccess) {
    hipFree(nk); hipFree(nv);
    return fail(MI_ERR_RUNTIME, "out of device memory growing the KV cache");
  }
  MI_HIP(hipMemsetAsync(nk, 0, slabs * new_pitch, e->stream));
  MI_HIP(hipMemsetAsync(nv, 0, slabs * new_pitch, e->stream));
  MI_HIP(hipMemcpy2DAsync(nk, new_pitch, kv->k, old_pitch, old_pitch, slabs, hipMemcpyDeviceToDevice, e->stream));
  MI_HIP(hipMemcpy2DAsync(nv, new_pitch, kv->v, old_pitch, old_pitch, slabs, hipMemcpyDeviceToDevice, e->stream));
  MI_HIP(hipStreamSynchronize(e->stream));
  hipFree(kv->k); hipFree(kv->v);
  kv->k = nk; kv->v = nv; kv->cap = capacity_tokens;
  return MI_OK;
}

int mi_kv_offsets(const mi_kv* kv, int32_t* out) {
  if (!kv || !out) return fail(MI_ERR_INVALID, "null argument");
  for (int b = 0; b < kv->B; ++b) out[b] = kv->h_off[b];
  return MI_OK;
}

int mi_kv_capacity(const mi_kv* kv) { return kv ? kv->cap : 0; }

// ---- forward / sampling --------------------------------------------------------------------
int mi_forward(mi_engine* e, mi_kv* kv, const int32_t* tokens, int B, int L, float* logits_out, int all_pos) {
  MI_TRY(check_call(e, kv, B, L));
  if (!tokens) return fail(MI_ERR_INVALID, "null tokens");
  const size_t R = (size_t)B * L;
  MI_TRY(ensure_workspace(e, R, all_pos ? R : (size_t)B, B));
  MI_TRY(upload_tokens(e, tokens, B, L));
  MI_TRY(forward_device(e, kv, B, L, all_pos != 0, logits_out != nullptr));
  if (logits_out)
    MI_HIP(hipMemcpyAsync(logits_out, e->logits, (all_pos ? R : (size_t)B) * e->d.vocab_size * sizeof(float),
                          hipMemcpyDeviceToHost, e->stream));
  MI_HIP(hipStreamSynchronize(e->stream));
  return seam_check_sync(e);
}

int mi_score_tokens(mi_engine* e, mi_kv* kv, const int32_t* tokens, const int32_t* targets, int B, int L,
                    const mi_sample_params* sp, float* logprob_out, int32_t* topk_ids, float* topk_logprobs) {
  MI_TRY(check_params(sp));
  MI_TRY(check_call(e, kv, B, L));
  if (!tokens || !targets || !logprob_out) return fail(MI_ERR_INVALID, "null argument");
  const size_t R = (size_t)B * L;
  if (R > (size_t)(1 << 20)) return fail(MI_ERR_INVALID, "score: too many positions in one call");
  const int k = sp ? sp->top_logprobs : 0;
  if (k > 0 && (!topk_ids || !topk_logprobs)) return fail(MI_ERR_INVALID, "score: top_logprobs > 0 needs output buffers");
  MI_TRY(ensure_workspace(e, R, R, (int)R));
  hipStream_t st = e->stream;
  if (R > e->d_forced_cap) {
    MI_HIP(hipStreamSynchronize(st));
    hipFree(e->d_forced);
    MI_HIP(hipMalloc(&e->d_forced, R * sizeof(int32_t)));
    e->d_forced_cap = R;
  }
  MI_TRY(upload_tokens(e, tokens, B, L));
  MI_HIP(hipMemcpyAsync(e->d_forced, targets, R * sizeof(int32_t), hipMemcpyHostToDevice, st));
  MI_TRY(forward_device(e, kv, B, L, true, true));
  MI_TRY(run_sample(e, (int)R, sp, e->d_forced));
  MI_HIP(hipMemcpyAsync(logprob_out, e->d_logprob, R * sizeof(float), hipMemcpyDeviceToHost, st));
  if (k > 0) {
    MI_HIP(hipMemcpyAsync(topk_ids, e->d_topk_ids, R * k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipMemcpyAsync(topk_logprobs, e->d_topk_lp, R * k * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  MI_HIP(hipStreamSynchronize(st));
  return seam_check_sync(e);
}

int mi_step_enqueue(mi_engine* e, mi_kv* kv, const int32_t* tokens_in, int B, int L, const mi_sample_params* sp,
                    int64_t* ticket) {
  MI_TRY(check_params(sp));
  MI_TRY(check_call(e, kv, B, L));
  if (!ticket) return fail(MI_ERR_INVALID, "null ticket");
  if (!tokens_in && L != 1) return fail(MI_ERR_INVALID, "device-resident token feed needs L == 1");
  if (!tokens_in && e->maxB < B) return fail(MI_ERR_INVALID, "no previous step to take tokens from");
  MI_TRY(ensure_workspace(e, (size_t)B * L, (size_t)B, B));
  hipStream_t st = e->stream;
  if (tokens_in) MI_TRY(upload_tokens(e, tokens_in, B, L));
  e->tok_src = tokens_in ? nullptr : e->d_next;            // (no copy: the sampler overwrites d_next only at the end of this step)
  const int frc = forward_device(e, kv, B, L, false, true);
  e->tok_src = nullptr;
  MI_TRY(frc);
  MI_TRY(run_sample(e, B, sp));
  e->last_n = B;
  const int64_t t = e->next_ticket++;
  Slot& s = e->slots[t % NSLOT];
  s.B = B; s.topk = sp ? sp->top_logprobs : 0; s.ticket = t;
  MI_HIP(hipMemcpyAsync(s.tokens, e->d_next, ((size_t)2 * e->maxB + B) * sizeof(int32_t), hipMemcpyDeviceToHost, st));   // tokens | logprobs | row-0 probabilities
  if (s.topk > 0) {
    MI_HIP(hipMemcpyAsync(s.topk_ids, e->d_topk_ids, (size_t)B * s.topk * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipMemcpyAsync(s.topk_lp, e->d_topk_lp, (size_t)B * s.topk * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  MI_TRY(seam_record(e, t));
  MI_HIP(hipEventRecord(s.ev, st));
  *ticket = t;
  return MI_OK;
}

int mi_step_enqueue_rows(mi_engine* e, mi_kv* kv, const int32_t* rows, int n, const int32_t* tokens_in, int L,
                         const mi_sample_params* sp, int64_t* ticket) {
  if (!e || !kv || !rows) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(check_params(sp));
  if (n < 1 || n > kv->B) return fail(MI_ERR_INVALID, "mi_step_enqueue_rows: n must be in [1, batch of the kv]");
  for (int i = 0; i < n; ++i) {
    if (rows[i] < 0 || rows[i] >= kv->B) return fail(MI_ERR_INVALID, "mi_step_enqueue_rows: row out of range");
    for (int j = 0; j < i; ++j) if (rows[j] == rows[i]) return fail(MI_ERR_INVALID, "mi_step_enqueue_rows: duplicate row");
  }
  MI_TRY(check_call(e, kv, kv->B, L));
  if (!ticket) return fail(MI_ERR_INVALID, "null ticket");
  if (!tokens_in && L != 1) return fail(MI_ERR_INVALID, "device-resident token feed needs L == 1");
  if (!tokens_in && e->last_n != n) return fail(MI_ERR_INVALID, "device-resident token feed needs the row set of the previous step");
  MI_TRY(ensure_workspace(e, (size_t)n * L, (size_t)n, std::max(n, kv->B)));
  hipStream_t st = e->stream;
  if (tokens_in) MI_TRY(upload_tokens(e, tokens_in, n, L));
  e->tok_src = tokens_in ? nullptr : e->d_next;
  const int frc = forward_device(e, kv, n, L, false, true, rows);
  e->tok_src = nullptr;
  MI_TRY(frc);
  MI_TRY(run_sample(e, n, sp));
  e->last_n = n;
  const int64_t t = e->next_ticket++;
  Slot& s = e->slots[t % NSLOT];
  s.B = n; s.topk = sp ? sp->top_logprobs : 0; s.ticket = t;
  MI_HIP(hipMemcpyAsync(s.tokens, e->d_next, ((size_t)2 * e->maxB + n) * sizeof(int32_t), hipMemcpyDeviceToHost, st));   // tokens | logprobs | row-0 probabilities
  if (s.topk > 0) {
    MI_HIP(hipMemcpyAsync(s.topk_ids, e->d_topk_ids, (size_t)n * s.topk * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipMemcpyAsync(s.topk_lp, e->d_topk_lp, (size_t)n * s.topk * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  MI_TRY(seam_record(e, t));
  MI_HIP(hipEventRecord(s.ev, st));
  *ticket = t;
  return MI_OK;
}

int mi_step_enqueue_mixed(mi_engine* e, mi_kv* kv, const int32_t* rows, const int32_t* lens, const int32_t* want, int n,
                          const int32_t* tokens, const mi_sample_params* sp, int64_t* ticket) {
  if (!e || !kv || !rows || !lens || !want || !tokens) return fail(MI_ERR_INVALID, "null argument");
  MI_TRY(check_params(sp));
  if (n < 1 || n > kv->B) return fail(MI_ERR_INVALID, "mi_step_enqueue_mixed: n must be in [1, batch of the kv]");
  size_t R = 0; int n_out = 0;
  for (int i = 0; i < n; ++i) {
    if (rows[i] < 0 || rows[i] >= kv->B) return fail(MI_ERR_INVALID, "mi_step_enqueue_mixed: row out of range");
    for (int j = 0; j < i; ++j) if (rows[j] == rows[i]) return fail(MI_ERR_INVALID, "mi_step_enqueue_mixed: duplicate row");
    if (lens[i] < 1) return fail(MI_ERR_INVALID, "mi_step_enqueue_mixed: every segment needs at least one token");
    R += (size_t)lens[i]; n_out += want[i] ? 1 : 0;
  }
  MI_TRY(check_call(e, kv, kv->B, 1));
  if (!ticket) return fail(MI_ERR_INVALID, "null ticket");
  if (sp && (sp->row_temperature != nullptr) != (sp->row_top_p != nullptr)) return fail(MI_ERR_INVALID, "row_temperature and row_top_p come together");
  MI_TRY(ensure_workspace(e, R, (size_t)std::max(n_out, 1), std::max(n, kv->B)));
  hipStream_t st = e->stream;
  for (size_t i = 0; i < R; ++i)
    if (tokens[i] < 0 || tokens[i] >= e->d.vocab_size) return fail(MI_ERR_INVALID, "token id out of range");
  MI_HIP(hipMemcpyAsync(e->d_tokens, tokens, R * sizeof(int32_t), hipMemcpyHostToDevice, st));
  MI_TRY(forward_mixed(e, kv, rows, lens, want, n, n_out));
  const int64_t t = e->next_ticket++;
  Slot& s = e->slots[t % NSLOT];
  s.B = n_out; s.topk = sp ? sp->top_logprobs : 0; s.ticket = t;
  if (n_out > 0) {
    MI_TRY(run_sample(e, n_out, sp));
    MI_HIP(hipMemcpyAsync(s.tokens, e->d_next, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipMemcpyAsync(s.logprob, e->d_logprob, n_out * sizeof(float), hipMemcpyDeviceToHost, st));
    MI_HIP(hipMemcpyAsync(s.prob0, e->d_prob0, n_out * sizeof(float), hipMemcpyDeviceToHost, st));
    if (s.topk > 0) {
      MI_HIP(hipMemcpyAsync(s.topk_ids, e->d_topk_ids, (size_t)n_out * s.topk * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      MI_HIP(hipMemcpyAsync(s.topk_lp, e->d_topk_lp, (size_t)n_out * s.topk * sizeof(float), hipMemcpyDeviceToHost, st));
    }
  }
  e->last_n = -1;                                // the device-resident token feed does not survive a mixed step
  MI_TRY(seam_record(e, t));
  MI_HIP(hipEventRecord(s.ev, st));
  *ticket = t;
  return MI_OK;
}

int mi_step_wait(mi_engine* e, int64_t ticket, int32_t* tokens_out, float* logprob_out, float* prob_row0_out,
                 int32_t* topk_ids, float* topk_logprobs) {
  if (!e) return fail(MI_ERR_INVALID, "null engine");
  if (ticket < 0 || ticket >= e->next_ticket) return fail(MI_ERR_INVALID, "unknown ticket");
  Slot& s = e->slots[ticket % NSLOT];
  if (s.ticket != ticket) return fail(MI_ERR_INVALID, "ticket expired (more than 4 steps in flight)");
  MI_HIP(hipSetDevice(e->device));
  MI_HIP(hipEventSynchronize(s.ev));
  if (e->h_seam_err != nullptr && e->h_seam_err[ticket % NSLOT] != 0) {
    e->h_seam_err[ticket % NSLOT] = 0;
    return seam_fail(e);
  }
  if (tokens_out) memcpy(tokens_out, s.tokens, s.B * sizeof(int32_t));
  if (logprob_out) memcpy(logprob_out, s.logprob, s.B * sizeof(float));
  if (prob_row0_out) memcpy(prob_row0_out, s.prob0, s.B * sizeof(float));
  if (topk_ids && s.topk > 0) memcpy(topk_ids, s.topk_ids, (size_t)s.B * s.topk * sizeof(int32_t));
  if (topk_logprobs && s.topk > 0) memcpy(topk_logprobs, s.topk_lp, (size_t)s.B * s.topk * sizeof(float));
  return MI_OK;
}

int mi_decode_sample(mi_engine* e, mi_kv* kv, const int32_t* tokens_in, int B, int L, const mi_sample_params* sp,
                     int32_t* tokens_out, float* logprob_out, float* prob_row0_out, int32_t* topk_ids,
                     float* topk_logprobs) {
  int64_t t = -1;
  MI_TRY(mi_step_enqueue(e, kv, tokens_in, B, L, sp, &t));
  return mi_step_wait(e, t, tokens_out, logprob_out, prob_row0_out, topk_ids, topk_logprobs);
}

// ---- measurement hooks ----------------------------------------------------------------------
int mi_profile_select(mi_engine* e, const char* name) {
  if (!e) return fail(MI_ERR_INVALID, "null engine");
  hipStreamSynchronize(e->stream);
  for (auto& p : e->prof_events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  e->prof_events.clear();
  e->prof_name = name ? name : "";
  return MI_OK;
}

int mi_profile_read(mi_engine* e, int64_t* n_launches, double* total_ms) {
  if (!e || !n_launches || !total_ms) return fail(MI_ERR_INVALID, "null argument");
  MI_HIP(hipStreamSynchronize(e->stream));
  double tot = 0.0;
  for (auto& p : e->prof_events) {
    float ms = 0.f;
    MI_HIP(hipEventElapsedTime(&ms, p.first, p.second));
    tot += ms;
    hipEventDestroy(p.first); hipEventDestroy(p.second);
  }
  *n_launches = (int64_t)e->prof_events.size();
  *total_ms = tot;
  e->prof_events.clear();
  return MI_OK;
}

int mi_engine_set_option(mi_engine* e, const char* key, int64_t value) {
  if (!e || !key) return fail(MI_ERR_INVALID, "null argument");
  const std::string k(key);
  if (k == "force_generic_gemv") { e->opt_force_v1 = value != 0; return MI_OK; }
  if (k == "fused_decode_attention") { e->opt_fused_attn = value != 0; return MI_OK; }
  if (k == "prefill_gemm") { e->opt_prefill_gemm = value != 0; return MI_OK; }
  if (k == "skinny_gemm") { e->opt_skinny_gemm = value != 0; return MI_OK; }
  if (k == "norm_handover") { e->opt_norm_handover = value != 0; return MI_OK; }
  if (k == "defer_norm") { e->opt_defer_norm = value != 0; return MI_OK; }
  if (k == "prefill_x_terms") { if (value != 2 && value != 3) return fail(MI_ERR_INVALID, "prefill_x_terms: 2 or 3"); e->opt_prefill_x_terms = (int)value; return MI_OK; }
  if (k == "short_prefill_skinny") { e->opt_short_prefill_skinny = value != 0; return MI_OK; }
  if (k == "decode_attention_mfma") { e->opt_attn_mfma = value != 0; return MI_OK; }
  if (k == "fused_gemv_pairs") { e->opt_fused_pairs = (int)value; return MI_OK; }
  if (k == "f16_hilo") { e->opt_f16_hilo = value != 0; return MI_OK; }
  if (k == "gate_up_interleave") { e->opt_gu8 = value != 0; return MI_OK; }
  if (k == "seam_spin_limit") {
    if (value < 0 || value > (int64_t)0x7fffffff) return fail(MI_ERR_INVALID, "seam_spin_limit out of range");
    e->seam_spin_limit = (unsigned)value; return MI_OK;
  }
  if (k == "tile_weights") {
    if (e->finalized) return fail(MI_ERR_INVALID, "tile_weights must be set before mi_engine_finalize");
    e->opt_tile_weights = value != 0; return MI_OK;
  }
  return fail(MI_ERR_NOTFOUND, "unknown option: " + k);
}

int mi_engine_sync(mi_engine* e) {
  if (!e) return fail(MI_ERR_INVALID, "null engine");
  MI_HIP(hipSetDevice(e->device));
  MI_HIP(hipStreamSynchronize(e->stream));
  return seam_check_sync(e);
}

}  // extern "C"
