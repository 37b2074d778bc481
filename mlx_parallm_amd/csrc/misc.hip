// misc.hip -- embedding gather, sampler, RoPE tables, small utilities.
#include <type_traits>

#include "kernels.h"

namespace mi {

namespace {

// ------------------------------------------------------------------------------------------
// nn.Embedding / nn.QuantizedEmbedding row gather (llama.py:212, qwen3.py:166).
// Quantised rows are dequantised as round_T(scale*q + bias).
template <typename AT, int WK>
__global__ __launch_bounds__(256) void embed_kernel(LinearW W, EmbedCall c) {
  const int row = blockIdx.x;
  int tok = c.tokens[row];
  tok = min(max(tok, 0), W.N - 1);
  AT* out = (AT*)c.out + (size_t)row * W.K;
  if constexpr (WK == WK_BF16 || WK == WK_F16) {
    // 16-bit tables: 16-byte pieces (8 elements) per thread instead of one element per trip (10 -> ~4 us at the head of every
    // decode step); same values, store_act is applied element by element as below
    if (W.K % 8 == 0) {
      using WT = typename std::conditional<WK == WK_BF16, bf16, f16>::type;
      for (int p8 = threadIdx.x; p8 < W.K / 8; p8 += 256) {
        const char* src = W.layout ? (const char*)W.w + tiled_piece_dense16(tok, p8 * 8, W.K)
                                   : (const char*)W.w + ((size_t)tok * W.K + (size_t)p8 * 8) * 2;
        const uint4 raw = *(const uint4*)src;
        const WT* e = (const WT*)&raw;
        AT o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = store_act<AT>((float)e[i], c.rnd);
        if constexpr (sizeof(AT) == 2) {
          *(uint4*)(out + (size_t)p8 * 8) = *(const uint4*)o;
        } else {
          *(uint4*)(out + (size_t)p8 * 8) = *(const uint4*)o;
          *(uint4*)(out + (size_t)p8 * 8 + 4) = *(const uint4*)(o + 4);
        }
      }
      return;
    }
  }
  for (int k = threadIdx.x; k < W.K; k += 256) {
    float v;
    const int k8 = k & ~7;
    if constexpr (WK == WK_F32) v = ((const float*)W.w)[(size_t)tok * W.K + k];
    else if constexpr (WK == WK_BF16)
      v = W.layout ? (float)((const bf16*)((const char*)W.w + tiled_piece_dense16(tok, k8, W.K)))[k & 7]
                   : (float)((const bf16*)W.w)[(size_t)tok * W.K + k];
    else if constexpr (WK == WK_F16)
      v = W.layout ? (float)((const f16*)((const char*)W.w + tiled_piece_dense16(tok, k8, W.K)))[k & 7]
                   : (float)((const f16*)W.w)[(size_t)tok * W.K + k];
    else {
      constexpr int BITS = (WK >= WK_Q8_F32) ? 8 : 4;
      constexpr int SDT = (WK - 3) % 3;
      constexpr int PER = 32 / BITS;
      const size_t gi = (size_t)tok * (W.K / W.group) + k / W.group;
      float s, b;
      uint32_t word;
      if (BITS == 4 && SDT != 0 && W.layout) {     // tile-major int4 (repack.hip)
        const char* blk = (const char*)W.w + tiled_block_q4(tok, k, W.K);
        const int so = tiled_q4_scale_off(tok, k);
        if constexpr (SDT == 1) { s = (float)*(const bf16*)(blk + so); b = (float)*(const bf16*)(blk + so + 64); }
        else { s = (float)*(const f16*)(blk + so); b = (float)*(const f16*)(blk + so + 64); }
        word = *(const uint32_t*)(blk + tiled_q4_code_off(tok, k8));
      } else if (BITS == 8 && SDT != 0 && W.layout) {   // tile-major int8
        const char* blk = (const char*)W.w + tiled_block_q8(tok, k, W.K);
        const int so = tiled_q8_scale_off(tok);
        if constexpr (SDT == 1) { s = (float)*(const bf16*)(blk + so); b = (float)*(const bf16*)(blk + so + 32); }
        else { s = (float)*(const f16*)(blk + so); b = (float)*(const f16*)(blk + so + 32); }
        word = *(const uint32_t*)(blk + tiled_q8_code_off(tok, k & ~3));
      } else {
      if constexpr (SDT == 0) { s = ((const float*)W.scales)[gi]; b = ((const float*)W.biases)[gi]; }
      else if constexpr (SDT == 1) { s = (float)((const bf16*)W.scales)[gi]; b = (float)((const bf16*)W.biases)[gi]; }
      else { s = (float)((const f16*)W.scales)[gi]; b = (float)((const f16*)W.biases)[gi]; }
      word = ((const uint32_t*)W.w)[(size_t)tok * (W.K / PER) + k / PER];
      }
      const float q = (float)((word >> (BITS * (k % PER))) & ((1u << BITS) - 1u));
      v = mul_add_unfused(q, s, b);
    }
    out[k] = store_act<AT>(v, c.rnd);
  }
}

template <typename AT>
int launch_embed_wk(const LinearW& W, const EmbedCall& c, hipStream_t st) {
  const dim3 grid(c.rows), block(256);
#define EK(WKV) hipLaunchKernelGGL((embed_kernel<AT, WKV>), grid, block, 0, st, W, c); break
  switch (W.wk) {
    case WK_F32: EK(WK_F32);
    case WK_BF16: EK(WK_BF16);
    case WK_F16: EK(WK_F16);
    case WK_Q4_F32: EK(WK_Q4_F32);
    case WK_Q4_BF16: EK(WK_Q4_BF16);
    case WK_Q4_F16: EK(WK_Q4_F16);
    case WK_Q8_F32: EK(WK_Q8_F32);
    case WK_Q8_BF16: EK(WK_Q8_BF16);
    case WK_Q8_F16: EK(WK_Q8_F16);
    default: return fail(MI_ERR_UNSUPPORTED, "embed: bad weight kind");
  }
#undef EK
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// ------------------------------------------------------------------------------------------
// Sampler: the `sample` closure of generate_step (utils.py:345-364) + top_p_sampling
// (sample_utils.py:3-38).  One 1024-thread workgroup per row; logits stay in L2.
//   greedy: argmax, lowest index among ties (mx.argmax).
//   temp>0: candidates in descending-probability order (ties: ascending id); nucleus keeps the
//           prefix whose INCLUSIVE cumulative probability is <= top_p (empty -> top-1, quirk Q5);
//           the draw is an inverse-CDF pick with one uniform per row.
// Cumulative masses are integers (probability * 2^40 summed with integer atomics), so the
// result does not depend on the order in which threads add.
constexpr int ST = 1024;

// f(value, index) over one row of logits.  A pass is latency-bound if every thread walks it one dependent 4-byte load
// at a time (measured: ~9 us per pass over 32000 logits); 16-byte loads, two in flight per thread.  (Four in flight:
// measured in round 4, no change -- 55.7 us per top-p draw at V = 32000 either way.  What bounds the histogram passes is
// the same-address serialisation of their LDS atomics: a row's logits share a handful of exponents.)
template <class F>
__device__ __forceinline__ void for_row(const float* lg, int V, F f) {
  if ((V & 3) == 0 && (((uintptr_t)lg) & 15) == 0) {
    const float4* p4 = (const float4*)lg;
    const int n4 = V >> 2;
    int i = threadIdx.x;
    for (; i + ST < n4; i += 2 * ST) {
      const float4 a = p4[i], b = p4[i + ST];
      const int j = 4 * i, k = 4 * (i + ST);
      f(a.x, j); f(a.y, j + 1); f(a.z, j + 2); f(a.w, j + 3);
      f(b.x, k); f(b.y, k + 1); f(b.z, k + 2); f(b.w, k + 3);
    }
    if (i < n4) {
      const float4 a = p4[i];
      const int j = 4 * i;
      f(a.x, j); f(a.y, j + 1); f(a.z, j + 2); f(a.w, j + 3);
    }
  } else {
    for (int i = threadIdx.x; i < V; i += ST) f(lg[i], i);
  }
}

__device__ __forceinline__ uint32_t order_key(float f) {  // larger float -> larger key
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax am_better(ArgMax a, ArgMax b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

__device__ ArgMax block_argmax(ArgMax a, ArgMax* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b{__shfl_xor(a.v, o, 64), __shfl_xor(a.i, o, 64)};
    a = am_better(a, b);
  }
  __syncthreads();
  if (lane == 0) sh[wave] = a;
  __syncthreads();
  ArgMax r = sh[0];
  for (int w = 1; w < ST / 64; ++w) r = am_better(r, sh[w]);
  return r;
}

__device__ float block_sum(float v, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < ST / 64; ++w) r += sh[w];
  return r;
}

__device__ __forceinline__ unsigned long long mass_fx(float l, float mx, float inv_t) {
  // exp((l - max)/T) in (0, 1] as 2^40 fixed point
  return (unsigned long long)(__expf((l - mx) * inv_t) * 1099511627776.0f);
}

// Philox-4x32-10 (key = seed, counter = (step, row)) -> one uniform in [0,1)
__device__ float philox_uniform(uint64_t seed, uint64_t step, uint32_t row) {
  uint32_t c0 = (uint32_t)step, c1 = (uint32_t)(step >> 32), c2 = row, c3 = 0x9E3779B9u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// Find the descending-order prefix whose cumulative mass is <= target.
// Returns (through sh_*) the key of the first element NOT wholly inside the prefix (`kstar`),
// how many elements with key == kstar are inside the prefix (`ntie`), and the prefix mass.
// `limit_key/limit_tie`: only elements in a previously selected prefix are considered
// (key > limit_key, or key == limit_key and tie-rank < limit_tie); pass limit_key = 0,
// limit_tie = INT_MAX for "all".
struct Prefix { uint32_t kstar; int ntie; unsigned long long mass; int count; };

// Level-0 histogram (top 8 key bits) of the whole row, built once per row and shared by every descent.  The logits of
// one row share a few exponents, i.e. a few level-0 bins: the adds are spread over 8 private copies (by lane) to cut
// the same-address serialisation of the LDS atomics, then folded.
struct Hist0 { unsigned long long m[256]; int c[256]; };

__device__ void build_hist0(const float* lg, int V, float mx, float inv_t, unsigned long long (*pm)[8], int (*pc)[8],
                            Hist0* h0) {
  // [bin][copy]: the 8 copies of a bin are neighbours, i.e. on different LDS banks ([copy][bin] would put them 2 KiB
  // apart -- all on one bank)
  for (int i = threadIdx.x; i < 8 * 256; i += ST) { pm[i >> 3][i & 7] = 0; pc[i >> 3][i & 7] = 0; }
  __syncthreads();
  const int cp = threadIdx.x & 7;
  for_row(lg, V, [&](float l, int) {
    const int bin = order_key(l) >> 24;
    atomicAdd(&pm[bin][cp], mass_fx(l, mx, inv_t));
    atomicAdd(&pc[bin][cp], 1);
  });
  __syncthreads();
  if (threadIdx.x < 256) {
    unsigned long long m = 0; int n = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { m += pm[threadIdx.x][k]; n += pc[threadIdx.x][k]; }
    h0->m[threadIdx.x] = m; h0->c[threadIdx.x] = n;
  }
  __syncthreads();
}

// Which bin holds the element at which the descending cumulative mass (starting from `above`) first exceeds
// `target`?  One wave scans the 256 bins from 255 down (4 bins per lane + a wave scan: two block barriers instead of
// a block-wide scan); returns through out_sh: mass / count of everything above that bin, and the bin (-1: all fits).
__device__ void select_bin(const unsigned long long* hist_m, const int* hist_c, unsigned long long above, int above_cnt,
                           unsigned long long target, unsigned long long*, int*, int*, Prefix* out_sh) {
  __syncthreads();
  if (threadIdx.x < 64) {
    const int L = threadIdx.x;
    unsigned long long m[4], tm = 0; int c[4], tc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[j] = hist_m[255 - (4 * L + j)]; c[j] = hist_c[255 - (4 * L + j)]; tm += m[j]; tc += c[j]; }
    unsigned long long im = tm; int ic = tc;                       // inclusive scan over lanes = descending bins
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long om = __shfl_up(im, off, 64);
      const int oc = __shfl_up(ic, off, 64);
      if (L >= off) { im += om; ic += oc; }
    }
    unsigned long long cum = above + im - tm; int cnt = above_cnt + ic - tc;
    int pos = 256; unsigned long long pm = 0; int pc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (pos == 256) {
        if (c[j] > 0 && cum + m[j] > target) { pos = 4 * L + j; pm = cum; pc = cnt; }
        else { cum += m[j]; cnt += c[j]; }
      }
    }
    int best = pos;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
    if (best == 256) { if (L == 63) { out_sh->mass = cum; out_sh->count = cnt; out_sh->ntie = -1; } }
    else if (pos == best) { out_sh->mass = pm; out_sh->count = pc; out_sh->ntie = 255 - best; }
  }
  __syncthreads();
}

// Find the descending-order prefix whose cumulative mass is <= target (radix descent over the order key, 8 bits per
// level; level 0 comes from the cached histogram, the lower levels touch only the elements under the chosen prefix).
__device__ Prefix find_prefix(const float* lg, int V, float mx, float inv_t, unsigned long long target,
                              unsigned long long* hist_m, int* hist_c, Prefix* out_sh, const Hist0* h0,
                              unsigned long long* sc_m, int* sc_c, int* sel) {
  uint32_t prefix_bits = 0;   // fixed high bits of kstar so far
  uint32_t prefix_mask = 0;
  unsigned long long above = 0;  // mass of keys strictly greater than the current candidate bin
  int above_cnt = 0;
  for (int level = 0; level < 4; ++level) {
    const int shift = 24 - 8 * level;
    if (level == 0) {
      select_bin(h0->m, h0->c, above, above_cnt, target, sc_m, sc_c, sel, out_sh);
    } else {
      __syncthreads();
      for (int i = threadIdx.x; i < 256; i += ST) { hist_m[i] = 0; hist_c[i] = 0; }
      __syncthreads();
      for_row(lg, V, [&](float l, int) {
        const uint32_t k = order_key(l);
        if ((k & prefix_mask) == prefix_bits) {
          const int bin = (k >> shift) & 255;
          atomicAdd(&hist_m[bin], mass_fx(l, mx, inv_t));
          atomicAdd(&hist_c[bin], 1);
        }
      });
      __syncthreads();
      select_bin(hist_m, hist_c, above, above_cnt, target, sc_m, sc_c, sel, out_sh);
    }
    above = out_sh->mass;
    above_cnt = out_sh->count;
    const int bin = out_sh->ntie;
    if (bin < 0) {
      __syncthreads();
      Prefix r{0u, 0, above, above_cnt};
      return r;  // the whole (sub)set is inside the prefix
    }
    prefix_bits |= (uint32_t)bin << shift;
    prefix_mask |= 255u << shift;
  }
  // kstar = prefix_bits: all elements with this exact key have the same mass; count how many fit
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long each = mass_fx(__uint_as_float((prefix_bits & 0x80000000u) ? (prefix_bits & 0x7fffffffu) : ~prefix_bits), mx, inv_t);
    unsigned long long cum = above;
    int n = 0;
    while (each > 0 && cum + each <= target) { cum += each; ++n; }
    out_sh->mass = cum;
    out_sh->ntie = n;
    out_sh->count = above_cnt + n;
  }
  __syncthreads();
  Prefix r{prefix_bits, out_sh->ntie, out_sh->mass, out_sh->count};
  __syncthreads();
  return r;
}

// index of the (rank)-th smallest token id among those whose key == kstar.
// 16-bit logits make big tie groups (hundreds of ids share one value), so rank can be large: ONE counting pass
// in index order (chunks of 4 * ST ids, thread t of chunk j owns ids 4 * (ST * j + t) .. + 3; per-(chunk, wave)
// tie counts via ballots), a scan of that small table, and the owning wave finds the id among its 256.
constexpr int TIE_TBL = 1024;       // (chunk, wave) entries: V <= 4 * ST * (TIE_TBL / 16) = 262144 ids
__device__ int nth_tie(const float* lg, int V, uint32_t kstar, int rank, int* sh_i, int* tbl, int* found) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (V + 4 * ST - 1) / (4 * ST);
  const bool vec = (V & 3) == 0 && (((uintptr_t)lg) & 15) == 0 && nch * (ST / 64) <= TIE_TBL;
  if (vec) {
    auto ties_of = [&](int j, bool (&is)[4]) {
      const int base = 4 * (ST * j + (int)threadIdx.x);
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (base < V) v = *(const float4*)(lg + base);
      is[0] = base < V && order_key(v.x) == kstar; is[1] = base < V && order_key(v.y) == kstar;
      is[2] = base < V && order_key(v.z) == kstar; is[3] = base < V && order_key(v.w) == kstar;
    };
    for (int j = 0; j < nch; ++j) {
      bool is[4];
      ties_of(j, is);
      int n = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) n += __popcll(__ballot(is[e]));
      if (lane == 0) tbl[j * (ST / 64) + wave] = n;
    }
    if (threadIdx.x == 0) *found = -1;
    __syncthreads();
    if (threadIdx.x == 0) {                       // (chunk, wave) that holds the rank-th tie, and the rank inside it
      int left = rank, at = -1;
      for (int i = 0; i < nch * (ST / 64); ++i) {
        const int n = tbl[i];
        if (left < n) { at = i; break; }
        left -= n;
      }
      sh_i[0] = at; sh_i[1] = left;
    }
    __syncthreads();
    const int at = sh_i[0], left = sh_i[1];
    if (at >= 0 && wave == at % (ST / 64)) {
      bool is[4];
      ties_of(at / (ST / 64), is);
      const unsigned long long lt = (1ull << lane) - 1ull;
      int before = 0, own = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { before += __popcll(__ballot(is[e]) & lt); own += is[e] ? 1 : 0; }
      if (left >= before && left < before + own) {
        int k = left - before;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (is[e]) { if (k == 0) *found = 4 * (ST * (at / (ST / 64)) + (int)threadIdx.x) + e; --k; }
      }
    }
    __syncthreads();
    const int r = *found;
    __syncthreads();
    return r;                                     // -1: fewer than rank + 1 ties (the caller falls back to the arg-max)
  }
  int last = -1;
  for (int r = 0; r <= rank; ++r) {
    int best = 0x7fffffff;
    for_row(lg, V, [&](float l, int i) {
      if (i > last && i < best && order_key(l) == kstar) best = i;
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
    __syncthreads();
    if (lane == 0) sh_i[wave] = best;
    __syncthreads();
    best = sh_i[0];
    for (int w = 1; w < ST / 64; ++w) best = min(best, sh_i[w]);
    last = best;
    __syncthreads();
  }
  return last;
}

__global__ __launch_bounds__(ST) void sample_kernel(SampleCall c) {
  __shared__ ArgMax sh_am[ST / 64];
  __shared__ float sh_f[ST / 64];
  __shared__ int sh_i[ST / 64];
  __shared__ unsigned long long hist_m[256];
  __shared__ int hist_c[256];
  __shared__ Prefix sh_p;
  __shared__ unsigned long long priv_m[256][8], sc_m[256];
  __shared__ int priv_c[256][8], sc_c[256], sel_sh;
  __shared__ Hist0 h0;
  __shared__ int tie_tbl[TIE_TBL];
  const int b = blockIdx.x, V = c.V;
  float* lg = c.logits + (size_t)b * V;
  // per-row sampling parameters (continuous batching: every request keeps its own) or the call's scalars
  const float temperature = c.row_temp ? c.row_temp[b] : c.temperature;
  const float top_p = c.row_top_p ? c.row_top_p[b] : c.top_p;

  if (c.n_bias > 0) {  // logits[:, indices] += values (utils.py:346-349)
    for (int i = threadIdx.x; i < c.n_bias; i += ST) {
      const int id = c.bias_ids[i];
      if (id >= 0 && id < V) lg[id] = round_rt(lg[id] + c.bias_vals[i], c.rnd);
    }
    __syncthreads();
  }
  // ---- max / argmax, log-sum-exp
  ArgMax a{-INFINITY, 0x7fffffff};
  for_row(lg, V, [&](float l, int i) { a = am_better(a, ArgMax{l, i}); });
  a = block_argmax(a, sh_am);
  const float mx = a.v;
  float se = 0.f;
  for_row(lg, V, [&](float l, int) { se += __expf(l - mx); });
  se = block_sum(se, sh_f);
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0 && c.row_stats) { c.row_stats[2 * b] = mx; c.row_stats[2 * b + 1] = lse; }

  int token = a.i;
  const int forced = c.forced ? c.forced[b] : 0;
  if (c.forced) {
    if (forced >= 0 && forced < V) token = forced;
  } else if (temperature != 0.f) {
    const float inv_t = 1.0f / temperature;
    // total mass Z (integer): target = +inf prefix
    // total mass Z (integer) = sum of the level-0 histogram, which every descent below reuses
    build_hist0(lg, V, mx, inv_t, priv_m, priv_c, &h0);
    select_bin(h0.m, h0.c, 0ull, 0, ~0ull, sc_m, sc_c, &sel_sh, &sh_p);       // target = +inf: everything fits
    Prefix all{0u, 0, sh_p.mass, sh_p.count};
    const unsigned long long Z = all.mass;
    Prefix kept = all;
    uint32_t keep_key = 0; int keep_tie = 0x7fffffff;
    if (top_p > 0.f && top_p < 1.f) {
      const unsigned long long tgt = (unsigned long long)((double)top_p * (double)Z);
      kept = find_prefix(lg, V, mx, inv_t, tgt, hist_m, hist_c, &sh_p, &h0, sc_m, sc_c, &sel_sh);
      keep_key = kept.kstar; keep_tie = kept.ntie;
      if (kept.count == 0) {  // top token alone exceeds top_p (reference: 0/0); keep top-1
        kept.count = 1; kept.mass = mass_fx(mx, mx, inv_t);
        keep_key = order_key(mx); keep_tie = 1;
      }
    }
    const float u = c.uniforms ? c.uniforms[b] : philox_uniform(c.seed, c.step, (uint32_t)b);
    // first candidate whose inclusive cumulative mass exceeds u * Z_kept
    unsigned long long y = (unsigned long long)((double)u * (double)kept.mass);
    if (y >= kept.mass) y = kept.mass - 1;
    Prefix pk = find_prefix(lg, V, mx, inv_t, y, hist_m, hist_c, &sh_p, &h0, sc_m, sc_c, &sel_sh);
    // pk.count candidates lie strictly before the pick; the pick is the next one in order
    int rank_in_key = pk.ntie;
    uint32_t key = pk.kstar;
    if (pk.count >= kept.count) {  // numerical corner: clamp to the last kept candidate
      key = keep_key; rank_in_key = max(keep_tie - 1, 0);
      if (keep_tie == 0x7fffffff) { key = pk.kstar; rank_in_key = max(pk.ntie - 1, 0); }
    }
    token = nth_tie(lg, V, key, rank_in_key, sh_i, tie_tbl, &sel_sh);
    if (token < 0 || token >= V) token = a.i;
  }
  // logprobs are reported as (lg - mx) * lp_scale - lp_lse: the plain log-softmax, or the one of logits / T
  float lp_scale = 1.0f, lp_lse = lse - mx;
  if (c.lp_temp && temperature > 0.f) {
    lp_scale = 1.0f / temperature;
    float st = se;                               // T = 1: the sum of the pass above, term for term (no second walk of the row)
    if (lp_scale != 1.0f) {                      // (uniform)
      st = 0.f;
      for_row(lg, V, [&](float l, int) { st += __expf((l - mx) * lp_scale); });
      st = block_sum(st, sh_f);
    }
    lp_lse = __logf(st);
  }
  // a row without a single comparable logit (all NaN / -inf: a broken checkpoint, an overflow upstream) has no
  // arg-max; it yields token 0 rather than an id that the next kernels would use as an address
  if (token < 0 || token >= V) token = 0;
  if (threadIdx.x == 0) {
    c.tokens_out[b] = token;
    if (c.logprob_out) c.logprob_out[b] = (c.forced && forced < 0) ? 0.f : (lg[token] - mx) * lp_scale - lp_lse;
  }
  // ---- top-k logprobs: k rounds of arg-max with exclusion of already emitted ids
  if (c.top_logprobs > 0) {
    float prev_v = INFINITY; int prev_i = -1;
    for (int r = 0; r < c.top_logprobs; ++r) {
      ArgMax t{-INFINITY, 0x7fffffff};
      for_row(lg, V, [&](float v, int i) {
        if (v < prev_v || (v == prev_v && i > prev_i)) t = am_better(t, ArgMax{v, i});
      });
      t = block_argmax(t, sh_am);
      if (threadIdx.x == 0) {
        c.topk_ids[(size_t)b * c.top_logprobs + r] = t.i < V ? t.i : 0;
        c.topk_logprobs[(size_t)b * c.top_logprobs + r] = (t.v - mx) * lp_scale - lp_lse;
      }
      prev_v = t.v; prev_i = t.i;
    }
  }
}

// probs = softmax(logits)[0, tokens] (utils.py:363, row-0 quirk Q6)
__global__ void prob_row0_kernel(const float* logits, const float* row_stats, const int32_t* tokens, float* out, int B) {
  const int b = threadIdx.x;
  if (b < B) out[b] = __expf(logits[tokens[b]] - row_stats[1]);
}

__global__ void advance_offsets_kernel(int32_t* offsets, const int32_t* rows, int B, int L) {
  const int i = threadIdx.x;
  if (i < B) offsets[rows ? rows[i] : i] += L;
}

__global__ void rope_tables_kernel(float* cos_tab, float* sin_tab, int max_pos, int D2, double base, double scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= max_pos * D2) return;
  const int pos = idx / D2, i = idx % D2;
  const double inv = pow(base, -2.0 * (double)i / (double)(2 * D2));
  const double ang = (double)pos * scale * inv;
  cos_tab[idx] = (float)cos(ang);
  sin_tab[idx] = (float)sin(ang);
}

template <typename S, typename T>
__global__ void convert_kernel(const S* src, T* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (T)(float)src[i];
}

}  // namespace

int launch_embed(const LinearW& W, const EmbedCall& c, hipStream_t st) {
  switch (c.act) {
    case MI_F32: return launch_embed_wk<float>(W, c, st);
    case MI_BF16: return launch_embed_wk<bf16>(W, c, st);
    case MI_F16: return launch_embed_wk<f16>(W, c, st);
  }
  return fail(MI_ERR_INVALID, "embed: bad activation dtype");
}

int launch_sample(const SampleCall& c, hipStream_t st) {
  if (c.top_logprobs < 0 || c.top_logprobs > MI_MAX_TOP_LOGPROBS) return fail(MI_ERR_INVALID, "sample: top_logprobs out of range");
  hipLaunchKernelGGL(sample_kernel, dim3(c.B), dim3(ST), 0, st, c);
  MI_HIP(hipGetLastError());
  if (c.prob_row0_out) {
    hipLaunchKernelGGL(prob_row0_kernel, dim3(1), dim3(64 * ((c.B + 63) / 64)), 0, st, c.logits, c.row_stats,
                       c.tokens_out, c.prob_row0_out, c.B);
    MI_HIP(hipGetLastError());
  }
  return MI_OK;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const u128* src, size_t row_vecs, const int32_t* idx, u128* dst) {
  const u128* s = src + (size_t)idx[blockIdx.x] * row_vecs;
  u128* d = dst + (size_t)blockIdx.x * row_vecs;
  for (size_t i = threadIdx.x; i < row_vecs; i += 256) d[i] = s[i];
}

int launch_gather_rows(const void* src, size_t row_bytes, const int32_t* idx, int n, void* dst, hipStream_t st) {
  if (row_bytes % 16 != 0) return fail(MI_ERR_UNSUPPORTED, "gather_rows: rows must be multiples of 16 bytes");
  if (n <= 0) return MI_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, st, (const u128*)src, row_bytes / 16, idx, (u128*)dst);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_advance_offsets(int32_t* offsets, const int32_t* rows, int B, int L, hipStream_t st) {
  hipLaunchKernelGGL(advance_offsets_kernel, dim3(1), dim3(64 * ((B + 63) / 64)), 0, st, offsets, rows, B, L);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_rope_tables(float* cos_tab, float* sin_tab, int max_pos, int D, float base, float scale, hipStream_t st) {
  const int n = max_pos * (D / 2);
  hipLaunchKernelGGL(rope_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cos_tab, sin_tab, max_pos, D / 2,
                     (double)base, (double)scale);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_convert(const void* src, int sdt, void* dst, int ddt, size_t n, hipStream_t st) {
  const dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 4096)), block(256);
#define CV(S, T) hipLaunchKernelGGL((convert_kernel<S, T>), grid, block, 0, st, (const S*)src, (T*)dst, n)
  if (sdt == MI_F32 && ddt == MI_F32) CV(float, float);
  else if (sdt == MI_BF16 && ddt == MI_F32) CV(bf16, float);
  else if (sdt == MI_F16 && ddt == MI_F32) CV(f16, float);
  else if (sdt == MI_F32 && ddt == MI_BF16) CV(float, bf16);
  else if (sdt == MI_F32 && ddt == MI_F16) CV(float, f16);
  else if (sdt == MI_BF16 && ddt == MI_BF16) CV(bf16, bf16);
  else if (sdt == MI_F16 && ddt == MI_F16) CV(f16, f16);
  else return fail(MI_ERR_UNSUPPORTED, "convert: dtype pair not supported");
#undef CV
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
