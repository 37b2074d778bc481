// attn.hip -- q/k norm + RoPE + KV-cache append, and KV-cache attention.
//
// Replaces, for one transformer block (llama.py:93-141 / qwen3.py:63-113):
//   * qwen3 per-head q_norm / k_norm RMSNorm over head_dim        (qwen3.py:65-70)
//   * nn.RoPE, non-traditional half-split, per-row offsets          (llama.py:77-82,107-117)
//   * cache.update_and_fetch(keys, values)                          (base.py:66-85 / 119-140)
//   * mx.fast.scaled_dot_product_attention(q, k, v, scale, mask)    (llama.py:139-141)
// The additive causal mask of base.py:17-40 (-1e9 where query_pos < key_pos, left pads NOT
// masked) is applied by construction: query t of row b sees keys [0, offsets[b] + t].
//
// KV layout in HBM: [B][Hkv][capacity][D], one row of D elements per token, so the keys of one
// (sequence, kv-head) are one contiguous stream; a wave reads four rows per instruction
// (16 lanes x 16 B per 128-element bf16 row).  K/V go straight to VGPRs (read once, no reuse
// across waves); the GQA group (Hq/Hkv query heads) shares every K/V row read.
#include "kernels.h"

namespace mi {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T, int N>
__device__ __forceinline__ void load_row_piece(const T* p, float (&o)[N]) {
  constexpr int BYTES = N * sizeof(T);
  if constexpr (BYTES % 16 == 0) {
#pragma unroll
    for (int i = 0; i < BYTES / 16; ++i) {
      const u32x4 v = *((const u32x4*)p + i);
      const T* e = (const T*)&v;
#pragma unroll
      for (int j = 0; j < 16 / (int)sizeof(T); ++j) o[i * (16 / sizeof(T)) + j] = (float)e[j];
    }
  } else if constexpr (BYTES == 8) {
    const u32x2 v = *(const u32x2*)p;
    const T* e = (const T*)&v;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (float)e[j];
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (float)p[j];
  }
}

// ------------------------------------------------------------------------------------------
// rope_append: grid (Hq + 2*Hkv, B*L), block 64.  One wave per (token, head).
template <typename AT, typename KT, bool PAGED>
__global__ __launch_bounds__(64) void rope_append_kernel(RopeAppendCall c) {
  const AttnShape& s = c.s;
  const int head = blockIdx.x, row = blockIdx.y;
  const int b = row / s.L, t = row % s.L;
  const int lane = threadIdx.x;
  const int D = s.D, D2 = D / 2;
  const int kb = s.rows ? s.rows[b] : b;          // cache row of batch entry b (continuous batching: a subset of rows)
  const int pos = c.offsets[kb] + t;
  const int nq = s.Hq * D;
  const AT* src = (const AT*)c.qkv + (size_t)row * (nq + 2 * s.Hkv * D) + (size_t)head * D;
  const bool is_q = head < s.Hq, is_k = !is_q && head < s.Hq + s.Hkv;
  if (pos >= s.cap || pos >= c.max_pos) return;  // host checks capacity; never write out of bounds

  if (!is_q && !is_k) {  // values: plain append
    const int kh = head - s.Hq - s.Hkv;
    KT* dst = (KT*)c.vcache + kv_elem<PAGED>(s, kb, kh, pos);
    for (int i = lane; i < D; i += 64) dst[i] = (KT)(float)src[i];
    return;
  }
  const AT* nw = (const AT*)(is_q ? c.q_norm_w : c.k_norm_w);
  float rs = 1.f;
  if (nw != nullptr) {
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = (float)src[i]; ss = fmaf(v, v, ss); }
    ss = wave_sum(ss);
    rs = 1.0f / sqrtf(ss / (float)D + c.eps);
  }
  for (int i = lane; i < D2; i += 64) {
    float x1 = (float)src[i], x2 = (float)src[i + D2];
    if (nw != nullptr) {
      x1 = to_f32(store_act<AT>(to_f32(store_act<AT>(x1 * rs, s.rnd)) * (float)nw[i], s.rnd));
      x2 = to_f32(store_act<AT>(to_f32(store_act<AT>(x2 * rs, s.rnd)) * (float)nw[i + D2], s.rnd));
    }
    const float cs = c.cos_tab[(size_t)pos * D2 + i], sn = c.sin_tab[(size_t)pos * D2 + i];
    const float o1 = round_rt(x1 * cs - x2 * sn, s.rnd);
    const float o2 = round_rt(x1 * sn + x2 * cs, s.rnd);
    if (is_q) {
      AT* dst = (AT*)c.q_out + (size_t)row * nq + (size_t)head * D;
      dst[i] = (AT)o1; dst[i + D2] = (AT)o2;
    } else {
      const int kh = head - s.Hq;
      KT* dst = (KT*)c.kcache + kv_elem<PAGED>(s, kb, kh, pos);
      // the 16-bit rounding of the model dtype happens before the (possibly wider) cache store
      dst[i] = (KT)to_f32((AT)o1); dst[i + D2] = (KT)to_f32((AT)o2);
    }
  }
}

// The same for 16-bit activations / caches with head_dim 128 (the prefill of both target models): one 256-thread
// workgroup per token, a lane owns 8 + 8 elements (i .. i+8 and i+64 .. i+72) of one head, 16-byte loads and stores
// -- the one-wave-per-(token, head) form above is 393 k tiny workgroups of 2-byte accesses at 8 x 1024 tokens
// (104 us per layer).  Element-wise arithmetic as above; the q/k-norm sum runs over a lane's 16 elements and then over
// the head's 8 lanes.
template <typename AT, bool PAGED>
__global__ __launch_bounds__(256) void rope_append_rows_kernel(RopeAppendCall c) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const AttnShape& s = c.s;
  constexpr int D = 128, D2 = 64;
  const int row = blockIdx.x, b = row / s.L, t = row % s.L;
  const int kb = s.rows ? s.rows[b] : b;
  const int pos = c.offsets[kb] + t;
  if (pos >= s.cap || pos >= c.max_pos) return;
  const int nq = s.Hq * D, nh = s.Hq + 2 * s.Hkv;
  const AT* src_row = (const AT*)c.qkv + (size_t)row * (nq + 2 * s.Hkv * D);
  for (int task = threadIdx.x; task < nh * 8; task += 256) {     // (nh * 8) % 8 == 0: a head's 8 lanes stay together
    const int head = task >> 3, i0 = (task & 7) * 8;
    const AT* src = src_row + (size_t)head * D;
    u32x4 v1 = *(const u32x4*)(src + i0), v2 = *(const u32x4*)(src + i0 + D2);
    const bool is_q = head < s.Hq, is_k = !is_q && head < s.Hq + s.Hkv;
    if (!is_q && !is_k) {                                        // values: plain append
      AT* dst = (AT*)c.vcache + kv_elem<PAGED>(s, kb, head - s.Hq - s.Hkv, pos);
      *(u32x4*)(dst + i0) = v1; *(u32x4*)(dst + i0 + D2) = v2;
      continue;
    }
    AT* e1 = (AT*)&v1;
    AT* e2 = (AT*)&v2;
    const AT* nw = (const AT*)(is_q ? c.q_norm_w : c.k_norm_w);
    if (nw != nullptr) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float a = (float)e1[j], bb = (float)e2[j]; ss = fmaf(a, a, ss); ss = fmaf(bb, bb, ss); }
      ss = lane8_sum(ss);
      const float rs = 1.0f / sqrtf(ss / (float)D + c.eps);
      const u32x4 w1 = *(const u32x4*)(nw + i0), w2 = *(const u32x4*)(nw + i0 + D2);
      const AT* q1 = (const AT*)&w1;
      const AT* q2 = (const AT*)&w2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e1[j] = (AT)((float)(AT)((float)e1[j] * rs) * (float)q1[j]);
        e2[j] = (AT)((float)(AT)((float)e2[j] * rs) * (float)q2[j]);
      }
    }
    const f32x4* ct = (const f32x4*)(c.cos_tab + (size_t)pos * D2 + i0);
    const f32x4* st = (const f32x4*)(c.sin_tab + (size_t)pos * D2 + i0);
    const f32x4 c0 = ct[0], c1 = ct[1], s0 = st[0], s1 = st[1];
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x1 = (float)e1[j], x2 = (float)e2[j];
      e1[j] = (AT)(x1 * cs[j] - x2 * sn[j]);
      e2[j] = (AT)(x1 * sn[j] + x2 * cs[j]);
    }
    AT* dst = is_q ? (AT*)c.q_out + (size_t)row * nq + (size_t)head * D
                   : (AT*)c.kcache + kv_elem<PAGED>(s, kb, is_q ? 0 : head - s.Hq, pos);
    *(u32x4*)(dst + i0) = v1; *(u32x4*)(dst + i0 + D2) = v2;
  }
}

// ------------------------------------------------------------------------------------------
// attention: grid (nsplit, B*Hkv, L), block 256 (4 waves).  Wave w, lane group gq = lane>>4
// handles key positions s0 + 16*i + 4*w + gq; the 16 lanes of a group each own D/16 elements.
template <typename AT, typename KT, int D, int G, bool PAGED>
__global__ __launch_bounds__(256) void attn_kernel(AttnCall c) {
  constexpr int EPL = D / 16;
  const AttnShape& s = c.s;
  const int split = blockIdx.x, bh = blockIdx.y, t = blockIdx.z;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, gq = lane >> 4;
  const int row = b * s.L + t;
  const int kb = s.rows ? s.rows[b] : b;
  const int n_keys = c.offsets[kb] + t + 1;
  const int chunk = (n_keys + c.nsplit - 1) / c.nsplit;
  const int s0 = split * chunk, s1 = min(n_keys, s0 + chunk);

  __shared__ float st_m[4][G], st_l[4][G];
  __shared__ float st_o[4][G][D];

  float q[G][EPL];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const AT* qp = (const AT*)c.q + (size_t)row * s.Hq * D + (size_t)(kh * G + g) * D + li * EPL;
    load_row_piece<AT, EPL>(qp, q[g]);
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[g][e] *= c.scale;
  }
  float m[G], l[G], o[G][EPL];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[g][e] = 0.f;
  }
  const KT* kbase = (const KT*)c.kcache + li * EPL;
  const KT* vbase = (const KT*)c.vcache + li * EPL;

  constexpr int U = 2;
  for (int sb = s0 + 4 * wave + gq; sb < s1; sb += 16 * U) {
    float kk[U][EPL], vv[U][EPL];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sp = sb + 16 * u;
      ok[u] = sp < s1;
      const int spc = ok[u] ? sp : s0;
      const size_t ro = kv_elem<PAGED>(s, kb, kh, spc);
      load_row_piece<KT, EPL>(kbase + ro, kk[u]);
      load_row_piece<KT, EPL>(vbase + ro, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) d = fmaf(q[g][e], kk[u][e], d);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        d += __shfl_xor(d, 8, 64);
        if (ok[u]) {
          const float mn = fmaxf(m[g], d);
          const float corr = __expf(m[g] - mn);   // exp(-inf) = 0 on the first key
          const float pr = __expf(d - mn);
          l[g] = l[g] * corr + pr;
#pragma unroll
          for (int e = 0; e < EPL; ++e) o[g][e] = fmaf(pr, vv[u][e], o[g][e] * corr);
          m[g] = mn;
        }
      }
    }
  }

  // ---- merge the four lane groups of the wave (lanes li, li+16, li+32, li+48)
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(m[g], off, 64);
      const float lo = __shfl_xor(l[g], off, 64);
      const float mn = fmaxf(m[g], mo);
      const float ca = (m[g] == -INFINITY) ? 0.f : __expf(m[g] - mn);
      const float cb = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
      l[g] = l[g] * ca + lo * cb;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const float oo = __shfl_xor(o[g][e], off, 64);
        o[g][e] = o[g][e] * ca + oo * cb;
      }
      m[g] = mn;
    }
    if (gq == 0) {
      if (li == 0) { st_m[wave][g] = m[g]; st_l[wave][g] = l[g]; }
#pragma unroll
      for (int e = 0; e < EPL; ++e) st_o[wave][g][li * EPL + e] = o[g][e];
    }
  }
  __syncthreads();
  // ---- merge the four waves; thread -> (g, d)
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, d = idx % D;
    float mn = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mn = fmaxf(mn, st_m[w][g]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float cw = (st_m[w][g] == -INFINITY) ? 0.f : __expf(st_m[w][g] - mn);
      L = fmaf(st_l[w][g], cw, L);
      O = fmaf(st_o[w][g][d], cw, O);
    }
    const int h = kh * G + g;
    if (c.nsplit == 1) {
      ((AT*)c.out)[(size_t)row * s.Hq * D + (size_t)h * D + d] = store_act<AT>(O / L, s.rnd);
    } else {
      float* pp = c.partial + (((size_t)row * s.Hq + h) * c.nsplit + split) * (D + 2);
      pp[2 + d] = O;
      if (d == 0) { pp[0] = mn; pp[1] = L; }
    }
  }
}

// combine split partials: grid (B*L*Hq), block D
template <typename AT>
__global__ void attn_combine_kernel(const float* partial, AT* out, int nsplit, int D, int rnd) {
  const int rh = blockIdx.x, d = threadIdx.x;
  const float* pp = partial + (size_t)rh * nsplit * (D + 2);
  float mn = -INFINITY;
  for (int i = 0; i < nsplit; ++i) mn = fmaxf(mn, pp[i * (D + 2)]);
  float L = 0.f, O = 0.f;
  for (int i = 0; i < nsplit; ++i) {
    const float mi_ = pp[i * (D + 2)];
    const float cw = (mi_ == -INFINITY) ? 0.f : __expf(mi_ - mn);
    L = fmaf(pp[i * (D + 2) + 1], cw, L);
    O = fmaf(pp[i * (D + 2) + 2 + d], cw, O);
  }
  out[(size_t)rh * D + d] = store_act<AT>(O / L, rnd);
}

template <typename AT, typename KT, int D>
int launch_attn_g(const AttnCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  const dim3 grid(c.nsplit, s.B * s.Hkv, s.L), block(256);
  const int G = s.Hq / s.Hkv;
#define AK(GV) do { if (s.btab) hipLaunchKernelGGL((attn_kernel<AT, KT, D, GV, true>), grid, block, 0, st, c); \
                    else hipLaunchKernelGGL((attn_kernel<AT, KT, D, GV, false>), grid, block, 0, st, c); } while (0)
  switch (G) {
    case 1: AK(1); break;
    case 2: AK(2); break;
    case 4: AK(4); break;
    case 5: AK(5); break;
    case 8: AK(8); break;
    default: return fail(MI_ERR_UNSUPPORTED, "attention: Hq/Hkv must be 1, 2, 4, 5 or 8");
  }
#undef AK
  MI_HIP(hipGetLastError());
  if (c.nsplit > 1) {
    hipLaunchKernelGGL((attn_combine_kernel<AT>), dim3(s.B * s.L * s.Hq), dim3(D), 0, st, c.partial,
                       (AT*)c.out, c.nsplit, D, s.rnd);
    MI_HIP(hipGetLastError());
  }
  return MI_OK;
}

template <typename AT, typename KT>
int launch_attn_d(const AttnCall& c, hipStream_t st) {
  switch (c.s.D) {
    case 16: return launch_attn_g<AT, KT, 16>(c, st);
    case 32: return launch_attn_g<AT, KT, 32>(c, st);
    case 64: return launch_attn_g<AT, KT, 64>(c, st);
    case 128: return launch_attn_g<AT, KT, 128>(c, st);
  }
  return fail(MI_ERR_UNSUPPORTED, "attention: head_dim must be 16, 32, 64 or 128");
}


}  // namespace

int launch_rope_append(const RopeAppendCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  if (s.D % 2 != 0) return fail(MI_ERR_UNSUPPORTED, "rope: head_dim must be even");
  if (s.D == 128 && s.rnd == RND_NONE && s.L > 1 && s.kv == s.act && (s.act == MI_BF16 || s.act == MI_F16)) {
    const dim3 grid_r(s.B * s.L), block_r(256);
    if (s.act == MI_BF16) {
      if (s.btab) hipLaunchKernelGGL((rope_append_rows_kernel<bf16, true>), grid_r, block_r, 0, st, c);
      else hipLaunchKernelGGL((rope_append_rows_kernel<bf16, false>), grid_r, block_r, 0, st, c);
    } else {
      if (s.btab) hipLaunchKernelGGL((rope_append_rows_kernel<f16, true>), grid_r, block_r, 0, st, c);
      else hipLaunchKernelGGL((rope_append_rows_kernel<f16, false>), grid_r, block_r, 0, st, c);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  const dim3 grid(s.Hq + 2 * s.Hkv, s.B * s.L), block(64);
#define RA(AT, KT) do { if (s.btab) hipLaunchKernelGGL((rope_append_kernel<AT, KT, true>), grid, block, 0, st, c); \
                        else hipLaunchKernelGGL((rope_append_kernel<AT, KT, false>), grid, block, 0, st, c); } while (0)
  if (s.act == MI_F32 && s.kv == MI_F32) RA(float, float);
  else if (s.act == MI_BF16 && s.kv == MI_BF16) RA(bf16, bf16);
  else if (s.act == MI_F16 && s.kv == MI_F16) RA(f16, f16);
  else return fail(MI_ERR_UNSUPPORTED, "rope_append: activation / KV dtype combination not supported");
#undef RA
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_attention(const AttnCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  if (s.Hq % s.Hkv != 0) return fail(MI_ERR_INVALID, "attention: Hq must be a multiple of Hkv");
  if (c.nsplit < 1 || (c.nsplit > 1 && (s.L != 1 || c.partial == nullptr)))
    return fail(MI_ERR_INVALID, "attention: bad split configuration");
  if (s.L > 1 && c.variant != 1 && c.nsplit == 1 && attention_prefill_supported(s)) return launch_attention_prefill(c, st);
  if (s.act == MI_F32 && s.kv == MI_F32) return launch_attn_d<float, float>(c, st);
  if (s.act == MI_BF16 && s.kv == MI_BF16) return launch_attn_d<bf16, bf16>(c, st);
  if (s.act == MI_F16 && s.kv == MI_F16) return launch_attn_d<f16, f16>(c, st);
  return fail(MI_ERR_UNSUPPORTED, "attention: activation / KV dtype combination not supported");
}

}  // namespace mi
