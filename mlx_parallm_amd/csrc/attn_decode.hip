// attn_decode.hip -- fused decode step of one attention block (L == 1), ONE launch:
//   q/k RMSNorm (qwen3.py:65-70) + RoPE (llama.py:107-117) + KV append (base.py:66-85 / 119-140)
//   + scaled_dot_product_attention over the cache (llama.py:139-141) + split-KV combine.
//
// Grid (nsplit, B*Hkv), 512 threads = 8 waves.  A lane group of 16 lanes (one DPP row) owns one
// key at a time and D/16 elements of it: a wave reads four consecutive KV rows (1 KiB for
// bf16 D = 128) per instruction, and every (wave, lane group) issues U K-rows and U V-rows before
// touching any of them, so a 256-key split is entirely in flight at once.
//   * the GQA group (G = Hq/Hkv query heads) shares every K/V row;
//   * q.k on 16-bit caches uses v_dot2c_f32_{bf16,f16}; the 16-lane reductions are DPP row
//     rotations (no LDS traffic); softmax runs in the exp2 domain, one rescale per U-key block;
//   * the RoPE partner (i <-> i + D/2) is the lane 8 positions away in the row;
//   * the split that owns the new position appends K/V and takes the new key from registers;
//   * splits publish (max, sum, O) partials and take a ticket; the last arriver of a
//     (sequence, kv-head) combines them (agent-scope release -> ticket -> acquire; nobody waits).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gemv_phase.h"

namespace mi {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float row16_ror8(float v) { return __builtin_amdgcn_update_dpp(0.f, v, 0x128, 0xf, 0xf, false); }

template <typename T>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
  else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
}

// one lane's piece of a row: NW32 dwords of raw bits
template <int NW32>
__device__ __forceinline__ void load_raw(const void* p, uint32_t (&o)[NW32]) {
  if constexpr (NW32 % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NW32 / 4; ++i) {
      const u32x4 v = *((const u32x4*)p + i);
      o[4 * i] = v.x; o[4 * i + 1] = v.y; o[4 * i + 2] = v.z; o[4 * i + 3] = v.w;
    }
  } else if constexpr (NW32 == 2) {
    const u32x2 v = *(const u32x2*)p;
    o[0] = v.x; o[1] = v.y;
  } else {
#pragma unroll
    for (int i = 0; i < NW32; ++i) o[i] = ((const uint32_t*)p)[i];
  }
}

template <typename T, int EPL, int NW32>
__device__ __forceinline__ void raw_to_f32(const uint32_t (&r)[NW32], float (&o)[EPL]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = __uint_as_float(r[e]);
  } else if constexpr (std::is_same<T, bf16>::value) {
#pragma unroll
    for (int e = 0; e < EPL / 2; ++e) { o[2 * e] = __uint_as_float(r[e] << 16); o[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
  } else {
#pragma unroll
    for (int e = 0; e < EPL / 2; ++e) {
      const f16x2 h = __builtin_bit_cast(f16x2, r[e]);
      o[2 * e] = (float)h.x; o[2 * e + 1] = (float)h.y;
    }
  }
}


// The last split's combine: (max, sum, O[d]) of every split for one (head, d).  The loads of four splits are issued
// together (a loop over a run-time split count with the loads inside waits one L2 round trip per load: ~8 in a row
// for 4 splits), then merged in split order -- deterministic whoever arrived last.
template <int D>
__device__ __forceinline__ float combine_splits(const float* pp, int nsplit, int d) {
  float mn = -1e30f, L = 0.f, O = 0.f;
  for (int i0 = 0; i0 < nsplit; i0 += 4) {
    float mv[4], lv[4], ov[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* q = pp + (size_t)min(i0 + j, nsplit - 1) * (D + 2);
      mv[j] = __hip_atomic_load(&q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lv[j] = __hip_atomic_load(&q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ov[j] = __hip_atomic_load(&q[2 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float bm = mn;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i0 + j < nsplit) bm = fmaxf(bm, mv[j]);
    const float cs = __builtin_amdgcn_exp2f(mn - bm);
    L *= cs; O *= cs; mn = bm;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < nsplit) {
        const float cw = __builtin_amdgcn_exp2f(mv[j] - mn);
        L = fmaf(lv[j], cw, L);
        O = fmaf(ov[j], cw, O);
      }
  }
  return O / L;
}

// D must be a multiple of 32 for 16-bit caches (two elements per dword per lane); float caches any D % 16 == 0
template <typename T, int D, int G, bool NORM, bool PAGED>
__global__ __launch_bounds__(512) void attn_decode_kernel(AttnDecodeCall c) {
  constexpr int EPL = D / 16, NWV = 8, U = 8;
  constexpr int NW32 = EPL * (int)sizeof(T) / 4;
  constexpr bool PACKED = sizeof(T) == 2;
  constexpr float LOG2E = 1.4426950408889634f;
  const AttnShape& s = c.s;
  const int split = blockIdx.x, bh = blockIdx.y;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, gq = lane >> 4;
  // the host knows every row's length (it advances them itself); taking it from the kernel arguments
  // removes a dependent global load from the head of the chain
  const int kb = c.n_host_off > 0 ? c.host_row[b] : (s.rows ? s.rows[b] : b);   // cache row of batch entry b
  const int pos = c.n_host_off > 0 ? c.host_off[b] : c.offsets[kb];
  const int n_keys = pos + 1;
  const int chunk = (n_keys + c.nsplit - 1) / c.nsplit;
  const int s0 = split * chunk, s1 = min(n_keys, s0 + chunk);
  const bool owner = pos >= s0 && pos < s1;
  const int nq = s.Hq * D;
  const T* row = (const T*)c.qkv + (size_t)b * (nq + 2 * s.Hkv * D);

  __shared__ float st_m[NWV][G], st_l[NWV][G];
  __shared__ float st_o[NWV][G][D];
  __shared__ int is_last_sh;

  // ---- the K/V rows of the first block go out before anything else: they depend on nothing
  // but the offsets, and their latency then hides the q / k_new prologue
  T* kc = (T*)c.kcache;                          // + kv_elem(row, head, key): contiguous slabs or the block arena
  T* vc = (T*)c.vcache;
  const int send = min(s1, pos);                 // cached keys of this split: [s0, send)
  const T* kbase = kc + li * EPL;
  const T* vbase = vc + li * EPL;
  uint32_t kr[U][NW32], vr[U][NW32];
  bool ok[U];
  auto issue_kv = [&](int base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sp = base + 4 * wave + gq + 4 * NWV * u;
      ok[u] = sp < send;
      const int spc = ok[u] ? sp : s0;
      const size_t ro = kv_elem<PAGED>(s, kb, kh, spc);
      load_raw<NW32>(kbase + ro, kr[u]);
      load_raw<NW32>(vbase + ro, vr[u]);
    }
  };
  if (s0 < send) issue_kv(s0);

  // ---- prologue: this lane's pieces of q (G heads), k_new, v_new; norm + RoPE in registers.
  // Every load is issued before the first use (straight-line code: a load under a branch costs a
  // vmcnt(0) drain of the K/V rows already in flight and a serialized L2 round trip each).
  const bool hi = li >= 8;                       // this lane holds the second half of a RoPE pair
  uint32_t qraw[G][NW32], kraw[NW32], vnr[NW32];
#pragma unroll
  for (int g = 0; g < G; ++g) load_raw<NW32>(row + (size_t)(kh * G + g) * D + li * EPL, qraw[g]);
  load_raw<NW32>(row + nq + (size_t)kh * D + li * EPL, kraw);
  load_raw<NW32>(row + nq + (size_t)(s.Hkv + kh) * D + li * EPL, vnr);
  float cs[EPL], sn[EPL];
  {
    const float* cp = c.cos_tab + (size_t)pos * (D / 2) + (li & 7) * EPL;
    const float* sp = c.sin_tab + (size_t)pos * (D / 2) + (li & 7) * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { cs[e] = cp[e]; sn[e] = sp[e]; }
  }
  float qnw[EPL], knw[EPL];
  if constexpr (NORM) {
    uint32_t r0[NW32], r1[NW32];
    load_raw<NW32>((const T*)c.q_norm_w + li * EPL, r0);
    load_raw<NW32>((const T*)c.k_norm_w + li * EPL, r1);
    raw_to_f32<T, EPL, NW32>(r0, qnw);
    raw_to_f32<T, EPL, NW32>(r1, knw);
  }
  auto norm_rope = [&](float (&x)[EPL], const float (&nw)[EPL]) {
    if constexpr (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) ss = fmaf(x[e], x[e], ss);
      ss = row16_sum(ss);
      const float rs = 1.0f / sqrtf(ss / (float)D + c.eps);
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        x[e] = to_f32(store_act<T>(to_f32(store_act<T>(x[e] * rs, s.rnd)) * nw[e], s.rnd));
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const float p = row16_ror8(x[e]);          // the partner lane (li ^ 8) of the same row
      const float o = hi ? (p * sn[e] + x[e] * cs[e]) : (x[e] * cs[e] - p * sn[e]);
      x[e] = to_f32(store_act<T>(o, s.rnd));
    }
  };
  auto pack = [&](const float (&x)[EPL], uint32_t (&r)[NW32]) {
    if constexpr (PACKED) {
#pragma unroll
      for (int e = 0; e < EPL / 2; ++e) {
        T a[2] = {(T)x[2 * e], (T)x[2 * e + 1]};
        r[e] = __builtin_bit_cast(uint32_t, a);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPL; ++e) r[e] = __float_as_uint(x[e]);
    }
  };

  uint32_t qr[G][NW32];                          // q after norm + RoPE, in the cache's element type
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float x[EPL];
    raw_to_f32<T, EPL, NW32>(qraw[g], x);
    norm_rope(x, qnw);
    pack(x, qr[g]);
  }
  uint32_t knr[NW32];
  {
    float x[EPL];
    raw_to_f32<T, EPL, NW32>(kraw, x);
    norm_rope(x, knw);
    pack(x, knr);
  }
  if (owner && wave == 0 && gq == 0 && pos < s.cap) {
    const size_t ro = kv_elem<PAGED>(s, kb, kh, pos) + li * EPL;
#pragma unroll
    for (int i = 0; i < NW32; ++i) {
      ((uint32_t*)(kc + ro))[i] = knr[i];
      ((uint32_t*)(vc + ro))[i] = vnr[i];
    }
  }

  // ---- streaming softmax state (exp2 domain; finite "minus infinity" keeps every exp2 argument finite)
  const float sc2 = c.scale * LOG2E;
  float m[G], l[G], o[G][EPL];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -1e30f; l[g] = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[g][e] = 0.f;
  }
  auto score = [&](const uint32_t (&kr)[NW32], int g) -> float {
    float d = 0.f;
    if constexpr (PACKED) {
#pragma unroll
      for (int i = 0; i < NW32; ++i) d = dot2<T>(kr[i], qr[g][i], d);
    } else {
#pragma unroll
      for (int i = 0; i < NW32; ++i) d = fmaf(__uint_as_float(kr[i]), __uint_as_float(qr[g][i]), d);
    }
    return d;
  };
  // NK keys held by this lane group: scores -> one rescale -> P.V
  auto block = [&](auto& kr, auto& vr, const bool* ok, auto nk_tag) {
    constexpr int NK = decltype(nk_tag)::value;
    float d[NK][G];
#pragma unroll
    for (int u = 0; u < NK; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) d[u][g] = score(kr[u], g);
#pragma unroll
    for (int u = 0; u < NK; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float v = row16_sum(d[u][g]) * sc2;
        d[u][g] = ok[u] ? v : -INFINITY;
      }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float mn = m[g];
#pragma unroll
      for (int u = 0; u < NK; ++u) mn = fmaxf(mn, d[u][g]);
      const float corr = __builtin_amdgcn_exp2f(m[g] - mn);
      l[g] *= corr;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[g][e] *= corr;
#pragma unroll
      for (int u = 0; u < NK; ++u) { d[u][g] = __builtin_amdgcn_exp2f(d[u][g] - mn); l[g] += d[u][g]; }
      m[g] = mn;
    }
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      float vf[EPL];
      raw_to_f32<T, EPL, NW32>(vr[u], vf);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[g][e] = fmaf(d[u][g], vf[e], o[g][e]);
    }
  };

  for (int base = s0; base < send; base += 4 * NWV * U) {     // uniform trip count
    if (base != s0) issue_kv(base);
    block(kr, vr, ok, std::integral_constant<int, U>{});
  }
  if (owner && wave == 0) {                      // the new key, from registers (wave-uniform branch)
    uint32_t k1[1][NW32], v1[1][NW32];
#pragma unroll
    for (int i = 0; i < NW32; ++i) { k1[0][i] = knr[i]; v1[0][i] = vnr[i]; }
    const bool ok1[1] = {gq == 0};
    block(k1, v1, ok1, std::integral_constant<int, 1>{});
  }

  // ---- merge the four lane groups of the wave (lanes li, li+16, li+32, li+48), then the waves
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(m[g], off, 64);
      const float lo = __shfl_xor(l[g], off, 64);
      const float mn = fmaxf(m[g], mo);
      const float ca = __builtin_amdgcn_exp2f(m[g] - mn), cb = __builtin_amdgcn_exp2f(mo - mn);
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
  T* out = (T*)c.out + (size_t)b * nq;
  for (int idx = tid; idx < G * D; idx += 512) {
    const int g = idx / D, d = idx % D;
    float mn = -1e30f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) mn = fmaxf(mn, st_m[w][g]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const float cw = __builtin_amdgcn_exp2f(st_m[w][g] - mn);
      L = fmaf(st_l[w][g], cw, L);
      O = fmaf(st_o[w][g][d], cw, O);
    }
    const int h = kh * G + g;
    if (c.nsplit == 1) {
      out[(size_t)h * D + d] = store_act<T>(O / L, c.rnd_out);
    } else {
      // partials are published with write-through (sc1) stores and read back with sc1 loads: no
      // release / acquire fence is needed (an agent release fence here writes back the whole XCD
      // L2 and was measured at ~8 us per launch with the previous kernel's output still dirty)
      float* pp = c.partial + (((size_t)b * s.Hq + h) * c.nsplit + split) * (D + 2);
      __hip_atomic_store(&pp[2 + d], O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) {
        __hip_atomic_store(&pp[0], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pp[1], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (c.nsplit == 1 || c.counters == nullptr) return;    // (counters == nullptr: timing experiments only)

  // ---- every storing wave drains its stores, the workgroup meets, ONE lane takes a ticket; the
  // workgroup that draws the last ticket of this (b, kv-head) combines (it waits for nobody)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(&c.counters[bh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == c.nsplit - 1;
    if (last) __hip_atomic_store(&c.counters[bh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    is_last_sh = last;
  }
  __syncthreads();
  if (!is_last_sh) return;
  for (int idx = tid; idx < G * D; idx += 512) {
    const int g = idx / D, d = idx % D, h = kh * G + g;
    const float* pp = c.partial + ((size_t)b * s.Hq + h) * c.nsplit * (D + 2);
    out[(size_t)h * D + d] = store_act<T>(combine_splits<D>(pp, c.nsplit, d), c.rnd_out);
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA form of the same launch for 16-bit caches (D a multiple of 32).  Same grid and split
// publication.  The prologue (q / k_new RMSNorm + RoPE, K/V append) is spread over the lane groups,
// one vector each, and meets in LDS; the new key is scored from there and merged as a ninth partial;
// the cached keys -- cut evenly over the splits, 256 per workgroup per round -- go through the
// matrix cores.  Measured (Mistral-7B geometry, B = 8, 1024 keys, 4 splits): 14.1 us against 17.8 us
// for the VALU kernel; a workgroup's own timeline is ~8 us, of which ~3 us are the first global loads.
//   * a wave owns 32 consecutive keys per round (8 waves: 256 keys per workgroup per round) and has
//     their K fragments (16 B per lane, straight from the cache rows: lane (c16, g) of tile t reads
//     K[key 16t + c16][32kk + 8g ..]) and their V rows in flight before the prologue starts;
//   * S^T = K Q^T: v_mfma_16x16x32 with A = K tile, B = Q^T (the G query heads in columns 0..G-1,
//     the other columns zero) -> lane (c16 = head, g) holds the scores of keys 4g + r of each tile;
//   * after exp2 those eight values ARE the B fragment of the next product: the MFMA k index is a
//     free labelling, slot j of lane group g = key 4g + j (tile 0) / 16 + 4g + j - 4 (tile 1);
//   * O^T = V^T P: V rows are written to a per-wave LDS image [16-d tile][key][16 d] (32-B rows) and
//     read back transposed with ds_read_b64_tr_b16 (lane group g: the 4 keys 4g..4g+3 of a tile, lane
//     i of the group receives column i) -- conflict-free, and exactly the key order of P above;
//   * lane (c16 = head, g) ends up with O[head][16 dt + 4g + r]; the online-softmax rescale is a per-
//     lane scalar.  Waves, the new key and the splits are merged as in the VALU kernel.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ f32x4 mfma_kq(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  T v[2] = {(T)a, (T)b};
  return __builtin_bit_cast(uint32_t, v);
}

// P as TWO 16-bit operands (round 4): hi = T(p), lo = T(p - hi).  The reference's attention keeps the softmax numerators in
// float32 (App. A.4); feeding them to the matrix core as ONE 16-bit value was a rounding the oracle does not have -- the
// CPU variant with that rounding (oracle/numerics.py SDPA_P16) reproduces the 1.05 - 1.33 x excess of the device's mean
// |logprob - exact| over the float32-accumulating variants.  hi + lo carries 16+ mantissa bits; the second
// MFMA per V tile costs ~0.2 us per launch.
template <typename T>
__device__ __forceinline__ void pack2_split(float a, float b, uint32_t& hi, uint32_t& lo) {
  const T ha = (T)a, hb = (T)b;
  T h[2] = {ha, hb};
  T l[2] = {(T)(a - (float)ha), (T)(b - (float)hb)};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

template <int G, int D, int ES = 2, int NWV = 8>
__host__ __device__ constexpr size_t attn_mfma_vimg_bytes() {     // float32 caches, exact products (ES 4): no V image; split operands (ES 8): a hi and a lo image per wave
  return ES == 2 ? (size_t)NWV * 32 * D * 2 : ES == 8 ? (size_t)NWV * 32 * D * 4 : 0;
}

template <int G, int D, int ES = 2, int NWV = 8>
__host__ __device__ constexpr size_t attn_mfma_lds_bytes() {
  // [V images: NWV waves x 32 keys x D x 2 B (16-bit caches only)][q + new key: (G + 1) x D x ES B]
  // [st_o: (NWV + 1) x G x D floats][st_m, st_l: (NWV + 1) x G floats each]
  return attn_mfma_vimg_bytes<G, D, ES, NWV>() + (size_t)(G + 1) * D * (ES == 8 ? 4 : ES) + (size_t)(NWV + 1) * G * D * 4 + (size_t)2 * (NWV + 1) * G * 4 + 16;
}

// float32 caches (the PagedKVCache mode, base.py:104-140): v_mfma_f32_16x16x4_f32 -- exact float32 products, one float per
// lane and operand.  The MFMA k index and the row index of the A operand are free labellings, which lets every operand
// come straight from 16-byte global loads with no LDS image and no transpose:
//   * S^T = K Q^T: lane (c16 = key, g4) loads K[key][16 i + 4 g4 .. + 3] (i < D/16: per instruction a row gives 64
//     contiguous bytes, as in the 16-bit kernel); step (i, e) multiplies element e of piece i, i.e. lane group g4
//     supplies d = 16 i + 4 g4 + e, and lane (c16 = head, g4) holds the same d of q.  Scores land as in the 16-bit
//     kernel: lane (head, g4) has keys 4 g4 + r.
//   * O^T = V^T P: step r of a 16-key tile takes key 4 g4 + r from lane group g4 -- exactly the P value that lane holds
//     in register r.  Lane (c16, g4) loads V[key 4 g4 + r][64 h + 4 c16 .. + 3]: per instruction 256 contiguous bytes of
//     each of 4 rows; element e of that load is row c16 of the A tile (h, e), so output tile (h, e) row c16' = 4 g4 + reg
//     is d = 64 h + 4 c16' + e.  D/4 MFMAs per key tile for each product (32 + 32 at D = 128, 32 cycles each).
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct NoHook { __device__ __forceinline__ void operator()() const {} };

template <typename T>
__device__ __forceinline__ void store_out(T* p, T v, bool write_through) {
  if (write_through) {   // read by other workgroups of this same launch (attn_decode_o_kernel)
    if constexpr (sizeof(T) == 2)
      __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      __hip_atomic_store((unsigned*)p, __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

// The launch as a device function: `after_loads()` runs once the body's own first global loads are in flight;
// WT: publish `out` write-through.  Every thread of the workgroup returns from it (no thread exits the kernel
// inside).  The two hooks exist for hosting another phase in the same launch: o_proj + residual behind the
// attention (its whole weight matrix fits the workgroups' prefetch registers) was built that way, bit-identical,
// and measured SLOWER than two launches -- 36.7 us with the weights prefetched from `after_loads`, 33.7 us
// prefetched after the attention, against 19.6 + 12.1 us: the per-CU memory queue is in order, so the 128 KiB
// of weight loads per CU sit in front of every dependent load of the attention's latency chain.  Dropped.
// NWV = waves per workgroup: 8.  A twelve-wave form (768 threads, three waves per SIMD, <= 168 registers) that covers
// 4 splits x 12 waves x 32 keys = 1536 keys in ONE round -- the contexts just above 1024, where eight waves pay a second,
// nearly empty round -- was built, oracle-tested and MEASURED on the bench (KV 1032 -> 1100): bf16 caches 2251 vs
// 2257 tok/s, float32 caches 1928 vs 1962: the extra round of eight waves costs less than four more waves' prologue,
// LDS merge (13 slots) and the third wave per SIMD.  Kept behind -DMI_ATTN_WIDE (compile-time: it doubles this file's
// instantiations), not built by default.
// SPLIT (float32 caches only, round 4; -DMI_ATTN_DECODE_SPLIT_BUILD): q, K, P and V as two bf16 terms each (hi + lo, 16+ mantissa
// bits) and every product as three v_mfma_f32_16x16x32_bf16 (hi.hi + hi.lo + lo.hi) in the 16-bit kernel's layouts -- 32
// keys per wave and round (a 1024-key context in ONE round of four splits, where the exact float32 form below needs two) and
// a fifth of the matrix-core cycles of v_mfma_f32_16x16x4_f32.  K fragments are split in registers, V rows while they are
// written to the wave's two LDS images.  Measured SLOWER than the exact form (launch_mfma_g has the numbers): not
// instantiated by default.  (The same idea pays in the PREFILL attention, where the matrix core is the bound.)
template <typename T, int D, int G, bool NORM, bool WT, bool PAGED, int NWV, bool SPLIT, class Hook>
__device__ __forceinline__ void attn_decode_mfma_body(const AttnDecodeCall& c, unsigned char* smem, Hook after_loads) {
  constexpr bool T32 = sizeof(T) == 4;           // float32 q / caches / outputs
  constexpr bool F32 = T32 && !SPLIT;            // ... multiplied exactly on v_mfma_f32_16x16x4_f32
  constexpr bool F32S = T32 && SPLIT;            // ... or as two-term bf16 splits on v_mfma_f32_16x16x32_bf16
  static_assert(!SPLIT || T32, "SPLIT is a mode of the float32 caches");
  constexpr int ESL = F32S ? 8 : (int)sizeof(T); // the LDS layout's element-size code (attn_mfma_lds_bytes)
  static_assert(D % 32 == 0 && D <= 128 && G <= 8 && (!F32 || D % 64 == 0), "16-bit caches: head_dim 32/64/96/128; float32: 64/128");
  constexpr int EPL = D / 16, NW32 = EPL * (int)sizeof(T) / 4, NTH = NWV * 64;
  constexpr int KK = D / 32, DT = D / 16, NV = (32 * D * 2) / (64 * 16);   // K steps, 16-d tiles, 16-B V loads per lane
  constexpr int NVS = (32 * D * 4) / (64 * 16);  // F32S: 16-B (4-float) V loads per lane and round
  constexpr int KPW = F32 ? 16 : 32;             // keys per wave and round
  constexpr int NP = D / 16, NH = D / 64 > 0 ? D / 64 : 1;   // float32: 16-byte K pieces per lane and tile, 64-d halves of a V row
  constexpr float LOG2E = 1.4426950408889634f;
  const AttnShape& s = c.s;
  const int split = blockIdx.x, bh = blockIdx.y;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, gq = lane >> 4;      // VALU prologue view: lane li of a row owns D/16 elements
  const int c16 = li, g4 = gq;                   // MFMA view: column / lane group
  // the host knows every row's length (it advances them itself); taking it from the kernel arguments
  // removes a dependent global load from the head of the chain
  const int kb = c.n_host_off > 0 ? c.host_row[b] : (s.rows ? s.rows[b] : b);   // cache row of batch entry b
  const int pos = c.n_host_off > 0 ? c.host_off[b] : c.offsets[kb];
  // the pos cached keys are cut evenly over the splits (a 1024-key context = 4 x 256 = one round each); the
  // new key is merged by the last split from registers / LDS
  const int chunk = (pos + c.nsplit - 1) / c.nsplit;
  const int s0 = split * chunk;
  const bool owner = split == c.nsplit - 1;
  const int nq = s.Hq * D;
  const T* row = (const T*)c.qkv + (size_t)b * (nq + 2 * s.Hkv * D);

  constexpr size_t VIMG = attn_mfma_vimg_bytes<G, D, ESL, NWV>();
  unsigned char* vimg = smem + (size_t)wave * (32 * D * 2) * (F32S ? 2 : 1);   // this wave's V image (16-bit caches; F32S: hi image, lo image behind it)
  T* q_sh = (T*)(smem + VIMG);                                        // [G + 1][D]: q heads, then the new key
  float* st_o = (float*)(smem + VIMG + (size_t)(G + 1) * D * sizeof(T));   // [NWV + 1][G][D]
  float* st_m = st_o + (NWV + 1) * G * D;                             // [NWV + 1][G]
  float* st_l = st_m + (NWV + 1) * G;
  int& is_last_sh = *(int*)(st_l + (NWV + 1) * G);       // (no static __shared__ in front of the dynamic region)

  T* kc = (T*)c.kcache;                          // + kv_elem(row, head, key): contiguous slabs or the block arena
  T* vc = (T*)c.vcache;
  const int send = min(s0 + chunk, pos);         // cached keys of this split: [s0, send)

  // ---- the K fragments and V rows of the first round go out before anything else
  u32x4 kf[T32 ? 1 : 2][T32 ? 1 : KK], vrow[T32 ? 1 : NV];
  f32x4 kf32[F32 ? NP : 1], vv32[F32 ? 4 : 1][F32 ? NH : 1];
  f32x4 kraw[F32S ? 2 : 1][F32S ? KK : 1][2], vraw[F32S ? NVS : 1];   // F32S: the float32 bits of the 16-bit kernel's fragments / rows
  const int klast = max(send - 1, 0);            // every address is clamped to a valid row: no load sits under a branch
  auto issue_k = [&](int base) {                 // keys base + KPW wave + [0, KPW)
    const int k0 = base + KPW * wave;
    if constexpr (F32) {
      const int key = min(k0 + c16, klast);
      const size_t ro = kv_elem<PAGED>(s, kb, kh, key);
#pragma unroll
      for (int i = 0; i < NP; ++i) kf32[i] = *(const f32x4*)(kc + ro + 16 * i + 4 * g4);
    } else if constexpr (F32S) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = min(k0 + 16 * t + c16, klast);
        const size_t ro = kv_elem<PAGED>(s, kb, kh, key);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          kraw[t][kk][0] = *(const f32x4*)(kc + ro + 32 * kk + 8 * g4);
          kraw[t][kk][1] = *(const f32x4*)(kc + ro + 32 * kk + 8 * g4 + 4);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = min(k0 + 16 * t + c16, klast);
        const size_t ro = kv_elem<PAGED>(s, kb, kh, key);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[t][kk] = *(const u32x4*)(kc + ro + 32 * kk + 8 * g4);
      }
    }
  };
  auto issue_v = [&](int base) {
    const int k0 = base + KPW * wave;
    if constexpr (F32) {                         // key 4 g4 + r of the tile, 256 contiguous bytes per row and instruction
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = min(k0 + 4 * g4 + r, klast);
        const size_t ro = kv_elem<PAGED>(s, kb, kh, key);
#pragma unroll
        for (int h = 0; h < NH; ++h) vv32[r][h] = *(const f32x4*)(vc + ro + 64 * h + 4 * c16);
      }
    } else if constexpr (F32S) {                 // 32 keys x (D/4) 16-byte pieces of 4 floats, lane-linear
#pragma unroll
      for (int i = 0; i < NVS; ++i) {
        const int piece = i * 64 + lane, kl = piece / (D / 4), dc4 = piece % (D / 4);
        const int key = min(k0 + kl, klast);
        vraw[i] = *(const f32x4*)(vc + kv_elem<PAGED>(s, kb, kh, key) + 4 * dc4);
      }
    } else {                                     // 32 keys x (D/8) 16-byte pieces, lane-linear: piece = i * 64 + lane
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int piece = i * 64 + lane, kl = piece / (D / 8), dc = piece % (D / 8);
        const int key = min(k0 + kl, klast);
        vrow[i] = *(const u32x4*)(vc + kv_elem<PAGED>(s, kb, kh, key) + 8 * dc);
      }
    }
  };
  issue_k(s0);
  issue_v(s0);

  // ---- prologue, spread over the workgroup: lane group (wave, gq) = vector vi owns ONE of the G query
  // heads (vi < G), the new key (vi == G) or the new value (vi == G + 1): raw load, RMSNorm, RoPE,
  // result to LDS.  One code path for every group (pointers are selected, nothing branches), so a wave
  // spends the time of one vector, not of G + 1.
  const int vi = wave * 4 + gq;
  const bool is_q = vi < G, is_k = vi == G, is_v = vi == G + 1;
  const bool hi = li >= 8;
  const T* src = row + (is_q ? (size_t)(kh * G + vi) * D : is_k ? (size_t)nq + (size_t)kh * D
                                                               : (size_t)nq + (size_t)(s.Hkv + kh) * D) + li * EPL;
  uint32_t raw[NW32];
  load_raw<NW32>(src, raw);
  float cs[EPL], sn[EPL];
  {
    const float* cp = c.cos_tab + (size_t)pos * (D / 2) + (li & 7) * EPL;
    const float* sp = c.sin_tab + (size_t)pos * (D / 2) + (li & 7) * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { cs[e] = cp[e]; sn[e] = sp[e]; }
  }
  uint32_t r0[NW32];
  if constexpr (NORM) load_raw<NW32>((const T*)(is_k ? c.k_norm_w : c.q_norm_w) + li * EPL, r0);
  __builtin_amdgcn_sched_barrier(0);
  after_loads();                                 // every global load the prologue waits for is in the queue ahead of this
  __builtin_amdgcn_sched_barrier(0);
  float x[EPL];
  raw_to_f32<T, EPL, NW32>(raw, x);
  if constexpr (NORM) {
    float nw[EPL];
    raw_to_f32<T, EPL, NW32>(r0, nw);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) ss = fmaf(x[e], x[e], ss);
    ss = row16_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)D + c.eps);
#pragma unroll
    for (int e = 0; e < EPL; ++e)
      x[e] = to_f32(store_act<T>(to_f32(store_act<T>(x[e] * rs, s.rnd)) * nw[e], s.rnd));
  }
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const float p = row16_ror8(x[e]);
    const float o = hi ? (p * sn[e] + x[e] * cs[e]) : (x[e] * cs[e] - p * sn[e]);
    x[e] = to_f32(store_act<T>(o, s.rnd));
  }
  uint32_t pk[NW32];
  if constexpr (T32) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) pk[e] = is_v ? raw[e] : __float_as_uint(x[e]);
  } else {
#pragma unroll
    for (int e = 0; e < EPL / 2; ++e) pk[e] = is_v ? raw[e] : pack2<T>(x[2 * e], x[2 * e + 1]);   // v: untouched bits
  }
  if (is_q || is_k) {                            // q heads at rows 0..G-1 of q_sh, the new key at row G
#pragma unroll
    for (int i = 0; i < NW32; ++i) ((uint32_t*)(q_sh + (size_t)vi * D + li * EPL))[i] = pk[i];
  }
  if (owner && (is_k || is_v) && pos < s.cap) {  // KV append (base.py:66-85)
    T* dst = (is_k ? kc : vc) + kv_elem<PAGED>(s, kb, kh, pos) + li * EPL;
#pragma unroll
    for (int i = 0; i < NW32; ++i) ((uint32_t*)dst)[i] = pk[i];
  }
  if (is_v) {                                    // merge slot 8 = the new key: O = v_new (for every head)
    float vf[EPL];
    raw_to_f32<T, EPL, NW32>(raw, vf);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < EPL; ++e) st_o[(NWV * G + g) * D + li * EPL + e] = owner ? vf[e] : 0.f;
  }
  const float sc2 = c.scale * LOG2E;
  __syncthreads();
  if (wave == NWV - 1) {                         // scores of the new key: q_sh rows 0..G-1 against row G
    uint32_t kn[NW32];
#pragma unroll
    for (int i = 0; i < NW32; ++i) kn[i] = ((const uint32_t*)(q_sh + (size_t)G * D + li * EPL))[i];
#pragma unroll
    for (int j = 0; j < (G + 3) / 4; ++j) {
      const int g = min(gq + 4 * j, G - 1);
      float d = 0.f;
      if constexpr (T32) {
#pragma unroll
        for (int i = 0; i < NW32; ++i)
          d = fmaf(__uint_as_float(kn[i]), __uint_as_float(((const uint32_t*)(q_sh + (size_t)g * D + li * EPL))[i]), d);
      } else {
#pragma unroll
        for (int i = 0; i < NW32; ++i) d = dot2<T>(kn[i], ((const uint32_t*)(q_sh + (size_t)g * D + li * EPL))[i], d);
      }
      d = row16_sum(d) * sc2;
      if (li == 0 && gq + 4 * j < G) { st_m[NWV * G + g] = owner ? d : -1e30f; st_l[NWV * G + g] = owner ? 1.f : 0.f; }
    }
  }
  u32x4 qf[T32 ? 1 : KK];
  u32x4 qh[F32S ? KK : 1], ql[F32S ? KK : 1];
  f32x4 qf32[F32 ? NP : 1];
  auto split8 = [](const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {     // 8 floats -> 8 bf16 hi + 8 bf16 lo
    uint32_t h[4], l[4];
    pack2_split<bf16>(a.x, a.y, h[0], l[0]);
    pack2_split<bf16>(a.z, a.w, h[1], l[1]);
    pack2_split<bf16>(b.x, b.y, h[2], l[2]);
    pack2_split<bf16>(b.z, b.w, h[3], l[3]);
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
  };
  if constexpr (F32S) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const float* qp = (const float*)q_sh + (size_t)(c16 < G ? c16 : 0) * D + 32 * kk + 8 * g4;
      f32x4 a = *(const f32x4*)qp, bq = *(const f32x4*)(qp + 4);
      if (c16 >= G) { a = f32x4{0.f, 0.f, 0.f, 0.f}; bq = a; }
      split8(a, bq, qh[kk], ql[kk]);
    }
  } else if constexpr (F32) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      qf32[i] = *(const f32x4*)(q_sh + (size_t)(c16 < G ? c16 : 0) * D + 16 * i + 4 * g4);
      if (c16 >= G) qf32[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      qf[kk] = *(const u32x4*)(q_sh + (size_t)(c16 < G ? c16 : 0) * D + 32 * kk + 8 * g4);
      if (c16 >= G) qf[kk] = u32x4{0u, 0u, 0u, 0u};
    }
  }

  // ---- rounds of 256 keys per workgroup
  float m_run = -1e30f, l_run = 0.f;             // of head c16 (lanes c16 < G), over this lane group's keys
  f32x4 accO[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) accO[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (F32) {
    for (int base = s0; base < send; base += KPW * NWV) {       // uniform trip count; 16 keys per wave and round
      f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sc = mfma_f32(kf32[i][e], qf32[i][e], sc);
      __builtin_amdgcn_sched_barrier(0);
      issue_k(base + KPW * NWV);                 // rolling prefetch: the next round's K into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
      const int k0 = base + KPW * wave;
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = (k0 + 4 * g4 + r) < send;
        sc[r] = ok ? sc[r] * sc2 : -INFINITY;
        mx = fmaxf(mx, sc[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));    // max over the wave's 16 keys, per head
      const float mn = fmaxf(m_run, mx);
      const float corr = __builtin_amdgcn_exp2f(m_run - mn);
      m_run = mn;
      l_run *= corr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) accO[dt] *= corr;
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[r] - mn); l_run += p[r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) accO[h * 4 + e] = mfma_f32(vv32[r][h][e], p[r], accO[h * 4 + e]);
      __builtin_amdgcn_sched_barrier(0);
      issue_v(base + KPW * NWV);                 // ... and the next round's V rows
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (F32S) {
    for (int base = s0; base < send; base += 32 * NWV) {      // uniform trip count; 32 keys per wave and round
      f32x4 sc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          u32x4 kfh, kfl;
          split8(kraw[t][kk][0], kraw[t][kk][1], kfh, kfl);
          sc[t] = mfma_kq<bf16>(kfh, qh[kk], sc[t]);
          sc[t] = mfma_kq<bf16>(kfh, ql[kk], sc[t]);
          sc[t] = mfma_kq<bf16>(kfl, qh[kk], sc[t]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      issue_k(base + 32 * NWV);                  // rolling prefetch: the next round's K into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
      const int k0 = base + 32 * wave;
      float mx = -1e30f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (k0 + 16 * t + 4 * g4 + r) < send;
          sc[t][r] = ok ? sc[t][r] * sc2 : -INFINITY;
          mx = fmaxf(mx, sc[t][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));    // max over the wave's 32 keys, per head
      const float mn = fmaxf(m_run, mx);
      const float corr = __builtin_amdgcn_exp2f(m_run - mn);
      m_run = mn;
      l_run *= corr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) accO[dt] *= corr;
      float p[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[t][r] = __builtin_amdgcn_exp2f(sc[t][r] - mn); l_run += p[t][r]; }
      uint32_t ph[4], pw[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        pack2_split<bf16>(p[t][0], p[t][1], ph[2 * t], pw[2 * t]);
        pack2_split<bf16>(p[t][2], p[t][3], ph[2 * t + 1], pw[2 * t + 1]);
      }
      const u32x4 pf = {ph[0], ph[1], ph[2], ph[3]}, pl = {pw[0], pw[1], pw[2], pw[3]};
      // V rows -> hi / lo -> this wave's two images [16-d tile][key][16 d]
      unsigned char* vlo = vimg + 32 * D * 2;
#pragma unroll
      for (int i = 0; i < NVS; ++i) {
        const int piece = i * 64 + lane, kl = piece / (D / 4), dc4 = piece % (D / 4), dc = dc4 >> 1;
        const size_t va = (size_t)(dc >> 1) * 1024 + kl * 32 + (dc & 1) * 16 + (dc4 & 1) * 8;
        uint32_t h0, l0, h1, l1;
        pack2_split<bf16>(vraw[i].x, vraw[i].y, h0, l0);
        pack2_split<bf16>(vraw[i].z, vraw[i].w, h1, l1);
        *(u32x2v*)(vimg + va) = u32x2v{h0, h1};
        *(u32x2v*)(vlo + va) = u32x2v{l0, l1};
      }
      __builtin_amdgcn_sched_barrier(0);
      issue_v(base + 32 * NWV);                  // ... and the next round's V rows
      __builtin_amdgcn_sched_barrier(0);
      const int tq = c16 >> 2, tp = c16 & 3;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const size_t a0 = (size_t)dt * 1024 + (4 * g4 + tq) * 32 + tp * 8;
        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vimg + a0));
        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vimg + a0 + 16 * 32));
        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vlo + a0));
        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vlo + a0 + 16 * 32));
        const u32x4 vfh = {((const uint32_t*)&h0)[0], ((const uint32_t*)&h0)[1], ((const uint32_t*)&h1)[0], ((const uint32_t*)&h1)[1]};
        const u32x4 vfl = {((const uint32_t*)&l0)[0], ((const uint32_t*)&l0)[1], ((const uint32_t*)&l1)[0], ((const uint32_t*)&l1)[1]};
        accO[dt] = mfma_kq<bf16>(vfh, pf, accO[dt]);
        accO[dt] = mfma_kq<bf16>(vfh, pl, accO[dt]);
        accO[dt] = mfma_kq<bf16>(vfl, pf, accO[dt]);
      }
    }
  } else {
  for (int base = s0; base < send; base += 32 * NWV) {        // uniform trip count
    // S^T tiles: rows = keys, columns = heads
    f32x4 sc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) sc[t] = mfma_kq<T>(kf[t][kk], qf[kk], sc[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_k(base + 32 * NWV);                    // rolling prefetch: the next round's K into the registers just consumed
    __builtin_amdgcn_sched_barrier(0);
    const int k0 = base + 32 * wave;
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = (k0 + 16 * t + 4 * g4 + r) < send;
        sc[t][r] = ok ? sc[t][r] * sc2 : -INFINITY;
        mx = fmaxf(mx, sc[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // max over the wave's 32 keys, per head
    const float mn = fmaxf(m_run, mx);
    const float corr = __builtin_amdgcn_exp2f(m_run - mn);
    m_run = mn;
    l_run *= corr;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accO[dt] *= corr;
    float p[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[t][r] = __builtin_amdgcn_exp2f(sc[t][r] - mn); l_run += p[t][r]; }
    uint32_t ph[4], pw[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      pack2_split<T>(p[t][0], p[t][1], ph[2 * t], pw[2 * t]);
      pack2_split<T>(p[t][2], p[t][3], ph[2 * t + 1], pw[2 * t + 1]);
    }
    const u32x4 pf = {ph[0], ph[1], ph[2], ph[3]}, pl = {pw[0], pw[1], pw[2], pw[3]};
    // V rows -> this wave's image [16-d tile][key][16 d]
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int piece = i * 64 + lane, kl = piece / (D / 8), dc = piece % (D / 8);
      *(u32x4*)(vimg + (size_t)(dc >> 1) * 1024 + kl * 32 + (dc & 1) * 16) = vrow[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_v(base + 32 * NWV);                    // ... and the next round's V rows
    __builtin_amdgcn_sched_barrier(0);
    // O^T tiles: A = V^T (transposed reads), B = P
    const int tq = c16 >> 2, tp = c16 & 3;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const unsigned char* a0 = vimg + (size_t)dt * 1024 + (4 * g4 + tq) * 32 + tp * 8;
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 32));
      const uint32_t* w0 = (const uint32_t*)&v0;
      const uint32_t* w1 = (const uint32_t*)&v1;
      const u32x4 vf = {w0[0], w0[1], w1[0], w1[1]};
      accO[dt] = mfma_kq<T>(vf, pf, accO[dt]);
      accO[dt] = mfma_kq<T>(vf, pl, accO[dt]);
    }
  }
  }
  // ---- this wave's (m, l, O) for the heads in columns c16 < G
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (c16 < G) {
    if (g4 == 0) { st_m[wave * G + c16] = m_run; st_l[wave * G + c16] = l_run; }
    if constexpr (F32) {                         // tile (h, e), row 4 g4 + r  <->  d = 64 h + 4 (4 g4 + r) + e
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st_o[(wave * G + c16) * D + 64 * (dt >> 2) + 16 * g4 + 4 * r + (dt & 3)] = accO[dt][r];
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st_o[(wave * G + c16) * D + 16 * dt + 4 * g4 + r] = accO[dt][r];
    }
  }
  __syncthreads();
  T* out = (T*)c.out + (size_t)b * nq;
  for (int idx = tid; idx < G * D; idx += NTH) {
    const int g = idx / D, d = idx % D;
    float mn = -1e30f;
#pragma unroll
    for (int w = 0; w < NWV + 1; ++w) mn = fmaxf(mn, st_m[w * G + g]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NWV + 1; ++w) {
      const float cw = __builtin_amdgcn_exp2f(st_m[w * G + g] - mn);
      L = fmaf(st_l[w * G + g], cw, L);
      O = fmaf(st_o[(w * G + g) * D + d], cw, O);
    }
    const int h = kh * G + g;
    if (c.nsplit == 1) {
      store_out<T>(&out[(size_t)h * D + d], store_act<T>(O / L, c.rnd_out), WT);
    } else {
      float* pp = c.partial + (((size_t)b * s.Hq + h) * c.nsplit + split) * (D + 2);
      __hip_atomic_store(&pp[2 + d], O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) {
        __hip_atomic_store(&pp[0], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pp[1], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (c.nsplit == 1 || c.counters == nullptr) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(&c.counters[bh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == c.nsplit - 1;
    if (last) __hip_atomic_store(&c.counters[bh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last_sh = last;
  }
  __syncthreads();
  if (!is_last_sh) return;
  for (int idx = tid; idx < G * D; idx += NTH) {
    const int g = idx / D, d = idx % D, h = kh * G + g;
    const float* pp = c.partial + ((size_t)b * s.Hq + h) * c.nsplit * (D + 2);
    store_out<T>(&out[(size_t)h * D + d], store_act<T>(combine_splits<D>(pp, c.nsplit, d), c.rnd_out), WT);
  }
}

template <typename T, int D, int G, bool NORM, bool PAGED, int NWV, bool SPLIT = false>
__global__ __launch_bounds__(NWV * 64) void attn_decode_mfma_kernel(AttnDecodeCall c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn_decode_mfma_body<T, D, G, NORM, false, PAGED, NWV, SPLIT>(c, smem, NoHook{});
}

template <typename T, int D, int G, bool NORM>
int launch_mfma_g(const AttnDecodeCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  const dim3 grid(c.nsplit, s.B * s.Hkv);
  // twelve waves where eight would need a second round and twelve do not (the host knows the row lengths)
  constexpr int KPW = sizeof(T) == 4 ? 16 : 32;
  bool wide = false;
#ifdef MI_ATTN_WIDE
  wide = c.variant == 2;                         // (variant 2: always twelve waves -- kernel tests)
  if (!wide && c.n_host_off > 0) {
    int mx = 0;
    for (int b = 0; b < c.n_host_off; ++b) mx = std::max(mx, c.host_off[b]);
    const int chunk = (mx + c.nsplit - 1) / c.nsplit;
    static const bool allow = getenv("MI_ATTN_NO_WIDE") == nullptr;
    wide = allow && (chunk + KPW * 8 - 1) / (KPW * 8) > (chunk + KPW * 12 - 1) / (KPW * 12);
  }
#endif
#define LAUNCH_MFMA(PG, NW) do { \
    auto kern = attn_decode_mfma_kernel<T, D, G, NORM, PG, NW>; \
    constexpr size_t lds = attn_mfma_lds_bytes<G, D, (int)sizeof(T), NW>(); \
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, c); } while (0)
#ifdef MI_ATTN_WIDE
  if (s.btab) { if (wide) LAUNCH_MFMA(true, 12); else LAUNCH_MFMA(true, 8); }
  else { if (wide) LAUNCH_MFMA(false, 12); else LAUNCH_MFMA(false, 8); }
#else
  (void)wide; (void)KPW;
#ifdef MI_ATTN_DECODE_SPLIT_BUILD
  if constexpr (sizeof(T) == 4) {
    // float32 caches on two-term bf16 operands (SPLIT): BUILT, oracle-tested (test_fused_decode_attention ran it as variant
    // 0) and MEASURED SLOWER than the exact float32 form on the bench -- 1913 / 1907 against 1998 / 1987 tok/s, same box,
    // alternating: at 8 x 1024 keys the exact kernel's matrix-core time is ~3.4 us of its 19.7 and what the split saves there
    // it spends converting K and V (2 x 64 floats per lane and round) and on the two LDS images.  Compiled only with
    // -DMI_ATTN_DECODE_SPLIT_BUILD (it doubles the float32 instantiations); then MI_ATTN_DECODE_F32_SPLIT=1 selects it.
    const char* env = getenv("MI_ATTN_DECODE_F32_SPLIT");
    constexpr bool fits = attn_mfma_lds_bytes<G, D, 8, 8>() <= (size_t)160 * 1024;   // (G = 8 at D = 128 does not fit the LDS)
    const bool split = fits && c.variant != 3 && env != nullptr && atoi(env) != 0;
#define LAUNCH_MFMA_S(PG) do { \
    auto kern = attn_decode_mfma_kernel<T, D, G, NORM, PG, 8, true>; \
    constexpr size_t lds = attn_mfma_lds_bytes<G, D, 8, 8>(); \
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, grid, dim3(8 * 64), lds, st, c); } while (0)
    if constexpr (fits) {
      if (split) { if (s.btab) LAUNCH_MFMA_S(true); else LAUNCH_MFMA_S(false); }
    }
    if (!split) { if (s.btab) LAUNCH_MFMA(true, 8); else LAUNCH_MFMA(false, 8); }
#undef LAUNCH_MFMA_S
  } else
#endif
  {
    if (s.btab) LAUNCH_MFMA(true, 8); else LAUNCH_MFMA(false, 8);
  }
#endif
#undef LAUNCH_MFMA
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename T, int D>
constexpr bool attn_mfma_ok() { return (sizeof(T) == 2 && D % 32 == 0) || (sizeof(T) == 4 && D % 64 == 0); }

template <typename T, int D, bool NORM>
int launch_mfma_gn(const AttnDecodeCall& c, hipStream_t st) {
  if constexpr (attn_mfma_ok<T, D>()) {
    switch (c.s.Hq / c.s.Hkv) {
      case 1: return launch_mfma_g<T, D, 1, NORM>(c, st);
      case 2: return launch_mfma_g<T, D, 2, NORM>(c, st);
      case 4: return launch_mfma_g<T, D, 4, NORM>(c, st);
      case 5: return launch_mfma_g<T, D, 5, NORM>(c, st);
      case 8: return launch_mfma_g<T, D, 8, NORM>(c, st);
    }
  }
  return fail(MI_ERR_UNSUPPORTED, "attention: Hq/Hkv must be 1, 2, 4, 5 or 8");
}

template <typename T, int D, bool NORM>
int launch_gn(const AttnDecodeCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  if constexpr (attn_mfma_ok<T, D>()) {
    if (c.variant != 1) return launch_mfma_gn<T, D, NORM>(c, st);
  }
  const dim3 grid(c.nsplit, s.B * s.Hkv), block(512);
#define DK(GV) do { if (s.btab) hipLaunchKernelGGL((attn_decode_kernel<T, D, GV, NORM, true>), grid, block, 0, st, c); \
                    else hipLaunchKernelGGL((attn_decode_kernel<T, D, GV, NORM, false>), grid, block, 0, st, c); } while (0)
  switch (s.Hq / s.Hkv) {
    case 1: DK(1); break;
    case 2: DK(2); break;
    case 4: DK(4); break;
    case 5: DK(5); break;
    case 8: DK(8); break;
    default: return fail(MI_ERR_UNSUPPORTED, "attention: Hq/Hkv must be 1, 2, 4, 5 or 8");
  }
#undef DK
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename T, int D>
int launch_g(const AttnDecodeCall& c, hipStream_t st) {
  const bool norm = c.q_norm_w != nullptr && c.k_norm_w != nullptr;
  if ((c.q_norm_w != nullptr) != (c.k_norm_w != nullptr)) return fail(MI_ERR_INVALID, "attention_decode: q_norm and k_norm must come together");
  return norm ? launch_gn<T, D, true>(c, st) : launch_gn<T, D, false>(c, st);
}

template <typename T>
int launch_d(const AttnDecodeCall& c, hipStream_t st) {
  switch (c.s.D) {
    case 16: if constexpr (sizeof(T) == 4) return launch_g<T, 16>(c, st); break;
    case 32: return launch_g<T, 32>(c, st);
    case 64: return launch_g<T, 64>(c, st);
    case 128: return launch_g<T, 128>(c, st);
  }
  return fail(MI_ERR_UNSUPPORTED, "attention_decode: head_dim not supported by the fused kernel");
}

}  // namespace

bool attention_decode_supported(const AttnShape& s) {
  if (s.L != 1 || s.act != s.kv) return false;
  const int G = s.Hkv > 0 ? s.Hq / s.Hkv : 0;
  if (!(G == 1 || G == 2 || G == 4 || G == 5 || G == 8)) return false;
  if (s.act == MI_F32) return s.D == 16 || s.D == 32 || s.D == 64 || s.D == 128;
  return s.D == 32 || s.D == 64 || s.D == 128;
}

int launch_attention_decode(const AttnDecodeCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  if (!attention_decode_supported(s)) return fail(MI_ERR_UNSUPPORTED, "attention_decode: shape / dtype not supported");
  if (c.nsplit < 1 || (c.nsplit > 1 && c.partial == nullptr))
    return fail(MI_ERR_INVALID, "attention_decode: bad split configuration");
  if (s.act == MI_F32) return launch_d<float>(c, st);
  if (s.act == MI_BF16) return launch_d<bf16>(c, st);
  return launch_d<f16>(c, st);
}

}  // namespace mi
