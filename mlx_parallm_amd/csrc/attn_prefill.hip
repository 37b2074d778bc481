// attn_prefill.hip -- causal attention of the prefill call (L > 1 new tokens per row) on the matrix cores.
//
// scaled_dot_product_attention(q, keys, values, scale, mask = create_additive_causal_mask_variable)
// (llama.py:139-141, qwen3.py:111-113, base.py:6-40) over a cache that rope_append has already extended
// with this call's K / V rows: query t of row b sees keys 0 .. offsets[b] + t (left padding is attended
// like any other token, quirk Q1).
//
// Workgroup = one (row b, kv head, tile of 16 queries): its G = Hq/Hkv waves are the G query heads
// of the group, so every wave needs exactly the same keys and the K / V blocks (32 keys) are staged
// once in LDS for all of them (registers -> LDS, the next block's global loads in flight during the
// current block's MFMAs; one barrier per block).  Per wave and block, as in the decode kernel
// (attn_decode.hip): S^T = K Q^T with A = K fragments read contiguously from the [tile][kk][g][key]
// image, B = Q^T held in registers; the eight scores of a lane are, after exp2, the B fragment of
// O^T = V^T P, whose A fragments come out of the [16-d tile][key][16 d] image through
// ds_read_b64_tr_b16.  Lane (c16 = query, g) owns the running max / sum of its query and
// O[query][16 dt + 4g + r].  Tiles are launched heaviest (latest queries) first.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace mi {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void pack2_split(float a, float b, uint32_t& hi, uint32_t& lo) {   // as in attn_decode.hip
  const T ha = (T)a, hb = (T)b;
  T h[2] = {ha, hb};
  T l[2] = {(T)(a - (float)ha), (T)(b - (float)hb)};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  T v[2] = {(T)a, (T)b};
  return __builtin_bit_cast(uint32_t, v);
}

constexpr int QT = 16;      // queries per query group (one MFMA column set)
constexpr int KB = 32;      // keys per block

// QG: query groups of 16 per wave (the launcher uses 1: see launch_pg).  With QG = 2 a workgroup covers 32 queries and every
// K / V fragment a wave reads from LDS feeds two MFMAs.
// (launch bounds: two waves per SIMD, i.e. up to 256 registers.  Left to itself hipcc aims at three, keeps the output
// accumulators in AGPRs and moves them to VGPRs and back around every rescale: 88 v_accvgpr moves per block of 16 MFMAs.)
template <typename T, int D, int G, bool PAGED, int QG>
__global__ __launch_bounds__(G * 64, 2) void attn_prefill_kernel(AttnCall c) {
  static_assert(sizeof(T) == 2 && D % 32 == 0 && D <= 128, "16-bit activations / caches, head_dim 32..128");
  constexpr int KK = D / 32, DT = D / 16, NT = G * 64;
  constexpr int QTT = QT * QG;                        // queries per workgroup
  constexpr int NPIECE = KB * D / 8;                  // 16-byte pieces of one K (or V) block
  constexpr int NP = (NPIECE + NT - 1) / NT;          // per thread
  constexpr int IMG = KB * D * 2;                     // bytes of one image
  constexpr float LOG2E = 1.4426950408889634f;
  const AttnShape& s = c.s;
  const int qt = gridDim.x - 1 - blockIdx.x, bh = blockIdx.y;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = s.rows ? s.rows[b] : b;              // cache row of batch entry b
  const int off = c.offsets[kb];
  const int t0 = qt * QTT;
  const int nk = off + min(t0 + QTT, s.L);            // keys any query of the tile may see
  const int nb = (nk + KB - 1) / KB;
  const int h = kh * G + wave;

  __shared__ __attribute__((aligned(16))) unsigned char kimg[2][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char vimg[2][IMG];

  const T* kc = (const T*)c.kcache;
  const T* vc = (const T*)c.vcache;

  // Two register sets: the loads of block j + 2 are issued at the head of iteration j and written to LDS at the end of
  // iteration j + 1 -- two iterations (~2 x 600 cycles of work) to arrive.  With ONE block ahead an iteration lasted as
  // long as an L2 / HBM round trip (3.3 us per workgroup and block at three workgroups per CU, stamps of the bench shape).
  u32x4 kreg[2][NP], vreg[2][NP];
  auto load_block = [&](int j, u32x4 (&kr)[NP], u32x4 (&vr)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = min(i * NT + tid, NPIECE - 1);
      const int key = min(j * KB + piece / (D / 8), nk - 1), dc = piece % (D / 8);
      const size_t ro = kv_elem<PAGED>(s, kb, kh, key) + 8 * dc;
      kr[i] = *(const u32x4*)(kc + ro);
      vr[i] = *(const u32x4*)(vc + ro);
    }
  };
  auto store_block = [&](int buf, const u32x4 (&kr)[NP], const u32x4 (&vr)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = i * NT + tid;
      if (piece < NPIECE) {
        const int key = piece / (D / 8), dc = piece % (D / 8);
        // K: [16-key tile][kk][g][key & 15] x 16 B -- a fragment read of (tile, kk) is 1 KiB contiguous
        *(u32x4*)(kimg[buf] + ((((key >> 4) * KK + (dc >> 2)) * 4 + (dc & 3)) * 16 + (key & 15)) * 16) = kr[i];
        // V: [16-d tile][key][16 d] (32-byte rows) for the transposed reads
        *(u32x4*)(vimg[buf] + (size_t)(dc >> 1) * (KB * 32) + key * 32 + (dc & 1) * 16) = vr[i];
      }
    }
  };

  load_block(0, kreg[0], vreg[0]);
  load_block(1, kreg[1], vreg[1]);                    // (past the last block: clamped keys, never stored)
  // Q^T fragments of this wave's head (B operand: column = query), one set per query group
  u32x4 qf[QG][KK];
  int tq[QG];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    tq[u] = min(t0 + QT * u + c16, s.L - 1);          // this lane's query of group u (clamped; stores are guarded)
    const T* qp = (const T*)c.q + ((size_t)b * s.L + tq[u]) * s.Hq * D + (size_t)h * D;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[u][kk] = *(const u32x4*)(qp + 32 * kk + 8 * g4);
  }
  store_block(0, kreg[0], vreg[0]);
  // The Q fragments count as loaded from HERE on.  Without this "use" hipcc's wait-count pass carries their pending loads
  // round the loop's back edge and puts s_waitcnt vmcnt(0) in front of the first MFMAs of EVERY iteration -- which drains
  // the K / V loads just issued for two blocks ahead: one exposed memory round trip per block (290 -> 1xx us per layer).
#pragma unroll
  for (int u = 0; u < QG; ++u)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" :: "v"(qf[u][kk]));
  __syncthreads();

  const float sc2 = c.scale * LOG2E;
  float m_run[QG], l_run[QG];
  f32x4 accO[QG][DT];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    m_run[u] = -1e30f; l_run[u] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accO[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tq4 = c16 >> 2, tp = c16 & 3;

  // iteration j: kfree / vfree = the set block j was stored from (free), knext / vnext = the set holding block j + 1
  auto iter = [&](int j, u32x4 (&kfree)[NP], u32x4 (&vfree)[NP], const u32x4 (&knext)[NP], const u32x4 (&vnext)[NP]) {
    const int buf = j & 1;
    load_block(j + 2, kfree, vfree);                  // straight-line (past the last block: clamped keys, never stored); these loads fly during two blocks' work
    f32x4 sc[QG][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int u = 0; u < QG; ++u) sc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const u32x4 kf = *(const u32x4*)(kimg[buf] + (((t * KK + kk) * 4 + g4) * 16 + c16) * 16);
#pragma unroll
        for (int u = 0; u < QG; ++u) sc[u][t] = mfma16<T>(kf, qf[u][kk], sc[u][t]);
      }
    }
    u32x4 pf[QG], pl[QG];                             // P = hi + lo, two 16-bit operands (attn_decode.hip pack2_split)
#pragma unroll
    for (int u = 0; u < QG; ++u) {
      // The vector ALU, not the matrix core, bounds this loop (~110 vector instructions + 9 transcendentals per 16 MFMAs
      // as first written), so: the causal mask only in the blocks that reach past the group's first query (uniform), the
      // scale folded into the exponent's FMA, and O / l rescaled only when some query's running max moved (uniform;
      // multiplying by exp2(0) = 1 is the identity, so skipping it changes no bit).
      const int qpos = off + t0 + QT * u + c16;       // key positions <= qpos are visible to this lane's query of group u
      float mx = -INFINITY;
      if (j * KB + KB - 1 > off + t0 + QT * u) {      // a diagonal block of this group
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (j * KB + 16 * t + 4 * g4 + r) <= qpos;
            sc[u][t][r] = ok ? sc[u][t][r] : -INFINITY;
          }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[u][t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m_run[u], mx * sc2);     // (scale > 0: the largest raw score is the largest scaled one)
      if (__builtin_amdgcn_ballot_w64(mn != m_run[u]) != 0) {
        const float corr = __builtin_amdgcn_exp2f(m_run[u] - mn);
        m_run[u] = mn;
        l_run[u] *= corr;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) accO[u][dt] *= corr;
      }
      float p[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[u][t][r], sc2, -mn)); l_run[u] += p[t][r]; }
      uint32_t ph[4], pw[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        pack2_split<T>(p[t][0], p[t][1], ph[2 * t], pw[2 * t]);
        pack2_split<T>(p[t][2], p[t][3], ph[2 * t + 1], pw[2 * t + 1]);
      }
      pf[u] = u32x4{ph[0], ph[1], ph[2], ph[3]};
      pl[u] = u32x4{pw[0], pw[1], pw[2], pw[3]};
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const unsigned char* a0 = vimg[buf] + (size_t)dt * (KB * 32) + (4 * g4 + tq4) * 32 + tp * 8;
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 32));
      const uint32_t* w0 = (const uint32_t*)&v0;
      const uint32_t* w1 = (const uint32_t*)&v1;
      const u32x4 vf = {w0[0], w0[1], w1[0], w1[1]};
#pragma unroll
      for (int u = 0; u < QG; ++u) {
        accO[u][dt] = mfma16<T>(vf, pf[u], accO[u][dt]);
        accO[u][dt] = mfma16<T>(vf, pl[u], accO[u][dt]);
      }
    }
    if (j + 1 < nb) store_block(buf ^ 1, knext, vnext);   // the other buffer: nobody reads it during this block
    __syncthreads();
  };
  for (int j = 0; j < nb; j += 2) {
    iter(j, kreg[0], vreg[0], kreg[1], vreg[1]);
    if (j + 1 < nb) iter(j + 1, kreg[1], vreg[1], kreg[0], vreg[0]);
  }

#pragma unroll
  for (int u = 0; u < QG; ++u) {
    float l = l_run[u];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (t0 + QT * u + c16 < s.L) {
      const float inv = 1.0f / l;
      T* op = (T*)c.out + ((size_t)b * s.L + tq[u]) * s.Hq * D + (size_t)h * D + 4 * g4;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const u32x2 v = {pack2<T>(accO[u][dt][0] * inv, accO[u][dt][1] * inv), pack2<T>(accO[u][dt][2] * inv, accO[u][dt][3] * inv)};
        *(u32x2*)(op + 16 * dt) = v;
      }
    }
  }
}

// one KiB: global (per-lane source address) -> LDS at a wave-uniform address (lane-linear)
__device__ __forceinline__ void dma_kib(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// (launch bounds: two waves per SIMD, i.e. up to 256 registers.  Left to itself hipcc aims at three, keeps the output
// accumulators in AGPRs and moves them to VGPRs and back around every rescale: 88 v_accvgpr moves per block of 16 MFMAs.)
template <typename T, int D, int G, bool PAGED, int QG>
__global__ __launch_bounds__(G * 64, G >= 2 ? 2 : 1) void attn_prefill_dma_kernel(AttnCall c) {
  static_assert(sizeof(T) == 2 && D % 32 == 0 && D <= 128, "16-bit activations / caches, head_dim 32..128");
  constexpr int KK = D / 32, DT = D / 16, NT = G * 64;
  constexpr int QTT = QT * QG;                        // queries per workgroup
  constexpr int NPIECE = KB * D / 8;                  // 16-byte pieces of one K (or V) block
  constexpr int NP = (NPIECE + NT - 1) / NT;          // per thread
  constexpr int IMG = KB * D * 2;                     // bytes of one image
  constexpr float LOG2E = 1.4426950408889634f;
  const AttnShape& s = c.s;
  const int qt = gridDim.x - 1 - blockIdx.x, bh = blockIdx.y;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = s.rows ? s.rows[b] : b;              // cache row of batch entry b
  const int off = c.offsets[kb];
  const int t0 = qt * QTT;
  const int nk = off + min(t0 + QTT, s.L);            // keys any query of the tile may see
  const int nb = (nk + KB - 1) / KB;
  const int h = kh * G + wave;

  // K / V blocks go global -> LDS by LDS-DMA into a ring of THREE buffers: block j + 2 is issued at the head of
  // iteration j (into the buffer block j - 1 was read from: every wave finished those reads before the barrier that
  // closed iteration j - 1) and waited for at the end of iteration j + 1 (s_waitcnt vmcnt(that block's DMAs of this
  // wave), then the barrier: the block is everybody's).  No staging registers, no ds_write -- and no compiler-visible
  // vector loads in the loop: with K / V loaded into registers hipcc's wait-count pass put s_waitcnt vmcnt(0) at the head
  // of the loop body (a register-reuse hazard it could not rule out across the back edge), which drained the prefetch
  // just issued and exposed a memory round trip every block: 268 us per layer at 8 x 1024 tokens against 191 here.
  // The images are those of attn_prefill_kernel; a DMA writes lane-linear, so the layout is realised through each
  // lane's SOURCE address: K instruction i covers slots 64 i .. 64 i + 63 of [16-key tile][kk][g][key & 15], V
  // instruction i the 32 keys x 32 bytes of 16-d tile i.
  constexpr int NINST = NPIECE / 64;                  // DMA instructions per image and block
  constexpr int CNT = (NINST + G - 1) / G;            // per wave (the last wave may repeat the last instruction: same bytes, same place)
  __shared__ __attribute__((aligned(16))) unsigned char kimg[3][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char vimg[3][IMG];
  const unsigned k_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&kimg[0][0];
  const unsigned v_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&vimg[0][0];

  const T* kc = (const T*)c.kcache;
  const T* vc = (const T*)c.vcache;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto dma_block = [&](int j, int buf) {
#pragma unroll
    for (int n = 0; n < CNT; ++n) {
      const int i = min(wv + G * n, NINST - 1);
      {   // K: slot S = 64 i + lane = (((kt * KK + kk) * 4 + dclo) * 16 + klo)
        const int S = i * 64 + lane;
        const int klo = S & 15, dclo = (S >> 4) & 3, tk = S >> 6, kk = tk % KK, kt = tk / KK;
        const int key = min(j * KB + kt * 16 + klo, nk - 1);
        const T* src = kc + kv_elem<PAGED>(s, kb, kh, key) + 8 * (kk * 4 + dclo);
        dma_kib(src, k_lds + buf * IMG + i * 1024);
      }
      {   // V: slot S = dt * 64 + key * 2 + (dc & 1)
        const int key = min(j * KB + (lane >> 1), nk - 1);
        const T* src = vc + kv_elem<PAGED>(s, kb, kh, key) + 8 * (i * 2 + (lane & 1));
        dma_kib(src, v_lds + buf * IMG + i * 1024);
      }
    }
  };

  // Q^T fragments of this wave's head (B operand: column = query), one set per query group
  u32x4 qf[QG][KK];
  int tq[QG];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    tq[u] = min(t0 + QT * u + c16, s.L - 1);          // this lane's query of group u (clamped; stores are guarded)
    const T* qp = (const T*)c.q + ((size_t)b * s.L + tq[u]) * s.Hq * D + (size_t)h * D;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[u][kk] = *(const u32x4*)(qp + 32 * kk + 8 * g4);
  }
  dma_block(0, 0);
  dma_block(1, 1);                                    // (past the last block: clamped keys, masked)
  // the Q fragments count as loaded from here on (hipcc waits for its own loads with vmcnt(0): blocks 0 and 1 have landed too)
#pragma unroll
  for (int u = 0; u < QG; ++u)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" :: "v"(qf[u][kk]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float sc2 = c.scale * LOG2E;
  float m_run[QG], l_run[QG];
  f32x4 accO[QG][DT];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    m_run[u] = -1e30f; l_run[u] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accO[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tq4 = c16 >> 2, tp = c16 & 3;

  int buf = 0;                                        // j mod 3
  for (int j = 0; j < nb; ++j) {
    const int bnext2 = buf == 0 ? 2 : buf - 1;        // (j + 2) mod 3
    dma_block(j + 2, bnext2);
    f32x4 sc[QG][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int u = 0; u < QG; ++u) sc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const u32x4 kf = *(const u32x4*)(kimg[buf] + (((t * KK + kk) * 4 + g4) * 16 + c16) * 16);
#pragma unroll
        for (int u = 0; u < QG; ++u) sc[u][t] = mfma16<T>(kf, qf[u][kk], sc[u][t]);
      }
    }
    u32x4 pf[QG], pl[QG];                             // P = hi + lo, two 16-bit operands (attn_decode.hip pack2_split)
#pragma unroll
    for (int u = 0; u < QG; ++u) {
      // The vector ALU, not the matrix core, bounds this loop (~110 vector instructions + 9 transcendentals per 16 MFMAs
      // as first written), so: the causal mask only in the blocks that reach past the group's first query (uniform), the
      // scale folded into the exponent's FMA, and O / l rescaled only when some query's running max moved (uniform;
      // multiplying by exp2(0) = 1 is the identity, so skipping it changes no bit).
      const int qpos = off + t0 + QT * u + c16;       // key positions <= qpos are visible to this lane's query of group u
      float mx = -INFINITY;
      if (j * KB + KB - 1 > off + t0 + QT * u) {      // a diagonal block of this group
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (j * KB + 16 * t + 4 * g4 + r) <= qpos;
            sc[u][t][r] = ok ? sc[u][t][r] : -INFINITY;
          }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[u][t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m_run[u], mx * sc2);     // (scale > 0: the largest raw score is the largest scaled one)
      if (__builtin_amdgcn_ballot_w64(mn != m_run[u]) != 0) {
        const float corr = __builtin_amdgcn_exp2f(m_run[u] - mn);
        m_run[u] = mn;
        l_run[u] *= corr;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) accO[u][dt] *= corr;
      }
      float p[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[u][t][r], sc2, -mn)); l_run[u] += p[t][r]; }
      uint32_t ph[4], pw[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        pack2_split<T>(p[t][0], p[t][1], ph[2 * t], pw[2 * t]);
        pack2_split<T>(p[t][2], p[t][3], ph[2 * t + 1], pw[2 * t + 1]);
      }
      pf[u] = u32x4{ph[0], ph[1], ph[2], ph[3]};
      pl[u] = u32x4{pw[0], pw[1], pw[2], pw[3]};
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const unsigned char* a0 = vimg[buf] + (size_t)dt * (KB * 32) + (4 * g4 + tq4) * 32 + tp * 8;
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 32));
      const uint32_t* w0 = (const uint32_t*)&v0;
      const uint32_t* w1 = (const uint32_t*)&v1;
      const u32x4 vf = {w0[0], w0[1], w1[0], w1[1]};
#pragma unroll
      for (int u = 0; u < QG; ++u) {
        accO[u][dt] = mfma16<T>(vf, pf[u], accO[u][dt]);
        accO[u][dt] = mfma16<T>(vf, pl[u], accO[u][dt]);
      }
    }
    // lgkmcnt(0): every LDS read of this iteration has RETURNED before the barrier lets another wave issue the DMA that
    // overwrites this buffer's predecessor -- hipcc may sink an MFMA, and the wait in front of it, below the barrier, and
    // under load (four workgroups per CU on the LDS queue) a read issued just before the barrier was still queued when a
    // fast wave's DMA landed: wrong tiles that came and went with the load on the chip (tests/test_gpu_fullsize.py).
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(2 * CNT) : "memory");    // and block j + 1 has landed (block j + 2 may still fly)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf = buf == 2 ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA of this workgroup may land after it has left the CU

#pragma unroll
  for (int u = 0; u < QG; ++u) {
    float l = l_run[u];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (t0 + QT * u + c16 < s.L) {
      const float inv = 1.0f / l;
      T* op = (T*)c.out + ((size_t)b * s.L + tq[u]) * s.Hq * D + (size_t)h * D + 4 * g4;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const u32x2 v = {pack2<T>(accO[u][dt][0] * inv, accO[u][dt][1] * inv), pack2<T>(accO[u][dt][2] * inv, accO[u][dt][3] * inv)};
        *(u32x2*)(op + 16 * dt) = v;
      }
    }
  }
}

template <typename T, int D, int G>
int launch_pg(const AttnCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  // one query group per wave.  Two (32 queries per workgroup, every K / V fragment feeding two MFMAs) was built and
  // measured: prefill 83.3 .. 83.8 k tok/s against 84.0 .. 84.1 k (same box, alternating runs) -- the vector ALU bounds the
  // loop, not the LDS traffic -- and 79 k once the leaner softmax below pushed it past 256 registers.
  const dim3 grid((s.L + QT - 1) / QT, s.B * s.Hkv), block(G * 64);
  const char* env = getenv("MI_ATTN_PREFILL_DMA");     // A/B and the bit-equality test: 0 = the register-staged kernel (read per call)
  if (env == nullptr || atoi(env) != 0) {
    if (s.btab) hipLaunchKernelGGL((attn_prefill_dma_kernel<T, D, G, true, 1>), grid, block, 0, st, c);
    else hipLaunchKernelGGL((attn_prefill_dma_kernel<T, D, G, false, 1>), grid, block, 0, st, c);
  } else {
    if (s.btab) hipLaunchKernelGGL((attn_prefill_kernel<T, D, G, true, 1>), grid, block, 0, st, c);
    else hipLaunchKernelGGL((attn_prefill_kernel<T, D, G, false, 1>), grid, block, 0, st, c);
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename T, int D>
int launch_pd(const AttnCall& c, hipStream_t st) {
  switch (c.s.Hq / c.s.Hkv) {
    case 1: return launch_pg<T, D, 1>(c, st);
    case 2: return launch_pg<T, D, 2>(c, st);
    case 4: return launch_pg<T, D, 4>(c, st);
    case 5: return launch_pg<T, D, 5>(c, st);
    case 8: return launch_pg<T, D, 8>(c, st);
  }
  return fail(MI_ERR_UNSUPPORTED, "attention_prefill: Hq/Hkv must be 1, 2, 4, 5 or 8");
}

template <typename T>
int launch_pt(const AttnCall& c, hipStream_t st) {
  switch (c.s.D) {
    case 32: return launch_pd<T, 32>(c, st);
    case 64: return launch_pd<T, 64>(c, st);
    case 128: return launch_pd<T, 128>(c, st);
  }
  return fail(MI_ERR_UNSUPPORTED, "attention_prefill: head_dim must be 32, 64 or 128");
}


// ---------------------------------------------------------------------------------------------------------------
// float32 activations and caches (the PagedKVCache mode: base.py:104-140 keeps float32 K / V, and SDPA then runs in
// float32 for every layer): the same workgroup shape on v_mfma_f32_16x16x4_f32 -- exact float32 products, one float per
// lane and operand.  Blocks of 16 keys; both images are laid out in the order the fragments are read, so every LDS
// read is a lane-linear KiB (conflict-free ds_read_b128):
//   * K image [i][g4][key] x 16 B: lane (c16 = key, g4) reads K[key][16 i + 4 g4 .. + 3]; MFMA step (i, e) takes element
//     e, i.e. lane group g4 supplies d = 16 i + 4 g4 + e, and the Q^T fragment holds the same d of the lane's query;
//   * V image [r][h][g4][c16] x 16 B: lane (c16, g4) reads V[key 4 g4 + r][64 h + 4 c16 .. + 3]; step r of the block takes
//     key 4 g4 + r from lane group g4 -- the P value that lane already holds in register r -- and element e is row c16 of
//     the A tile (h, e): output tile (h, e), row 4 g4' + reg, is d = 64 h + 16 g4' + 4 reg + e (attn_decode.hip has the
//     same labelling for the decode step).
constexpr int KB32 = 16;    // keys per block, float32 kernel

template <int D, int G, bool PAGED>
__global__ __launch_bounds__(G * 64, 2) void attn_prefill_f32_kernel(AttnCall c) {
  static_assert(D % 64 == 0 && D <= 128, "float32: head_dim 64 / 128");
  constexpr int NP = D / 16, NH = D / 64, DT = D / 16, NT = G * 64;
  constexpr int NPIECE = KB32 * D / 4;                // 16-byte pieces of one K (or V) block
  constexpr int NLD = (NPIECE + NT - 1) / NT;         // per thread
  constexpr int IMG = KB32 * D * 4;                   // bytes of one image
  constexpr float LOG2E = 1.4426950408889634f;
  const AttnShape& s = c.s;
  const int qt = gridDim.x - 1 - blockIdx.x, bh = blockIdx.y;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = s.rows ? s.rows[b] : b;              // cache row of batch entry b
  const int off = c.offsets[kb];
  const int t0 = qt * QT;
  const int tq = min(t0 + c16, s.L - 1);              // this lane's query (clamped; stores are guarded)
  const int nk = off + min(t0 + QT, s.L);             // keys any query of the tile may see
  const int nb = (nk + KB32 - 1) / KB32;
  const int h = kh * G + wave;

  __shared__ __attribute__((aligned(16))) unsigned char kimg[2][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char vimg[2][IMG];

  const float* kc = (const float*)c.kcache;
  const float* vc = (const float*)c.vcache;

  f32x4 kreg[NLD], vreg[NLD];
  auto load_block = [&](int j) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int piece = min(i * NT + tid, NPIECE - 1);
      const int key = min(j * KB32 + piece / (D / 4), nk - 1), dc = piece % (D / 4);
      const size_t ro = kv_elem<PAGED>(s, kb, kh, key) + 4 * dc;
      kreg[i] = *(const f32x4*)(kc + ro);
      vreg[i] = *(const f32x4*)(vc + ro);
    }
  };
  auto store_block = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int piece = i * NT + tid;
      if (piece < NPIECE) {
        const int key = piece / (D / 4), dc = piece % (D / 4);     // dc: 16-byte piece of the row, d = 4 dc
        *(f32x4*)(kimg[buf] + (size_t)(dc * 16 + key) * 16) = kreg[i];                                   // (i, g4) = (dc / 4, dc % 4)
        *(f32x4*)(vimg[buf] + (size_t)((((key & 3) * NH + (dc >> 4)) * 4 + (key >> 2)) * 16 + (dc & 15)) * 16) = vreg[i];
      }
    }
  };

  load_block(0);
  f32x4 qf[NP];                                       // Q^T fragments of this wave's head (B operand: column = query)
  {
    const float* qp = (const float*)c.q + ((size_t)b * s.L + tq) * s.Hq * D + (size_t)h * D;
#pragma unroll
    for (int i = 0; i < NP; ++i) qf[i] = *(const f32x4*)(qp + 16 * i + 4 * g4);
  }
  store_block(0);
#pragma unroll
  for (int i = 0; i < NP; ++i) asm volatile("" :: "v"(qf[i]));     // (see attn_prefill_kernel: the Q loads are complete from here on)
  __syncthreads();

  const float sc2 = c.scale * LOG2E;
  const int qpos = off + t0 + c16;                    // key positions <= qpos are visible to this lane's query
  float m_run = -1e30f, l_run = 0.f;
  f32x4 accO[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) accO[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < nb; ++j) {
    const int buf = j & 1;
    load_block(j + 1);                                // straight-line (past the last block: clamped keys, never stored); the loads fly during this block's MFMAs
    f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const f32x4 kf = *(const f32x4*)(kimg[buf] + (size_t)((i * 4 + g4) * 16 + c16) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[i][e], sc, 0, 0, 0);
    }
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (j * KB32 + 4 * g4 + r) <= qpos;
      sc[r] = ok ? sc[r] * sc2 : -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
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
      for (int hh = 0; hh < NH; ++hh) {
        const f32x4 vf = *(const f32x4*)(vimg[buf] + (size_t)(((r * NH + hh) * 4 + g4) * 16 + c16) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) accO[hh * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], p[r], accO[hh * 4 + e], 0, 0, 0);
      }
    if (j + 1 < nb) store_block(buf ^ 1);             // the other buffer: nobody reads it during this block
    __syncthreads();
  }

  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (t0 + c16 < s.L) {
    const float inv = 1.0f / l_run;
    float* op = (float*)c.out + ((size_t)b * s.L + tq) * s.Hq * D + (size_t)h * D + 16 * g4;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 v = {round_rt(accO[hh * 4 + 0][r] * inv, s.rnd), round_rt(accO[hh * 4 + 1][r] * inv, s.rnd),
                         round_rt(accO[hh * 4 + 2][r] * inv, s.rnd), round_rt(accO[hh * 4 + 3][r] * inv, s.rnd)};
        *(f32x4*)(op + 64 * hh + 4 * r) = v;
      }
  }
}

// The same launch on the bf16 matrix core with SPLIT OPERANDS (round 4; the default for this mode): q, K, P and V are each
// two bf16 terms, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16+ mantissa bits), and every product is three MFMAs --
// hi.hi + hi.lo + lo.hi (the lo.lo term is 2^-16 of the product) -- of v_mfma_f32_16x16x32_bf16: 48 MFMAs of 16 cycles per
// 32 keys against 128 of 32 cycles on v_mfma_f32_16x16x4_f32.  Workgroup shape, block size, LDS image layouts, masking and
// the online softmax are those of attn_prefill_kernel<bf16> (blocks of 32 keys, one query group per wave); K / V are read
// from the float32 caches and split while they are written to LDS (each element once per workgroup, not once per wave).
// What the 16+ bits cost was looked at on the CPU first (oracle/numerics.py SDPA_SPLIT2: the float32-accumulating
// variants do not move).  MI_ATTN_PREFILL_F32_EXACT=1 (read per call) runs attn_prefill_f32_kernel instead.
template <int D, int G, bool PAGED>
__global__ __launch_bounds__(G * 64, 2) void attn_prefill_f32s_kernel(AttnCall c) {
  static_assert(D % 32 == 0 && D <= 128, "head_dim 32..128");
  using T = bf16;
  constexpr int KK = D / 32, DT = D / 16, NT = G * 64;
  constexpr int NPIECE = KB * D / 4;                  // 16-byte pieces (4 floats) of one K (or V) block
  constexpr int NP = (NPIECE + NT - 1) / NT;          // per thread
  constexpr int IMG = KB * D * 2;                     // bytes of one 16-bit image
  constexpr float LOG2E = 1.4426950408889634f;
  const AttnShape& s = c.s;
  const int qt = gridDim.x - 1 - blockIdx.x, bh = blockIdx.y;
  const int b = bh / s.Hkv, kh = bh % s.Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = s.rows ? s.rows[b] : b;              // cache row of batch entry b
  const int off = c.offsets[kb];
  const int t0 = qt * QT;
  const int nk = off + min(t0 + QT, s.L);             // keys any query of the tile may see
  const int nb = (nk + KB - 1) / KB;
  const int h = kh * G + wave;

  __shared__ __attribute__((aligned(16))) unsigned char khi[2][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char klo[2][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char vhi[2][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char vlo[2][IMG];

  const float* kc = (const float*)c.kcache;
  const float* vc = (const float*)c.vcache;

  f32x4 kreg[NP], vreg[NP];
  auto load_block = [&](int j) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = min(i * NT + tid, NPIECE - 1);
      const int key = min(j * KB + piece / (D / 4), nk - 1), dc = piece % (D / 4);
      const size_t ro = kv_elem<PAGED>(s, kb, kh, key) + 4 * dc;
      kreg[i] = *(const f32x4*)(kc + ro);
      vreg[i] = *(const f32x4*)(vc + ro);
    }
  };
  auto split4 = [](const f32x4& v, u32x2& hi, u32x2& lo) {
    uint32_t h0, l0, h1, l1;
    pack2_split<T>(v.x, v.y, h0, l0);
    pack2_split<T>(v.z, v.w, h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
  };
  auto store_block = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = i * NT + tid;
      if (piece < NPIECE) {
        const int key = piece / (D / 4), dc4 = piece % (D / 4), dc = dc4 >> 1, half = (dc4 & 1) * 8;   // dc: the 8-element piece of the 16-bit images
        u32x2 hi, lo;
        split4(kreg[i], hi, lo);
        // K: [16-key tile][kk][g][key & 15] x 16 B -- a fragment read of (tile, kk) is 1 KiB contiguous
        const size_t ka = (size_t)((((key >> 4) * KK + (dc >> 2)) * 4 + (dc & 3)) * 16 + (key & 15)) * 16 + half;
        *(u32x2*)(khi[buf] + ka) = hi;
        *(u32x2*)(klo[buf] + ka) = lo;
        split4(vreg[i], hi, lo);
        // V: [16-d tile][key][16 d] (32-byte rows) for the transposed reads
        const size_t va = (size_t)(dc >> 1) * (KB * 32) + key * 32 + (dc & 1) * 16 + half;
        *(u32x2*)(vhi[buf] + va) = hi;
        *(u32x2*)(vlo[buf] + va) = lo;
      }
    }
  };

  load_block(0);
  // Q^T fragments of this wave's head (B operand: column = query), hi and lo
  u32x4 qh[KK], ql[KK];
  const int tq = min(t0 + c16, s.L - 1);              // this lane's query (clamped; stores are guarded)
  {
    const float* qp = (const float*)c.q + ((size_t)b * s.L + tq) * s.Hq * D + (size_t)h * D;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const f32x4 a = *(const f32x4*)(qp + 32 * kk + 8 * g4), bq = *(const f32x4*)(qp + 32 * kk + 8 * g4 + 4);
      u32x2 h0, l0, h1, l1;
      split4(a, h0, l0);
      split4(bq, h1, l1);
      qh[kk] = u32x4{h0.x, h0.y, h1.x, h1.y};
      ql[kk] = u32x4{l0.x, l0.y, l1.x, l1.y};
    }
  }
  store_block(0);
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) asm volatile("" :: "v"(qh[kk]), "v"(ql[kk]));     // (see attn_prefill_kernel: the Q loads are complete from here on)
  __syncthreads();

  const float sc2 = c.scale * LOG2E;
  const int qpos = off + t0 + c16;                    // key positions <= qpos are visible to this lane's query
  float m_run = -1e30f, l_run = 0.f;
  f32x4 accO[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) accO[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tq4 = c16 >> 2, tp = c16 & 3;

  for (int j = 0; j < nb; ++j) {
    const int buf = j & 1;
    load_block(j + 1);                                // straight-line (past the last block: clamped keys, never stored); the loads fly during this block's work
    f32x4 sc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const size_t fa = (size_t)(((t * KK + kk) * 4 + g4) * 16 + c16) * 16;
        const u32x4 kfh = *(const u32x4*)(khi[buf] + fa);
        const u32x4 kfl = *(const u32x4*)(klo[buf] + fa);
        sc[t] = mfma16<T>(kfh, qh[kk], sc[t]);
        sc[t] = mfma16<T>(kfh, ql[kk], sc[t]);
        sc[t] = mfma16<T>(kfl, qh[kk], sc[t]);
      }
    }
    float mx = -INFINITY;
    if (j * KB + KB - 1 > off + t0) {                 // a diagonal block (uniform)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (j * KB + 16 * t + 4 * g4 + r) <= qpos;
          sc[t][r] = ok ? sc[t][r] : -INFINITY;
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m_run, mx * sc2);          // (scale > 0: the largest raw score is the largest scaled one)
    if (__builtin_amdgcn_ballot_w64(mn != m_run) != 0) {
      const float corr = __builtin_amdgcn_exp2f(m_run - mn);
      m_run = mn;
      l_run *= corr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) accO[dt] *= corr;
    }
    float p[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], sc2, -mn)); l_run += p[t][r]; }
    uint32_t ph[4], pw[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      pack2_split<T>(p[t][0], p[t][1], ph[2 * t], pw[2 * t]);
      pack2_split<T>(p[t][2], p[t][3], ph[2 * t + 1], pw[2 * t + 1]);
    }
    const u32x4 pf = {ph[0], ph[1], ph[2], ph[3]}, pl = {pw[0], pw[1], pw[2], pw[3]};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const size_t va = (size_t)dt * (KB * 32) + (4 * g4 + tq4) * 32 + tp * 8;
      const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vhi[buf] + va));
      const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vhi[buf] + va + 16 * 32));
      const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vlo[buf] + va));
      const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vlo[buf] + va + 16 * 32));
      const u32x4 vfh = {((const uint32_t*)&h0)[0], ((const uint32_t*)&h0)[1], ((const uint32_t*)&h1)[0], ((const uint32_t*)&h1)[1]};
      const u32x4 vfl = {((const uint32_t*)&l0)[0], ((const uint32_t*)&l0)[1], ((const uint32_t*)&l1)[0], ((const uint32_t*)&l1)[1]};
      accO[dt] = mfma16<T>(vfh, pf, accO[dt]);
      accO[dt] = mfma16<T>(vfh, pl, accO[dt]);
      accO[dt] = mfma16<T>(vfl, pf, accO[dt]);
    }
    if (j + 1 < nb) store_block(buf ^ 1);             // the other buffer: nobody reads it during this block
    __syncthreads();
  }

  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (t0 + c16 < s.L) {
    const float inv = 1.0f / l_run;
    float* op = (float*)c.out + ((size_t)b * s.L + tq) * s.Hq * D + (size_t)h * D + 4 * g4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 v = {round_rt(accO[dt][0] * inv, s.rnd), round_rt(accO[dt][1] * inv, s.rnd),
                       round_rt(accO[dt][2] * inv, s.rnd), round_rt(accO[dt][3] * inv, s.rnd)};
      *(f32x4*)(op + 16 * dt) = v;
    }
  }
}

template <int D, int G>
int launch_pg32(const AttnCall& c, hipStream_t st) {
  const AttnShape& s = c.s;
  const dim3 grid((s.L + QT - 1) / QT, s.B * s.Hkv), block(G * 64);
  const char* env = getenv("MI_ATTN_PREFILL_F32_EXACT");   // A/B and the comparison test (read per call): 1 = exact float32 products
  if (env == nullptr || atoi(env) == 0) {
    if (s.btab) hipLaunchKernelGGL((attn_prefill_f32s_kernel<D, G, true>), grid, block, 0, st, c);
    else hipLaunchKernelGGL((attn_prefill_f32s_kernel<D, G, false>), grid, block, 0, st, c);
  } else if (s.btab) hipLaunchKernelGGL((attn_prefill_f32_kernel<D, G, true>), grid, block, 0, st, c);
  else hipLaunchKernelGGL((attn_prefill_f32_kernel<D, G, false>), grid, block, 0, st, c);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <int D>
int launch_pd32(const AttnCall& c, hipStream_t st) {
  switch (c.s.Hq / c.s.Hkv) {
    case 1: return launch_pg32<D, 1>(c, st);
    case 2: return launch_pg32<D, 2>(c, st);
    case 4: return launch_pg32<D, 4>(c, st);
    case 5: return launch_pg32<D, 5>(c, st);
    case 8: return launch_pg32<D, 8>(c, st);
  }
  return fail(MI_ERR_UNSUPPORTED, "attention_prefill: Hq/Hkv must be 1, 2, 4, 5 or 8");
}

}  // namespace

bool attention_prefill_supported(const AttnShape& s) {
  if (s.L < 2 || s.act != s.kv) return false;
  if (s.Hkv <= 0 || s.Hq % s.Hkv != 0) return false;
  const int G = s.Hq / s.Hkv;
  if (!(G == 1 || G == 2 || G == 4 || G == 5 || G == 8)) return false;
  if (s.act == MI_F32) return s.D == 64 || s.D == 128;        // (s.rnd = logical rounding of the output, applied at the store)
  if ((s.act != MI_BF16 && s.act != MI_F16) || s.rnd != RND_NONE) return false;
  return s.D == 32 || s.D == 64 || s.D == 128;
}

int launch_attention_prefill(const AttnCall& c, hipStream_t st) {
  if (!attention_prefill_supported(c.s)) return fail(MI_ERR_UNSUPPORTED, "attention_prefill: shape / dtype not supported");
  if (c.nsplit != 1) return fail(MI_ERR_INVALID, "attention_prefill: no split-KV");
  if (c.s.act == MI_F32) return c.s.D == 128 ? launch_pd32<128>(c, st) : launch_pd32<64>(c, st);
  return c.s.act == MI_BF16 ? launch_pt<bf16>(c, st) : launch_pt<f16>(c, st);
}

}  // namespace mi
