// gemv_phase.h -- the weight-streaming skinny GEMM of gemv_mfma.hip as a device-side building block (`Phase`):
// gemv_mfma_kernel runs one, gemv_pair_kernel two, attn_decode_o_kernel (attn_decode.hip) runs o_proj behind the
// decode attention.  See gemv_mfma.hip for the design notes.
#pragma once
#include <type_traits>

#include "kernels.h"

namespace mi {
namespace gemv {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct MfmaParams {
  const void* x; int ldx; int M;
  int pro; const void* norm_w; float eps;
  const void* w; const void* scales; const void* biases;
  int N, K;
  int epi; void* out; int ldo; void* resid; int pair_offset;
  int layout;       // always 1 here: tile-major (16 rows x 32 k blocks of 1 KiB, repack.hip)
  int kc;       // K elements staged in LDS per chunk
  const float* lora_t; int lora_t_ld;
  const float* lora_b0; const float* lora_b1;
  int lora_row0_0, lora_n_0, lora_rank_0; float lora_scale_0;
  int lora_row0_1, lora_n_1, lora_rank_1; float lora_scale_1;
#ifdef MI_SK_TRACE
  unsigned long long* trace;       // debug build (tools/debug/build_trace_lib.sh): [workgroup][16] wall-clock stamps
#endif
};

#ifdef MI_SK_TRACE
#define PH_STAMP(i) do { if (p.trace != nullptr && tid == 0) p.trace[(size_t)(w & 1023) * 16 + (i)] = wall_clock64(); } while (0)
#define PH_STAMP_WAVE(i) do { if (p.trace != nullptr && lane == 0) p.trace[(size_t)(w & 1023) * 16 + (i) + wave] = wall_clock64(); } while (0)
#else
#define PH_STAMP(i) do { } while (0)
#define PH_STAMP_WAVE(i) do { } while (0)
#endif

// in-launch seam of gemv_pair_kernel: every workgroup adds 1 to *counter after phase A; phase B starts
// when the counter has reached `target`.  A workgroup that polls `spin_limit` times without seeing it
// sets *error and carries on: never a hang, and never a silent wrong token -- the host copies *error back behind
// every step and fails that step with MI_ERR_RUNTIME (engine.hip: seam_record, mi_step_wait, seam_check_sync).
struct SeamParams {
  unsigned* counter;
  unsigned target;
  unsigned spin_limit;
  int* error;
};

template <typename T>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<T, bf16>::value) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
}

// int4 code pairs -> 16-bit floats.
//   bf16: (16 + q) = 2^4 * (1 + q/16): exponent 0x4180, q in mantissa bits 3..6 -- two VALU ops per
//         pair; the offset 16 is folded into the per-group bias (bias - 16*scale).  (gfx950 has no
//         packed bf16 add; a larger offset such as 128 costs ~8x more cancellation error.)
//   f16:  (1024 + q) via 0x6400 | q, then an exact v_pk_add_f16 of -1024 -> q itself, offset 0.
template <typename T> struct Magic;
template <> struct Magic<bf16> { static constexpr float offs = 16.f; };
template <> struct Magic<f16> { static constexpr float offs = 0.f; };

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ u32x4 unpack_q4(uint32_t v) {
  // fragment element 2p <- nibble p, element 2p+1 <- nibble p+4
  u32x4 r;
  if constexpr (std::is_same<T, bf16>::value) {
    // (x & mask) | magic as ONE v_and_or_b32 with both constants in registers: hipcc emits v_and + v_or with
    // literals (a VOP3 cannot carry two), and this kernel is bound by its VALU instruction count
    const uint32_t mask = 0x00780078u, magic = 0x41804180u;
    auto and_or = [&](uint32_t x) {
      uint32_t o;
      asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(mask), "v"(magic));
      return o;
    };
    r.x = and_or(v << 3);
    r.y = and_or(v >> 1);
    r.z = and_or(v >> 5);
    r.w = and_or(v >> 9);
  } else {
    const f16x2 off = {(_Float16)1024.f, (_Float16)1024.f};
    r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, (v & 0x000F000Fu) | 0x64006400u) - off);
    r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, ((v >> 4) & 0x000F000Fu) | 0x64006400u) - off);
    r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, ((v >> 8) & 0x000F000Fu) | 0x64006400u) - off);
    r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, ((v >> 12) & 0x000F000Fu) | 0x64006400u) - off);
  }
  return r;
}

// ---- activation staging ---------------------------------------------------------------------
// Dense: fragment for (k-block kb of 32, g) = x[m][32kb + 8g .. +8] at slot ((kb*4+g)*MB + m).
// Int4:  128-wide block, quant group s (64 wide), step t, lane group g hold the 8 k's of packed
//        dword d(g,t) = {0,4,2,6}[g] + t of that group, in nibble order (p, p+4 interleaved);
//        slot ((((kb*2+s)*2+t)*4+g)*MB + m).
template <bool Q4>
__device__ __forceinline__ int frag_slot(int k8, int m, int MB) {
  if constexpr (!Q4) {
    return k8 * MB + m;
  } else {
    const int kb = k8 >> 4, s = (k8 >> 3) & 1, d = k8 & 7;
    const int t = d & 1, e = d - t;                 // e in {0,2,4,6}
    const int g = (e == 0) ? 0 : (e == 4) ? 1 : (e == 2) ? 2 : 3;
    return ((((kb * 2 + s) * 2 + t) * 4 + g) * MB + m);
  }
}

// 16 bytes another workgroup of THIS launch has written (write-through): agent-scope loads, which
// are not served from this CU's L1
__device__ __forceinline__ u32x4 load16_agent(const void* p) {
  const unsigned long long* q = (const unsigned long long*)p;
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}

// a dword of two 16-bit floats <-> two floats (element 0 in the low half)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename AT>
__device__ __forceinline__ f32x2 unpack2(uint32_t v) {
  if constexpr (std::is_same<AT, bf16>::value) {
    return f32x2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
  } else {
    const f16x2 h = __builtin_bit_cast(f16x2, v);
    return f32x2{(float)h.x, (float)h.y};
  }
}
template <typename AT>
__device__ __forceinline__ uint32_t pack2(f32x2 f) {
  // round to nearest even, both halves in ONE v_cvt_pk_{bf16,f16}_f32 (written element by element hipcc converts each
  // half on its own and merges them with shifts)
  if constexpr (std::is_same<AT, bf16>::value) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
  } else {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));
  }
}

template <typename AT>
__device__ __forceinline__ void store_elem(AT* p, AT v, bool write_through) {
  if (write_through) {
    __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

// nbuf = 2 when K spans several activation chunks: chunk c+1 is staged into the other buffer while chunk c's
// MFMAs run (no staging bubble in the weight stream)
template <int NW, int NA, bool Q4>
__host__ __device__ constexpr size_t phase_lds_bytes(int kc, int MB, int nbuf) {
  return nbuf * ((size_t)kc * MB * 2 + (Q4 ? (size_t)(kc / 64) * MB * 4 : 0)) + (size_t)NW * NA * 64 * 4 * 4 + 16 * 4 +
         (size_t)NW * 16 * 4;
}
__host__ __device__ constexpr int phase_nbuf(int K, int kc, int MB, bool q4) { (void)q4; return (K > kc && MB == 8) ? 2 : 1; }

// One GEMV.  Work is cut into "batches": TB 16-row tiles x a KS-wide slice of K (KS = NW * UK * BK:
// every wave issues UK 16-byte loads per tile per batch, ALL of them before it touches the
// activations), so a workgroup has its whole batch -- 128 KiB for a dense bf16 tile at K = 4096 --
// in flight at once.  The next batch is issued right after the MFMAs of the current one retire its
// registers, i.e. before the cross-wave reduction and the epilogue, which keeps HBM busy across tiles.
template <typename AT, bool Q4, int MB, bool SWIGLU, int NW, int J, bool DB = false>
struct Phase {
  static constexpr int NT = NW * 64;
  static constexpr int NA = SWIGLU ? 2 : 1;
  static constexpr int BK = Q4 ? 128 : 32;      // k covered by one 16-byte load of the 4 lane groups
  static constexpr int UK = Q4 ? (32 / NW) : (128 / NW) / NA;   // loads per wave per tile per weight stream per batch
  static constexpr int TB = (Q4 && !SWIGLU) ? 2 : 1;   // tiles per batch (int4: 8 weight loads per wave per batch either way)
  static constexpr int KS = NW * UK * BK;       // k-span of a batch (4096 dense / 2048 SwiGLU / 4096 int4)
  // J = staging items per thread per activation row: 1 covers kc <= 8*NT (4096), 2 up to 8192
  using S = AT;                          // scale dtype == activation dtype on this path
  static constexpr bool DBUF = DB;       // double-buffered activation chunks (see phase_nbuf); an instantiation of its own,
                                         // so that the single-chunk kernels keep their register allocation

  const MfmaParams& p;
  // LDS: nbuf x {[frag: kc*MB*2 B][sx: (kc/64)*MB floats (int4)]} [red: NW*NA*64 float4][rs: 16][red2: NW*16]
  u32x4* frag; float* sx; float* red; float* rs_sh; float* red2;
  u32x4* frag_b[2]; float* sx_b[2]; int nbuf;
  u32x4* frag_w; float* sx_w;          // the buffer stage_x writes (frag / sx = the buffer the MFMAs read)
  int tid, lane, wave, c16, g;
  const AT* x;
  // Work items of this workgroup: 16-row tiles dealt round-robin (tile = w + i*G), so that
  // neighbouring workgroups stream neighbouring memory.  (Cutting the left-over tiles into 8-row
  // halves, or giving every workgroup the same number of tiles on a smaller grid, was measured and
  // did not help: the short kernels sit on their latency floor.)
  int G, w, ntiles_all, ntiles, nbatch, nchunks, klen0;

  u32x4 wr[NA][TB][UK];
  uint32_t sr[NA][TB][UK], br[NA][TB][UK];
  u32x4 xv[MB][J];
  f32x4 acc[NA][TB];
  float hpre = 0.f; bool hpre_ok = false;   // residual epilogue: h of this workgroup's FIRST tile, fetched at the head of the phase

  __device__ __forceinline__ Phase(const MfmaParams& pp, unsigned char* smem) : p(pp) {
    nbuf = DB ? 2 : 1;
    const size_t one = (size_t)p.kc * MB * 2 + (Q4 ? (size_t)(p.kc / 64) * MB * 4 : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned char* base = smem + (size_t)(i < nbuf ? i : 0) * one;
      frag_b[i] = (u32x4*)base;
      sx_b[i] = (float*)(base + (size_t)p.kc * MB * 2);
    }
    frag = frag_w = frag_b[0]; sx = sx_w = sx_b[0];
    red = (float*)(smem + (size_t)nbuf * one);
    rs_sh = red + NW * NA * 64 * 4;
    red2 = rs_sh + 16;
    tid = threadIdx.x; lane = tid & 63; wave = tid >> 6; c16 = lane & 15; g = lane >> 4;
    x = (const AT*)p.x;
    G = gridDim.x * gridDim.y; w = blockIdx.y * gridDim.x + blockIdx.x;   // (the attention launch that hosts o_proj is 2-D)
    ntiles_all = p.N / 16;
    ntiles = w < ntiles_all ? (ntiles_all - w + G - 1) / G : 0;
    nbatch = (ntiles + TB - 1) / TB;
    nchunks = (p.K + p.kc - 1) / p.kc;
    klen0 = min(p.kc, p.K);
  }

  __device__ __forceinline__ int item_row0(int i) const { return min(w + i * G, ntiles_all - 1) * 16; }

  // ---- issue slot u of the weight loads of batch (tbi, [k0, k0+KS) clipped to kend).
  // STRAIGHT-LINE on purpose: a load under a branch makes hipcc wait vmcnt(0) at the join, which
  // drains the whole prefetch queue at every step.  Slots past kend / items past the end load a
  // valid (already cached) address instead and are zeroed or ignored by the consumer.
  __device__ __forceinline__ void issue_u(int u, int tbi, int k0, int kend) {
    const int kq = k0 + (u * NW + wave) * BK;
    const int k = kq < kend ? kq : 0;
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int row = item_row0(tbi * TB + t) + c16;
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const size_t rowa = (size_t)(row + a * p.pair_offset);
        // tile-major weights (repack.hip): the 16-row x BK-k block of this load is ONE contiguous KiB
        // (+128 B of scales/biases for int4), lane l at 16 l; consecutive k blocks are consecutive
        if constexpr (!Q4) {
          const char* blk = (const char*)p.w + ((rowa >> 4) * (size_t)(p.K / 32) + (size_t)(k >> 5)) * 1024;
          wr[a][t][u] = __builtin_nontemporal_load((const u32x4*)(blk + (g * 16 + (int)(rowa & 15)) * 16));
        } else {
          const char* blk = (const char*)p.w + ((rowa >> 4) * (size_t)(p.K / 128) + (size_t)(k >> 7)) * 1152;
          wr[a][t][u] = __builtin_nontemporal_load((const u32x4*)(blk + (g * 16 + (int)(rowa & 15)) * 16));
          sr[a][t][u] = *(const uint32_t*)(blk + 1024 + (int)(rowa & 15) * 4);
          br[a][t][u] = *(const uint32_t*)(blk + 1088 + (int)(rowa & 15) * 4);
        }
      }
    }
  }
  __device__ __forceinline__ void issue_w(int tbi, int k0, int kend) {
#pragma unroll
    for (int u = 0; u < UK; ++u) issue_u(u, tbi, k0, kend);
  }
  // the first batch's weights: independent of any activation, so a caller may issue them early
  __device__ __forceinline__ void prefetch_weights() { issue_w(0, 0, ntiles > 0 ? klen0 : 0); }


  template <bool COH>
  __device__ __forceinline__ void load_x(int kbase, int klen) {
    const int n8 = klen / 8;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int k8 = tid + j * NT;
        xv[m][j] = u32x4{0u, 0u, 0u, 0u};
        if (m < p.M && k8 < n8) {
          const AT* src = x + (size_t)m * p.ldx + kbase + k8 * 8;
          if constexpr (COH) xv[m][j] = load16_agent(src);
          else xv[m][j] = *(const u32x4*)src;
        }
      }
  }

  // xv -> (RMSNorm) -> MFMA A-fragments in LDS.  The normalisation is written on dword pairs (two floats per v_pk_mul_f32,
  // one v_cvt_pk per rounding) and the norm weights are unpacked once for the MB rows: this pass is ~450 VALU instructions
  // per thread in front of the first MFMA of every normalised linear, on every workgroup.
  __device__ __forceinline__ void stage_x(int kbase, int klen) {
    const int n8 = klen / 8;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int k8 = tid + j * NT;
      if (k8 < n8) {
        f32x2 wf[4];
        if (p.pro == PRO_NORM) {
          const u32x4 wv = *(const u32x4*)((const AT*)p.norm_w + kbase + k8 * 8);
          wf[0] = unpack2<AT>(wv.x); wf[1] = unpack2<AT>(wv.y); wf[2] = unpack2<AT>(wv.z); wf[3] = unpack2<AT>(wv.w);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          u32x4 v = xv[m][j];
          float sum = 0.f;
          if (m < p.M) {
            if (p.pro == PRO_NORM) {
              const float rs = rs_sh[m];
              const f32x2 rs2 = {rs, rs};
              uint32_t* d = (uint32_t*)&v;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint32_t xn = pack2<AT>(unpack2<AT>(d[i]) * rs2);      // cast_T(x32 * rsqrt(..))
                d[i] = pack2<AT>(unpack2<AT>(xn) * wf[i]);                  // w * (.)  in T
              }
            }
            if constexpr (Q4) {
              AT* e = (AT*)&v;
#pragma unroll
              for (int i = 0; i < 8; ++i) sum += (float)e[i];
              AT t2[8];                                            // nibble order: 2q <- k+q, 2q+1 <- k+q+4
#pragma unroll
              for (int q = 0; q < 4; ++q) { t2[2 * q] = e[q]; t2[2 * q + 1] = e[q + 4]; }
#pragma unroll
              for (int i = 0; i < 8; ++i) e[i] = t2[i];
            }
          }
          frag_w[frag_slot<Q4>(k8, m, MB)] = v;
          if constexpr (Q4) {
            // 8 consecutive lanes hold the 8 pieces of one 64-wide quantisation group
            sum = lane8_sum(sum);
            if ((k8 & 7) == 0) sx_w[(k8 >> 3) * MB + m] = sum;
          }
        }
      }
    }
  }

  __device__ __forceinline__ void zero_acc() {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int t = 0; t < TB; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- MFMAs of slot u of batch [k0, k0+KS) clipped to kend; fragments are addressed relative to cbase
  __device__ __forceinline__ void mfma_u(int u, int k0, int kend, int cbase) {
    const int kq = k0 + (u * NW + wave) * BK;
    // A slot past kend holds clamped (valid but meaningless) weights: it is skipped (wave-uniform branch; no global
    // load inside).  With MB = 8 the fragment rows 8..15 are the rows 0..7 again (cm = c16 & 7): output rows 8..15
    // come out as copies and are never stored (finish: m < p.M), so no lane is masked -- the selects that zeroed
    // them were a sixth of the int4 kernel's VALU instructions (SQ_INSTS_VALU), and that kernel is VALU-bound.
    if (kq >= kend) return;
    const int kb = (kq - cbase) / BK;
    const int cm = c16 & (MB - 1);
    if constexpr (!Q4) {
      const u32x4 af = frag[(kb * 4 + g) * MB + cm];
#pragma unroll
      for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a][t] = mfma16<AT>(af, wr[a][t][u], acc[a][t]);
    } else {
      u32x4 af[2][2];
      f32x4 sxv[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          af[s][t2] = frag[((((kb * 2 + s) * 2 + t2) * 4 + g) * MB + cm)];
        }
        sxv[s] = *(const f32x4*)&sx[(kb * 2 + s) * MB + ((g * 4) & (MB - 1))];
      }
#pragma unroll
      for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          // lanes g<2 hold quant group A (dwords 0-7 of the 128 block), g>=2 group B.  Swap the
          // upper half's {x,y} with the lower half's {z,w}: afterwards {x,y} = group A and
          // {z,w} = group B in EVERY lane (v_permlane32_swap).
          u32x4 v = wr[a][t][u];
          auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
          const uint32_t dw[4] = {r0[0], r1[0], r0[1], r1[1]};
          const S* sp = (const S*)&sr[a][t][u];
          const S* bp = (const S*)&br[a][t][u];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = mfma16<AT>(af[s][0], unpack_q4<AT>(dw[s * 2 + 0]), d);
            d = mfma16<AT>(af[s][1], unpack_q4<AT>(dw[s * 2 + 1]), d);
            // y += scale * sum((OFFS+q) x) + (bias - OFFS*scale) * sum(x)   per output row m
            const float sc = (float)sp[s], bb = (float)bp[s] - Magic<AT>::offs * sc;
            acc[a][t].x = fmaf(sc, d.x, fmaf(bb, sxv[s].x, acc[a][t].x));
            acc[a][t].y = fmaf(sc, d.y, fmaf(bb, sxv[s].y, acc[a][t].y));
            acc[a][t].z = fmaf(sc, d.z, fmaf(bb, sxv[s].z, acc[a][t].z));
            acc[a][t].w = fmaf(sc, d.w, fmaf(bb, sxv[s].w, acc[a][t].w));
          }
        }
    }
  }

  // ---- cross-wave reduction + epilogue of the TB tiles of batch tbi.  WT: the outputs are read by
  // other workgroups of this same launch -> write-through stores
  template <bool WT>
  __device__ __forceinline__ void finish(int tbi) {
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int tile = tbi * TB + t;
      if (tile >= ntiles) break;                   // uniform
      if (t > 0) __syncthreads();
#pragma unroll
      for (int a = 0; a < NA; ++a) *(f32x4*)&red[((wave * NA + a) * 64 + lane) * 4] = acc[a][t];
      __syncthreads();
      if (tid < 256) {
        const int el = tid & 63, r = tid >> 6;
        const int m = 4 * (el >> 4) + r, n = item_row0(tile) + (el & 15);
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
          y0 += red[((ww * NA + 0) * 64 + el) * 4 + r];
          if constexpr (SWIGLU) y1 += red[((ww * NA + 1) * 64 + el) * 4 + r];
        }
        if (m < p.M) {
          AT* out = (AT*)p.out;
          if constexpr (SWIGLU) {
            const float gt = (float)(AT)y0, up = (float)(AT)y1;
            const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
            const float sl = (float)(AT)(gt * sig);
            store_elem<AT>(&out[(size_t)m * p.ldo + n], (AT)(sl * up), WT);
          } else if (p.epi == EPI_SWIGLU_GU8) {
            // row-interleaved gate|up tile: columns 0..7 are gate, 8..15 the matching up rows (same reduction order as above)
            if ((el & 15) < 8) {
              float yu = 0.f;
#pragma unroll
              for (int ww = 0; ww < NW; ++ww) yu += red[((ww * NA + 0) * 64 + el + 8) * 4 + r];
              const float gt = (float)(AT)y0, up = (float)(AT)yu;
              const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
              const float sl = (float)(AT)(gt * sig);
              store_elem<AT>(&out[(size_t)m * p.ldo + (n >> 4) * 8 + (el & 15)], (AT)(sl * up), WT);
            }
          } else {
            float y = (float)(AT)y0;
            if (p.lora_t != nullptr) {
#pragma unroll
              for (int sl = 0; sl < 2; ++sl) {
                const int r0 = sl ? p.lora_row0_1 : p.lora_row0_0;
                const int ln = sl ? p.lora_n_1 : p.lora_n_0;
                const int rk = sl ? p.lora_rank_1 : p.lora_rank_0;
                const float* lb = sl ? p.lora_b1 : p.lora_b0;
                if (lb != nullptr && n >= r0 && n < r0 + ln) {
                  const float* tt = p.lora_t + (size_t)m * p.lora_t_ld + sl * (p.lora_t_ld / 2);
                  float z = lora_dot(tt, lb + (n - r0), ln, rk);
                  z = (sl ? p.lora_scale_1 : p.lora_scale_0) * z;
                  y = (float)(AT)(y + (float)(AT)z);
                }
              }
            }
            if (p.epi == EPI_STORE) store_elem<AT>(&out[(size_t)m * p.ldo + n], (AT)y, WT);
            else if (p.epi == EPI_STORE_F32) ((float*)p.out)[(size_t)m * p.ldo + n] = y;
            else {
              AT* h = (AT*)p.resid;
              const float h0 = (hpre_ok && tile == 0) ? hpre : (float)h[(size_t)m * p.ldo + n];
              store_elem<AT>(&h[(size_t)m * p.ldo + n], (AT)(h0 + y), WT);
            }
          }
        }
      }
    }
  }

  // with several activation chunks the NEXT chunk's x is fetched into registers while the current
  // chunk's MFMAs run, so that re-staging costs two barriers but no exposed L2 round trip
  template <bool COH>
  __device__ __forceinline__ void prefetch_next_x(int c, int tbi) {
    if (nchunks == 1) return;
    const int nc = (c + 1) % nchunks;
    if (nc == 0 && tbi + 1 >= nbatch) return;
    load_x<COH>(nc * p.kc, min(p.kc, p.K - nc * p.kc));
  }

  // The whole phase.  PREFETCHED: prefetch_weights() has already been called.  COH: x was written by
  // other workgroups of this launch.  WT: the outputs will be read by other workgroups of this launch.
  // Every __syncthreads below is reached by the whole workgroup also when it owns no tile.
  template <bool PREFETCHED, bool COH, bool WT>
  __device__ __forceinline__ void run() {
    // ================= prologue: activations first (older in the vmcnt queue), then weights
    PH_STAMP(0);
    if (ntiles > 0) load_x<COH>(0, klen0);
    if constexpr (!PREFETCHED) prefetch_weights();
    if constexpr (!COH && !SWIGLU) {
      // residual epilogue: the h values of the first tile now -- only this workgroup writes them in this launch -- instead of
      // as a dependent load after the cross-wave reduction, at the very end of the chain (o_proj / down_proj: one tile per
      // workgroup).  Straight-line: the pointer is selected, the load is not under a branch.
      const bool res = p.epi == EPI_RESID && ntiles > 0;
      const AT* hp = res ? (const AT*)p.resid : (const AT*)p.x;
      const int el = tid & 63, r = (tid >> 6) & 3;
      const int mm = min(4 * (el >> 4) + r, p.M - 1);
      const size_t off = res ? (size_t)mm * p.ldo + item_row0(0) + (el & 15) : 0;
      hpre = (float)hp[off];
      hpre_ok = res;
    }
    if (p.pro == PRO_NORM) {
      float ss[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) ss[m] = 0.f;
      if (ntiles > 0) {
        if (nchunks == 1) {
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            f32x2 s2 = {0.f, 0.f};                     // dword pairs: two unpacks + one v_pk_fma_f32 per two elements
#pragma unroll
            for (int j = 0; j < J; ++j) {
              const uint32_t* d = (const uint32_t*)&xv[m][j];
#pragma unroll
              for (int i = 0; i < 4; ++i) { const f32x2 f = unpack2<AT>(d[i]); s2 = f * f + s2; }
            }
            ss[m] = s2.x + s2.y;
          }
        } else {
          for (int k = tid * 8; k < p.K; k += NT * 8) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              if (m < p.M) {
                u32x4 v;
                if constexpr (COH) v = load16_agent(x + (size_t)m * p.ldx + k);
                else v = *(const u32x4*)(x + (size_t)m * p.ldx + k);
                const AT* e = (const AT*)&v;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float f = (float)e[i]; ss[m] = fmaf(f, f, ss[m]); }
              }
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const float v = wave_sum(ss[m]);
        if (lane == 0) red2[wave * 16 + m] = v;
      }
      __syncthreads();
      if (tid < MB) {
        float v = 0.f;
        for (int ww = 0; ww < NW; ++ww) v += red2[ww * 16 + tid];
        rs_sh[tid] = 1.0f / sqrtf(v / (float)p.K + p.eps);
      }
      __syncthreads();
    }
    PH_STAMP(1);
    if (ntiles > 0) stage_x(0, klen0);
    __syncthreads();
    PH_STAMP(2);
    if (ntiles <= 0) return;
    int staged = 0;
    prefetch_next_x<COH>(0, 0);

    // ================= this workgroup's tile batches
    int cur = 0;                                // nbuf == 2: the buffer that holds the chunk being multiplied
    for (int tb = 0; tb < nbatch; ++tb) {
      zero_acc();
      for (int c = 0; c < nchunks; ++c) {
        const int cbase = c * p.kc, kend = min(p.K, cbase + p.kc);
        if ((!DBUF || nbuf == 1) && staged != c) {
          __syncthreads();                      // every wave is done reading the old fragments
          stage_x(cbase, kend - cbase);         // from the registers prefetched one chunk ago
          __syncthreads();
          staged = c;
          prefetch_next_x<COH>(c, tb);
        }
        // nbuf == 2: the chunk that follows this one (possibly chunk 0 of the next tile) is staged into the
        // other buffer in the middle of this chunk's first batch -- its x has been in registers since the
        // chunk began -- so the weight stream never waits for a staging pass
        const int nc = (c + 1) % nchunks;
        const bool stage_next = DBUF && nbuf == 2 && !(nc == 0 && tb + 1 >= nbatch);
        // (selects, not indexed loads: a dynamically indexed member array would push the whole struct to scratch)
        frag_w = cur ? frag_b[0] : frag_b[1]; sx_w = cur ? sx_b[0] : sx_b[1];
        for (int k0 = cbase; k0 < kend; k0 += KS) {
          // the batch that follows this one in this workgroup's sequence
          int ntb = tb, nk0 = k0 + KS, nkend = kend;
          if (nk0 >= kend) {
            if (c + 1 < nchunks) { nk0 = cbase + p.kc; nkend = min(p.K, cbase + 2 * p.kc); }
            else if (tb + 1 < nbatch) { ntb = tb + 1; nk0 = 0; nkend = klen0; }
            else { nk0 = 0; nkend = 0; }                         // nothing left: issue_u becomes a no-op
          }
          // rolling prefetch: as soon as the MFMAs of slot u have consumed its registers, the same
          // registers are re-loaded with slot u of the NEXT batch, so every wave keeps UK loads in
          // flight at all times.  The scheduling barriers pin this order (hoisting the loads would
          // double the register footprint, sinking them would drain the memory pipe).
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            mfma_u(u, k0, kend, cbase);
            __builtin_amdgcn_sched_barrier(0);
            issue_u(u, ntb, nk0, nkend);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DBUF) {
              if (u == UK / 2 && k0 == cbase && stage_next) stage_x(nc * p.kc, min(p.kc, p.K - nc * p.kc));
            }
          }
        }
        if (DBUF && stage_next) {
          __syncthreads();                      // the next chunk's fragments are complete; this chunk's are free
          cur ^= 1;
          frag = cur ? frag_b[1] : frag_b[0]; sx = cur ? sx_b[1] : sx_b[0];
          prefetch_next_x<COH>(nc, nc == 0 ? tb + 1 : tb);
        }
      }
      if (tb == nbatch - 1) { PH_STAMP(3); PH_STAMP_WAVE(8); }
      finish<WT>(tb);
      __syncthreads();                          // `red` / fragments are reused
    }
    PH_STAMP(7);
  }
};

// ---- in-launch seam: every workgroup calls seam_arrive once it has published its outputs (write-through stores,
// drained), may issue loads that do not depend on the other workgroups, then seam_wait.  Bounded: a workgroup
// that gives up sets *error and carries on; the engine fails the step that contains the launch (see SeamParams).
__device__ __forceinline__ void seam_arrive(const SeamParams& seam) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(seam.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void seam_wait(const SeamParams& seam) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(seam.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seam.target) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > seam.spin_limit) {
        __hip_atomic_store(seam.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

}  // namespace gemv

// host side (gemv_mfma.hip)
gemv::MfmaParams gemv_make_params(const LinearW& W, const GemvCall& c);
int gemv_cu_count();

}  // namespace mi
