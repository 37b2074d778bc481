// gemv_mfma.hip -- the hot decode kernel: weight-streaming skinny GEMM for M <= 16 rows on the
// CDNA4 matrix cores (v_mfma_f32_16x16x32_{bf16,f16}).
//
// Same operator as gemv_v1.hip (nn.Linear / nn.QuantizedLinear call sites llama.py:64-67,93,
// 143,160-165,250-252; qwen3.py:37-40,63,115 with RMSNorm / residual / SwiGLU fused), for the
// configurations BASELINE.json is quoted on: 16-bit activations with dense 16-bit weights or
// MLX-affine int4 (group 64) weights, both in the tile-major layout of repack.hip.  Why MFMA at
// M = 8: the op is HBM-bound (8 MACs per weight), and on the VALU the int4 unpack + 8 FMAs per
// weight would need ~80 % of the issue slots at the HBM rate; one MFMA per KiB of weights keeps the
// vector pipe nearly idle.
//
// Orientation: D[m][n] = sum_k A[m][k] B[k][n] with A = activations (from LDS), B = W^T
// (16 B per lane straight from HBM, non-temporal).  Lane l supplies W[n0 + (l&15)][k0 +
// 8(l>>4) .. +8] and receives y[m = 4(l>>4)+r][n0 + (l&15)], r = 0..3 -- so a quantisation
// scale (per weight row, per 64-k group) is a per-lane scalar.
//
// Workgroup = 512 threads = 8 waves, one per CU, grid = min(tiles, CUs); 16-row tiles of W are dealt
// round-robin.  The eight waves split K round-robin in 32-wide (dense) or 128-wide (int4) blocks, so
// one step of the workgroup reads contiguous KiB blocks; partial tiles are summed through LDS.
// Activations are staged (RMSNorm applied) as 16-bit MFMA A-fragments.
//
// One GEMV is a `Phase`.  gemv_mfma_kernel runs one phase; gemv_pair_kernel runs two dependent
// phases (o_proj -> gate|up, down_proj -> next layer's q|k|v) in ONE launch: after its last tile of
// phase A a workgroup publishes its outputs write-through (sc1), arrives on a device counter, issues
// the first weight loads of phase B -- they do not depend on phase A -- and only then polls the
// counter; phase B's activations are read with agent-scope (sc1) loads.  The seam costs a counter
// round trip that overlaps the weight prefetch, instead of a kernel boundary + a cold prologue.
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace mi {

namespace {

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
};

// in-launch seam of gemv_pair_kernel: every workgroup adds 1 to *counter after phase A; phase B starts
// when the counter has reached `target`.  A workgroup that polls `spin_limit` times without seeing it
// sets *error and carries on (wrong results, reported by the host; never a hang).
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
    r.x = ((v << 3) & 0x00780078u) | 0x41804180u;
    r.y = ((v >> 1) & 0x00780078u) | 0x41804180u;
    r.z = ((v >> 5) & 0x00780078u) | 0x41804180u;
    r.w = ((v >> 9) & 0x00780078u) | 0x41804180u;
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
// (dense weights only: the int4 variants have no registers to spare for the in-loop staging pass)
__host__ __device__ constexpr int phase_nbuf(int K, int kc, int MB, bool q4) { return (K > kc && MB == 8 && !q4) ? 2 : 1; }

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
    G = gridDim.x; w = blockIdx.x;
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

  // xv -> (RMSNorm) -> MFMA A-fragments in LDS
  __device__ __forceinline__ void stage_x(int kbase, int klen) {
    const int n8 = klen / 8;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int k8 = tid + j * NT;
        if (k8 < n8) {
          u32x4 v = xv[m][j];
          AT* e = (AT*)&v;
          float sum = 0.f;
          if (m < p.M) {
            if (p.pro == PRO_NORM) {
              const u32x4 wv = *(const u32x4*)((const AT*)p.norm_w + kbase + k8 * 8);
              const AT* we = (const AT*)&wv;
              const float rs = rs_sh[m];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const AT xn = (AT)((float)e[i] * rs);              // cast_T(x32 * rsqrt(..))
                e[i] = (AT)((float)xn * (float)we[i]);            // w * (.)  in T
              }
            }
            if constexpr (Q4) {
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

  __device__ __forceinline__ void zero_acc() {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int t = 0; t < TB; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- MFMAs of slot u of batch [k0, k0+KS) clipped to kend; fragments are addressed relative to cbase
  __device__ __forceinline__ void mfma_u(int u, int k0, int kend, int cbase) {
    const int kq = k0 + (u * NW + wave) * BK;
    const bool valid = kq < kend;                     // wave-uniform; an invalid slot multiplies zeros
    const int kb = valid ? (kq - cbase) / BK : 0;
    const bool lane_on = valid && (MB == 16 || c16 < MB);
    const int cm = c16 & (MB - 1);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    if constexpr (!Q4) {
      u32x4 af = frag[(kb * 4 + g) * MB + cm];
      af = lane_on ? af : zero4;
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
          af[s][t2] = lane_on ? af[s][t2] : zero4;
        }
        sxv[s] = *(const f32x4*)&sx[(kb * 2 + s) * MB + ((g * 4) & (MB - 1))];
        if (!(valid && (MB == 16 || g * 4 < MB))) sxv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                  float z = 0.f;
                  for (int j = 0; j < rk; ++j) z = fmaf(tt[j], lb[(size_t)j * ln + (n - r0)], z);
                  z = (sl ? p.lora_scale_1 : p.lora_scale_0) * z;
                  y = (float)(AT)(y + (float)(AT)z);
                }
              }
            }
            if (p.epi == EPI_STORE) store_elem<AT>(&out[(size_t)m * p.ldo + n], (AT)y, WT);
            else if (p.epi == EPI_STORE_F32) ((float*)p.out)[(size_t)m * p.ldo + n] = y;
            else {
              AT* h = (AT*)p.resid;
              store_elem<AT>(&h[(size_t)m * p.ldo + n], (AT)((float)h[(size_t)m * p.ldo + n] + y), WT);
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
    if (ntiles > 0) load_x<COH>(0, klen0);
    if constexpr (!PREFETCHED) prefetch_weights();
    if (p.pro == PRO_NORM) {
      float ss[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) ss[m] = 0.f;
      if (ntiles > 0) {
        if (nchunks == 1) {
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < J; ++j) {
              const AT* e = (const AT*)&xv[m][j];
#pragma unroll
              for (int i = 0; i < 8; ++i) { const float f = (float)e[i]; ss[m] = fmaf(f, f, ss[m]); }
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
    if (ntiles > 0) stage_x(0, klen0);
    __syncthreads();
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
      finish<WT>(tb);
      __syncthreads();                          // `red` / fragments are reused
    }
  }
};

template <typename AT, bool Q4, int MB, bool SWIGLU, int NW, int J, bool DB>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_kernel(MfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Phase<AT, Q4, MB, SWIGLU, NW, J, DB> ph(p, smem_raw);
  if (ph.ntiles <= 0) return;
  ph.template run<false, false, false>();
}

// Two dependent GEMVs in one launch: A (plain epilogue) then B (SwiGLU or plain), see the file header.
// grid = number of CUs, every workgroup resident at once (one 8-wave workgroup per CU).
template <typename AT, bool Q4, int MB, bool SWB, int NW, int JA, int JB>
__global__ __launch_bounds__(NW * 64) void gemv_pair_kernel(MfmaParams pa, MfmaParams pb, SeamParams seam) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  {
    Phase<AT, Q4, MB, false, NW, JA> A(pa, smem_raw);
    A.template run<false, false, true>();
  }
  // ---- seam: publish, arrive, prefetch B's first weights, wait
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(seam.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  Phase<AT, Q4, MB, SWB, NW, JB> B(pb, smem_raw);
  B.prefetch_weights();
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
  B.template run<true, true, false>();
}

static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

int cu_count() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

template <typename AT, bool Q4, int MB, bool SWIGLU, int NW, int J, bool DB>
int launch_j(const MfmaParams& p, hipStream_t st) {
  auto kern = gemv_mfma_kernel<AT, Q4, MB, SWIGLU, NW, J, DB>;
  constexpr int NA = SWIGLU ? 2 : 1;
  const size_t lds = phase_lds_bytes<NW, NA, Q4>(p.kc, MB, DB ? 2 : 1);
  MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nwg = std::min(p.N / 16, cu_count());        // one workgroup per CU; items are dealt in-kernel
  if (g_ev_start != nullptr)   // measurement: dispatch-level begin/end timestamps of THIS kernel
    hipExtLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, st, g_ev_start, g_ev_stop, 0, p);
  else
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, st, p);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename AT, bool Q4, int MB, bool SWIGLU, int NW>
int launch_one(const MfmaParams& p, hipStream_t st) {
  if constexpr (!Q4 && MB == 8) {
    if (phase_nbuf(p.K, p.kc, MB, Q4) == 2)      // K spans several activation chunks: the double-buffered instantiation
      return p.kc <= 8 * NW * 64 ? launch_j<AT, Q4, MB, SWIGLU, NW, 1, true>(p, st) : launch_j<AT, Q4, MB, SWIGLU, NW, 2, true>(p, st);
  }
  if (p.kc <= 8 * NW * 64) return launch_j<AT, Q4, MB, SWIGLU, NW, 1, false>(p, st);
  return launch_j<AT, Q4, MB, SWIGLU, NW, 2, false>(p, st);
}

template <typename AT>
int launch_at(bool q4, const MfmaParams& p, hipStream_t st) {
  const bool sw = p.epi == EPI_SWIGLU;
  const bool m8 = p.M <= 8;
  if (!q4) {
    if (m8) return sw ? launch_one<AT, false, 8, true, 8>(p, st) : launch_one<AT, false, 8, false, 8>(p, st);
    return sw ? launch_one<AT, false, 16, true, 8>(p, st) : launch_one<AT, false, 16, false, 8>(p, st);
  }
  if (m8) return sw ? launch_one<AT, true, 8, true, 8>(p, st) : launch_one<AT, true, 8, false, 8>(p, st);
  return sw ? launch_one<AT, true, 16, true, 8>(p, st) : launch_one<AT, true, 16, false, 8>(p, st);
}

template <typename AT, bool Q4, int MB, bool SWB, int JA, int JB>
int launch_pair_j(const MfmaParams& pa, const MfmaParams& pb, const SeamParams& seam, hipStream_t st) {
  constexpr int NW = 8;
  auto kern = gemv_pair_kernel<AT, Q4, MB, SWB, NW, JA, JB>;
  const size_t lds = std::max(phase_lds_bytes<NW, 1, Q4>(pa.kc, MB, 1), phase_lds_bytes<NW, SWB ? 2 : 1, Q4>(pb.kc, MB, 1));
  MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (g_ev_start != nullptr)
    hipExtLaunchKernelGGL(kern, dim3(cu_count()), dim3(NW * 64), lds, st, g_ev_start, g_ev_stop, 0, pa, pb, seam);
  else
    hipLaunchKernelGGL(kern, dim3(cu_count()), dim3(NW * 64), lds, st, pa, pb, seam);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename AT, bool Q4, int MB, bool SWB>
int launch_pair_mb(const MfmaParams& pa, const MfmaParams& pb, const SeamParams& seam, hipStream_t st) {
  const bool ja2 = pa.kc > 4096, jb2 = pb.kc > 4096;
  if (!ja2 && !jb2) return launch_pair_j<AT, Q4, MB, SWB, 1, 1>(pa, pb, seam, st);
  if (ja2 && jb2) return launch_pair_j<AT, Q4, MB, SWB, 2, 2>(pa, pb, seam, st);
  if (ja2) return launch_pair_j<AT, Q4, MB, SWB, 2, 1>(pa, pb, seam, st);
  return launch_pair_j<AT, Q4, MB, SWB, 1, 2>(pa, pb, seam, st);
}

template <typename AT>
int launch_pair_at(bool q4, const MfmaParams& pa, const MfmaParams& pb, const SeamParams& seam, hipStream_t st) {
  const bool sw = pb.epi == EPI_SWIGLU;
  if (!q4) return sw ? launch_pair_mb<AT, false, 8, true>(pa, pb, seam, st) : launch_pair_mb<AT, false, 8, false>(pa, pb, seam, st);
  return sw ? launch_pair_mb<AT, true, 8, true>(pa, pb, seam, st) : launch_pair_mb<AT, true, 8, false>(pa, pb, seam, st);
}

MfmaParams make_params(const LinearW& W, const GemvCall& c) {
  MfmaParams p{};
  p.x = c.x; p.ldx = c.ldx; p.M = c.M; p.pro = c.pro; p.norm_w = c.norm_w; p.eps = c.eps;
  p.w = W.w; p.scales = W.scales; p.biases = W.biases; p.N = (c.epi == EPI_SWIGLU) ? c.pair_offset : W.N; p.K = W.K;
  p.epi = c.epi; p.out = c.out; p.ldo = c.ldo; p.resid = c.resid; p.pair_offset = c.pair_offset;
  // activation chunk held in LDS as fragments: whole K up to 6144 (M <= 8) / 4096 (M <= 16),
  // else chunks of 4096 (a multiple of every batch span and of the int4 block; 8192 was measured: no gain)
  const int kc_max = c.M <= 8 ? 6144 : 4096;
  p.kc = (W.K <= kc_max) ? W.K : 4096;
  p.layout = W.layout;
  p.lora_t = c.lora_t; p.lora_t_ld = c.lora_t_ld;
  p.lora_b0 = W.lora_b[0]; p.lora_b1 = W.lora_b[1];
  p.lora_row0_0 = W.lora_row0[0]; p.lora_n_0 = W.lora_n[0]; p.lora_rank_0 = W.lora_rank[0]; p.lora_scale_0 = W.lora_scale[0];
  p.lora_row0_1 = W.lora_row0[1]; p.lora_n_1 = W.lora_n[1]; p.lora_rank_1 = W.lora_rank[1]; p.lora_scale_1 = W.lora_scale[1];
  return p;
}

}  // namespace

// true when the MFMA path can run this call
bool gemv_mfma_supported(const LinearW& W, const GemvCall& c) {
  if (c.force_v1) return false;
  if (W.layout != 1) return false;        // the MFMA kernel reads the tile-major layout only (repack.hip)
  if (c.rnd != RND_NONE) return false;
  if (c.M < 1 || c.M > 16) return false;
  if (c.act != MI_BF16 && c.act != MI_F16) return false;
  const bool q4 = (W.wk == WK_Q4_BF16 && c.act == MI_BF16) || (W.wk == WK_Q4_F16 && c.act == MI_F16);
  const bool dense = (W.wk == WK_BF16 && c.act == MI_BF16) || (W.wk == WK_F16 && c.act == MI_F16);
  if (!q4 && !dense) return false;
  if (q4 && (W.group != 64 || W.K % 128 != 0)) return false;
  if (dense && W.K % 32 != 0) return false;
  if (c.ldx % 8 != 0) return false;
  const int n = (c.epi == EPI_SWIGLU) ? c.pair_offset : W.N;
  if (n % 16 != 0) return false;
  return true;
}

int launch_gemv_mfma(const LinearW& W, const GemvCall& c, hipStream_t st) {
  const bool q4 = wk_is_quant(W.wk);
  const MfmaParams p = make_params(W, c);
  g_ev_start = (hipEvent_t)c.ev_start; g_ev_stop = (hipEvent_t)c.ev_stop;
  const int rc = (c.act == MI_BF16) ? launch_at<bf16>(q4, p, st) : launch_at<f16>(q4, p, st);
  g_ev_start = g_ev_stop = nullptr;
  return rc;
}

// Two dependent GEMVs (B reads what A wrote) as one launch; see gemv_pair_kernel.
bool gemv_pair_supported(const LinearW& WA, const GemvCall& a, const LinearW& WB, const GemvCall& b) {
  if (!gemv_mfma_supported(WA, a) || !gemv_mfma_supported(WB, b)) return false;
  if (a.M > 8 || b.M != a.M || a.act != b.act) return false;
  if (wk_is_quant(WA.wk) != wk_is_quant(WB.wk)) return false;
  if (a.epi != EPI_STORE && a.epi != EPI_RESID) return false;          // A's outputs are 16-bit activations
  if (b.epi == EPI_STORE_F32) return false;
  if (a.lora_t != nullptr || b.lora_t != nullptr) return false;        // the LoRA down-projection is a launch of its own
  return true;
}

int launch_gemv_pair(const LinearW& WA, const GemvCall& a, const LinearW& WB, const GemvCall& b, const GemvSeam& s,
                     hipStream_t st) {
  const bool q4 = wk_is_quant(WA.wk);
  const MfmaParams pa = make_params(WA, a), pb = make_params(WB, b);
  SeamParams seam{s.counter, s.base + (unsigned)cu_count(), 1u << 20, s.error};
  g_ev_start = (hipEvent_t)a.ev_start; g_ev_stop = (hipEvent_t)a.ev_stop;
  const int rc = (a.act == MI_BF16) ? launch_pair_at<bf16>(q4, pa, pb, seam, st) : launch_pair_at<f16>(q4, pa, pb, seam, st);
  g_ev_start = g_ev_stop = nullptr;
  return rc;
}

int gemv_pair_grid() { return cu_count(); }

}  // namespace mi
