// gemm_skinny.hip -- weight-streaming GEMM for 9..128 activation rows (int4 / int8 weights: 1..128): the decode step of a larger batch
// (BASELINE configs 4 / 5 put 32 / 64 sequences in one batch) and short prefills.
//
// Same operator as gemv_mfma.hip (nn.Linear / nn.QuantizedLinear call sites llama.py:64-67,93,143,160-165,
// 250-252; qwen3.py:37-40,63,115 with residual / SwiGLU fused), same tile-major weights (repack.hip), same
// orientation: D[m][n] = sum_k A[m][k] B[k][n], A = activations from LDS, B = W^T straight from HBM.
//
// Why another kernel.  gemv_mfma.hip splits K over the eight waves of a workgroup, so every workgroup stages
// ALL of x in LDS: at 64 rows x 4096 that is 512 KiB per 128 KiB of weights.  The tile GEMM of the prefill
// (gemm_prefill.hip) has N/128 workgroups and is tuned for thousands of rows: at 32 rows it ran the
// Mistral-7B step in 17.5 ms where the weights alone stream in 2.3 ms.  Here the op stays what it is at this
// size -- HBM-bound on the weights -- and is organised around one read of W:
//
//   * A workgroup is 8 waves; wave w owns the 16-row tile 8 j + w of W (gate AND up tile for SwiGLU) and
//     streams it with a rolling prefetch (8 KiB-loads in flight per wave; int4: 4 x 1152 B).
//   * All eight waves walk the same k range in 256-wide chunks, so one LDS copy of the activation chunk
//     (MFMA A-fragments, 16 * MT rows x 256 k, double-buffered; the next chunk is fetched into registers at
//     the top of a chunk and written in its middle) serves eight tiles: x moves L2 -> LDS at MT/8 of the
//     weight rate instead of MT x.
//   * N / 128 workgroups cannot fill 256 CUs (o_proj: 32), so K is also split over `ksplit` workgroups.
//     Partial tiles go to a float32 workspace; the LAST workgroup of a tile group to arrive (one agent-scope
//     counter per group, no spinning) adds the ksplit partials in slice order -- deterministic -- and runs the
//     epilogue.  ksplit comes from a small cost model on the host (skinny_plan).
//   * Two workgroups per CU (<= 66 KiB LDS, <= 128 VGPRs) hide each other's barriers and prologues.
//
// RMSNorm is not fused here: the caller runs rmsnorm_rows first (as for the prefill GEMM) -- every workgroup
// would otherwise need a full pass over its rows before the first MFMA.
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "gemv_phase.h"

namespace mi {

namespace {

using namespace gemv;

constexpr int SK_NW = 8;      // waves (= tiles) per workgroup
constexpr int SK_KC = 256;    // k per activation chunk
constexpr int SQ_PARTS_MAX = 64;   // norm hand-over: tile groups of the producing linear (N <= 8192)

// 8 int8 codes (two packed dwords) -> 8 16-bit floats: 0..255 are exact in bf16 and f16
template <typename T>
__device__ __forceinline__ u32x4 unpack_q8(uint32_t lo, uint32_t hi) {
  T e[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    e[j] = (T)(float)((lo >> (8 * j)) & 255u);
    e[4 + j] = (T)(float)((hi >> (8 * j)) & 255u);
  }
  return *(const u32x4*)e;
}

struct SkinnyParams {
  const void* x; int ldx; int M;
  const void* w; int N, K;
  int epi; void* out; int ldo; void* resid; int pair_offset;
  // RMSNorm hand-over between two launches of the 16-row instantiation (see "norm hand-over" in the kernel):
  float* sq_out;                   // producer (residual epilogue): [tile groups][sq_ld] sums of h^2 over the group's 128 columns
  const float* sq_in; int sq_parts;   // consumer: the producer's partial sums and their count
  int sq_ld;                       // rows of the table, padded: slabs x rows per workgroup (16 / 32 / 64 rows per workgroup only)
  const void* norm_w; float eps;   //           x -> w * T(x * rsqrt(mean(x^2) + eps)) while staging
  int rnd;                         // float32 activations: logical rounding of the outputs (RND_*)
  // float32 activations, RMSNorm in front (X32 instantiations only): y = rs[m] * sum_k (x[m][k] w_norm[k]) W[n][k] -- the
  // row scale rs = rsqrt(mean(x^2) + eps) is a per-row factor of the whole dot product, so it is applied in the EPILOGUE
  // and nothing at the head of the launch waits for the row statistics: x is multiplied by w_norm while it is staged,
  // its squares are summed on the way, the K slices' sums meet with the partial tiles.  In float32 this differs from
  // w * (x * rs) by one rounding per element (~6e-8 relative); rounded (layer-0) calls keep the separate norm launch.
  int defer_norm;
  float* sqws;                     // [tile groups][ksplit][16 MT] row sums of squares of the K slices (ksplit > 1)
  int kx;            // columns of x (= K, or K / 2 when W is an f16 model's [hi | lo] copy: x is walked twice, k mod kx)
  int ntiles;        // 16-row tiles (tile pairs for SwiGLU)
  int ksplit;        // workgroups per tile group along K
  // Row slabs (int4 above 32 rows): the rows are cut into `nslab` slabs of 16 MT rows and every (tile group, K slice) unit is
  // run once per slab by its own workgroup -- the 32-row instantiation keeps two workgroups (16 waves) on a CU where the
  // 64-row one fits once, and moves more weight bytes per CU and second than the 64-row loop does per row.  The slabs of a
  // unit stream the same weights; they sit 8 block ids apart, i.e. on one XCD under round-robin placement (speed only:
  // the second reader then finds the lines in that XCD's L2), so HBM sees the matrix about once.
  int nslab, nunits;   // nunits = tile groups x ksplit (the grid is padded to a multiple of 8 units)
  int nchunks;       // ceil(K / 256)
  float* ws;         // [ksplit][ntiles][NA][MT][64 lanes][4] partial accumulators (ksplit > 1)
  unsigned* ctr;     // [tile groups] arrival counters, zero between launches
  const float* lora_t; int lora_t_ld;
  const float* lora_b0; const float* lora_b1;
  int lora_row0_0, lora_n_0, lora_rank_0; float lora_scale_0;
  int lora_row0_1, lora_n_1, lora_rank_1; float lora_scale_1;
#ifdef MI_SK_TRACE
  unsigned long long* trace;       // debug build: [workgroup][8] wall-clock stamps of this launch (tools/debug/skinny_trace.py)
#endif
};

#ifdef MI_SK_TRACE
#define SK_STAMP(i) do { if (tid == 0) p.trace[(size_t)(blockIdx.x & 1023) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

// QB: 0 = dense 16-bit weights, 4 / 8 = MLX-affine int4 / int8 codes (group 64)
// X32: float32 activations (the PagedKVCache mode of a bf16 model; dense, int4 and int8 weights): x is split exactly into three 16-bit
// terms, x = hi + mid + lo, staged as three fragment images and multiplied by three MFMAs per weight fragment -- every
// product is exact in the float32 accumulator, so the result is a float32 dot product in another summation order.
// Outputs stay float32 with the run-time logical rounding `rnd` (layer 0 of that mode still rounds like the model).
template <typename AT, int QB, int MT, bool SWIGLU, bool X32 = false>
__global__ __launch_bounds__(SK_NW * 64, (QB == 4 || (QB == 8 && MT >= 3) || MT >= 5 || (X32 && QB == 8 && MT == 2)) ? 2 : 4) void skinny_kernel(const SkinnyParams p) {
  constexpr bool Q4 = QB == 4, Q8 = QB == 8, QUANT = QB != 0;
  static_assert(!X32 || MT <= 2, "float32 activations: 16- and 32-row instantiations");
  using XT = typename std::conditional<X32, float, AT>::type;
  constexpr int NIMG = X32 ? 3 : 1;
  constexpr int MB = 16 * MT, NA = SWIGLU ? 2 : 1;
  constexpr int UK = (Q4 ? 4 : 8) / NA;     // weight loads in flight per wave and stream (8 / 4 per wave in all; int4 with 8 / 16 at <= 32 rows measured 4 / 15 % slower: occupancy)
  constexpr int UPC = Q4 ? 2 : Q8 ? 4 : 8;  // loads per chunk (a load covers 128 / 64 / 32 k)
  constexpr int CPI = UK > UPC ? UK / UPC : 1;   // chunks per trip of the loop body (the slot ring has UK entries)
  constexpr int UB = Q4 ? 1152 : Q8 ? 1088 : 1024;   // bytes of one tile-major block
  constexpr int FRAG = MB * 512;            // fragment bytes per buffer (MB x 256 x 2)
  constexpr int BUF = NIMG * FRAG + (QUANT ? MB * 16 : 0);   // + sum(x) per (64-group, row) for the quantisation bias term
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int last_sh;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g = lane >> 4;
#ifdef MI_SK_TRACE_ENTRY
  if (lane == 0) p.trace[(size_t)(blockIdx.x & 1023) * 16 + 8 + wave] = wall_clock64();     // every wave: entry
#endif
  int bid = blockIdx.x, slab = 0;
  if (p.nslab > 1) {                                       // (uniform) blocks [16 j, 16 j + 8) = slab 0 of units 8 j .. 8 j + 7, the next 8 = slab 1, ...
    const int per = 8 * p.nslab, blk = bid / per, rem = bid - blk * per;
    slab = rem >> 3;
    bid = blk * 8 + (rem & 7);
    if (bid >= p.nunits) return;                           // grid padding: the whole workgroup leaves before any barrier
  }
  const int row0 = slab * MB;                              // first row of this workgroup's slab
  const int Mloc = min(p.M - row0, MB);                    // its rows (host: nslab = ceil(M / MB), so >= 1)
  const int grp = bid / p.ksplit, s = bid - grp * p.ksplit;
  const int tile_raw = grp * SK_NW + wave;
  const bool valid = tile_raw < p.ntiles;
  const int tile = valid ? tile_raw : p.ntiles - 1;        // a spare wave re-streams the last tile and stores nothing
  const int c0 = (s * p.nchunks) / p.ksplit, c1 = ((s + 1) * p.nchunks) / p.ksplit;
  const int nunits = p.K / (SK_KC / UPC);
  const int u_begin = c0 * UPC, u_end = min(c1 * UPC, nunits);

  const char* wb[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) wb[a] = (const char*)p.w + (size_t)(tile + a * (p.pair_offset >> 4)) * nunits * UB;

  u32x4 wr[NA][UK];
  uint32_t sr[NA][QUANT ? UK : 1], br[NA][QUANT ? UK : 1];
  f32x4 acc[NA][MT];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // STRAIGHT-LINE (see gemv_phase.h issue_u): a unit past the slice re-loads the slice's first block
  auto issue = [&](int slot, int unit) {
    const int uc = unit < u_end ? unit : u_begin;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const char* blk = wb[a] + (size_t)uc * UB;
      wr[a][slot] = __builtin_nontemporal_load((const u32x4*)(blk + lane * 16));
      if constexpr (Q4) {
        sr[a][slot] = *(const uint32_t*)(blk + 1024 + c16 * 4);
        br[a][slot] = *(const uint32_t*)(blk + 1088 + c16 * 4);
      } else if constexpr (Q8) {
        sr[a][slot] = *(const uint16_t*)(blk + 1024 + c16 * 2);
        br[a][slot] = *(const uint16_t*)(blk + 1056 + c16 * 2);
      }
    }
  };

  // ---- activation staging: piece q = (row m, 8 k) of a chunk; 8 consecutive lanes read 128 contiguous bytes
  // of a row.  Piece P of the chunk (dense: k/8; int4: the nibble-order permutation of gemv_phase.h) lands at
  // slot P * MB + (m ^ ((P & 7) << 1)): writers (8 P x 2 m per 16 lanes) and readers (16 m of one P) both
  // touch 16 different 16-byte columns.
  const XT* xrow[MT];
  int xk[MT], woff[MT], sxoff[MT], xmi[MT];
  bool xm[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int q = tid + i * (SK_NW * 64);
    const int k8lo = q & 7, r = q >> 3, m = r % MB, k8l = (r / MB) * 8 + k8lo;
    int P = k8l;
    if constexpr (Q4) {
      const int kb = k8l >> 4, sg = (k8l >> 3) & 1, t = k8lo & 1, e = k8lo - t;
      const int gg = (e == 0) ? 0 : (e == 4) ? 1 : (e == 2) ? 2 : 3;
      P = ((kb * 2 + sg) * 2 + t) * 4 + gg;
    } else if constexpr (Q8) {
      // a lane of the int8 block holds 16 consecutive k: its first 8 feed the first MFMA of the 64-block, the
      // other 8 the second (the MFMA k index is a free labelling) -> piece (g, t) of the block = k 16 g + 8 t
      const int kb = k8l >> 3, gg = (k8lo >> 1), t = k8lo & 1;
      P = (kb * 2 + t) * 4 + gg;
    }
    woff[i] = (P * MB + (m ^ ((P & 7) << 1))) * 16;
    sxoff[i] = NIMG * FRAG + ((k8l >> 3) * MB + m) * 4;
    xk[i] = k8l * 8;
    xm[i] = m < Mloc;
    xmi[i] = m;
    xrow[i] = (const XT*)p.x + (size_t)(row0 + min(m, Mloc - 1)) * p.ldx;
  }
  // ---- norm hand-over (16-row instantiation).  The linear in front of every RMSNorm is a residual add (o_proj,
  // down_proj): its epilogue leaves sum(h^2) per row and tile group next to h, and the consumer turns those 32..40
  // partial sums into rsqrt(mean + eps) and normalises while it stages x -- no launch for the norm, no second pass
  // over x.  (Letting every workgroup compute the statistics itself from x was measured slower than the launch.)
  // 16- and 32-row workgroups.  Measured per batch size (same box, norm_handover 1 vs 0): 8 rows int4 +2.6 %, 32 rows bf16
  // +1.8 % (Qwen3-14B) / +3.2 % (Mistral-7B), int4 +-0; the 64-row form was built too and LOST 3 % (bf16) / 6.5 % (int4 + LoRA):
  // eight table loads per thread at the head and the normalisation of four pieces per chunk inside a loop that is already
  // bound by its vector instructions -- above 32 rows the separate norm launch stays.
  // (32 rows: 16-bit weights only -- int4 measured +-0, and the int8 SwiGLU form would spill inside its 128-register budget)
  constexpr bool NH = !X32 && (MT == 1 || (MT == 2 && QB == 0));
  constexpr int MBC = MB < 64 ? MB : 64;
  __shared__ float rs_sh[NH ? MB : 16];
  __shared__ float sqp_sh[SK_NW * (NH ? MBC : 16)];
  const bool norm = NH && p.sq_in != nullptr;
  u32x4 xr[MT], xw[MT], xr2[X32 ? MT : 1], xnw[X32 ? MT : 1], xnw2[X32 ? MT : 1];
  float sqacc[X32 ? MT : 1];
#pragma unroll
  for (int i = 0; i < (X32 ? MT : 1); ++i) sqacc[i] = 0.f;
  const bool dn = X32 && p.defer_norm != 0;
  auto load_x = [&](int c) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      // ([hi | lo] weights, float32 activations only: the second half of K meets the same x again.  kx is a multiple of the
      // chunk, so the wrap is uniform -- scalar arithmetic, no register per thread)
      const int cb = c * SK_KC;
      const int cw = (X32 && cb >= p.kx) ? cb - p.kx : cb;
      const int k = cb + xk[i] < p.K ? cw + xk[i] : 0;
      xr[i] = *(const u32x4*)(xrow[i] + k);
      if constexpr (X32) {
        xr2[i] = *(const u32x4*)(xrow[i] + k + 4);
        if (dn) {
          xnw[i] = *(const u32x4*)((const float*)p.norm_w + k);
          xnw2[i] = *(const u32x4*)((const float*)p.norm_w + k + 4);
        }
      }
      if constexpr (NH) {
        if (norm) xw[i] = *(const u32x4*)((const AT*)p.norm_w + k);
      }
    }
  };
  auto store_x = [&](int c, unsigned char* buf, bool fresh = true) {   // fresh: first time this chunk is staged (its squares count)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool ok = xm[i] && (c * SK_KC + xk[i] < p.K);
      if constexpr (X32) {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        const u32x4 f0 = ok ? xr[i] : z4, f1 = ok ? xr2[i] : z4;
        float xf[8] = {__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z), __uint_as_float(f0.w),
                       __uint_as_float(f1.x), __uint_as_float(f1.y), __uint_as_float(f1.z), __uint_as_float(f1.w)};
        if (dn) {                                // deferred RMSNorm: squares of the raw x, then x * w_norm
          const float wf[8] = {__uint_as_float(xnw[i].x), __uint_as_float(xnw[i].y), __uint_as_float(xnw[i].z), __uint_as_float(xnw[i].w),
                               __uint_as_float(xnw2[i].x), __uint_as_float(xnw2[i].y), __uint_as_float(xnw2[i].z), __uint_as_float(xnw2[i].w)};
          float ss = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { ss = fmaf(xf[j], xf[j], ss); xf[j] *= wf[j]; }
          if (fresh && c * SK_KC < p.kx) sqacc[i] += ss;      // (each x once: not in the second pass of a [hi | lo] matrix)
        }
        AT hi[8], mid[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          hi[j] = (AT)xf[j];
          const float r1 = xf[j] - (float)hi[j];
          mid[j] = (AT)r1;
          lo[j] = (AT)(r1 - (float)mid[j]);
        }
        if constexpr (QUANT) {                   // sum(x) per 64-group in float32; int4: nibble order inside the piece
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) sum += xf[j];
          sum = lane8_sum(sum);
          if ((tid & 7) == 0) *(float*)(buf + sxoff[i]) = sum;
          if constexpr (Q4) {
            AT th[8], tm[8], tl[8];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              th[2 * qd] = hi[qd]; th[2 * qd + 1] = hi[qd + 4];
              tm[2 * qd] = mid[qd]; tm[2 * qd + 1] = mid[qd + 4];
              tl[2 * qd] = lo[qd]; tl[2 * qd + 1] = lo[qd + 4];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = th[j]; mid[j] = tm[j]; lo[j] = tl[j]; }
          }
        }
        *(u32x4*)(buf + woff[i]) = *(const u32x4*)hi;
        *(u32x4*)(buf + FRAG + woff[i]) = *(const u32x4*)mid;
        *(u32x4*)(buf + 2 * FRAG + woff[i]) = *(const u32x4*)lo;
        continue;
      }
      u32x4 v = ok ? xr[i] : u32x4{0u, 0u, 0u, 0u};
      if constexpr (NH) {
        if (norm) {
          AT* e = (AT*)&v;
          const AT* we = (const AT*)&xw[i];
          const float rs = rs_sh[xmi[i]];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const AT xn = (AT)((float)e[j] * rs);               // cast_T(x32 * rsqrt(..))
            e[j] = (AT)((float)xn * (float)we[j]);               // w * (.)  in T
          }
        }
      }
      if constexpr (QUANT) {
        AT* e = (AT*)&v;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)e[j];
        if constexpr (Q4) {                    // nibble order inside the piece (gemv_phase.h)
          AT t2[8];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) { t2[2 * qd] = e[qd]; t2[2 * qd + 1] = e[qd + 4]; }
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = t2[j];
        }
        sum = lane8_sum(sum);
        if ((tid & 7) == 0) *(float*)(buf + sxoff[i]) = sum;
      }
      *(u32x4*)(buf + woff[i]) = v;
    }
  };

  // reader side: fragment of piece P = 4 b + g, rows 16 mt + c16 -> byte (b*4*MB + 16 mt) * 16 + lane_off[b & 1]
  int lane_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) lane_off[par] = (g * MB + (c16 ^ ((par * 4 + g) << 1))) * 16;

  auto mfma_unit = [&](int slot, int i, const unsigned char* cur) {
    if constexpr (Q8) {
      // unit i = the i-th 64-block (= quantisation group) of the chunk: y += s * sum(q x) + b * sum(x)
      u32x4 wq[NA][2];
      float sc[NA], bb[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const u32x4 v = wr[a][slot];
        wq[a][0] = unpack_q8<AT>(v.x, v.y);
        wq[a][1] = unpack_q8<AT>(v.z, v.w);
        sc[a] = (float)__builtin_bit_cast(AT, (unsigned short)sr[a][slot]);
        bb[a] = (float)__builtin_bit_cast(AT, (unsigned short)br[a][slot]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int pb = (i * 2) * 4 * MB + mt * 16;
        const f32x4 sxv = *(const f32x4*)(cur + NIMG * FRAG + (i * MB + mt * 16 + g * 4) * 4);
        f32x4 dq[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) dq[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int img = 0; img < NIMG; ++img) {
          const u32x4 af0 = *(const u32x4*)(cur + img * FRAG + lane_off[0] + pb * 16);
          const u32x4 af1 = *(const u32x4*)(cur + img * FRAG + lane_off[1] + (pb + 4 * MB) * 16);
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            dq[a] = mfma16<AT>(af0, wq[a][0], dq[a]);
            dq[a] = mfma16<AT>(af1, wq[a][1], dq[a]);
          }
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const f32x4 d = dq[a];
          acc[a][mt].x = fmaf(sc[a], d.x, fmaf(bb[a], sxv.x, acc[a][mt].x));
          acc[a][mt].y = fmaf(sc[a], d.y, fmaf(bb[a], sxv.y, acc[a][mt].y));
          acc[a][mt].z = fmaf(sc[a], d.z, fmaf(bb[a], sxv.z, acc[a][mt].z));
          acc[a][mt].w = fmaf(sc[a], d.w, fmaf(bb[a], sxv.w, acc[a][mt].w));
        }
      }
    } else if constexpr (!Q4) {
#pragma unroll
      for (int img = 0; img < NIMG; ++img) {
        u32x4 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          af[mt] = *(const u32x4*)(cur + img * FRAG + lane_off[i & 1] + (i * 4 * MB + mt * 16) * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int a = 0; a < NA; ++a) acc[a][mt] = mfma16<AT>(af[mt], wr[a][slot], acc[a][mt]);
      }
    } else {
      // (see Phase::mfma_u) after the swap {x,y} = quant group A, {z,w} = group B in every lane
      uint32_t dw[NA][4];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const u32x4 v = wr[a][slot];
        auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
        dw[a][0] = r0[0]; dw[a][1] = r1[0]; dw[a][2] = r0[1]; dw[a][3] = r1[1];
      }
      // The 2 MT (64-group, row tile) steps of a unit as a software pipeline over the LDS operands (round 4, as in gemm_q4.hip):
      // the reads of step j + 1 -- two x fragments and the group sums -- are issued in front of the MFMAs of step j.  Dense
      // 16-bit activations only (one image); the float32-activation form keeps the plain loop below.
      // Measured (same box, alternating libraries): <= 16 rows +0.6 % (Mistral-7B int4, B = 8: 3293 / 3285 -> 3313 / 3308 tok/s);
      // 32-row slabs -0.3 % (config-5 shard 7653 / 7621 -> 7633 / 7599), 6 row tiles spill -- so: the 16-row instantiations only.
      if constexpr (NIMG == 1 && MT == 1) {
        auto rd = [&](int idx, u32x4& a0, u32x4& a1, f32x4& sv) {
          const int sg = idx / MT, mt = idx % MT;
          const int pb = ((i * 2 + sg) * 2) * 4 * MB + mt * 16;
          a0 = *(const u32x4*)(cur + lane_off[0] + pb * 16);
          a1 = *(const u32x4*)(cur + lane_off[1] + (pb + 4 * MB) * 16);
#ifdef MI_ABL_NOSX
          sv = f32x4{1.f, 1.f, 1.f, 1.f};
#else
          sv = *(const f32x4*)(cur + NIMG * FRAG + ((i * 2 + sg) * MB + mt * 16 + g * 4) * 4);
#endif
        };
        u32x4 c0, c1, n0, n1;
        f32x4 csx, nsx;
        rd(0, c0, c1, csx);
        u32x4 wq[NA][2];
        float sc[NA], bb[NA];
#pragma unroll
        for (int idx = 0; idx < 2 * MT; ++idx) {
          const int sg = idx / MT, mt = idx % MT;
          if (idx + 1 < 2 * MT) rd(idx + 1, n0, n1, nsx);
          if (mt == 0) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
#ifdef MI_ABL_NOUNPACK      // timing-only ablation builds (tools/debug/build_ablation_libs.sh): results are wrong on purpose
              wq[a][0] = u32x4{dw[a][sg * 2 + 0], dw[a][sg * 2 + 1], dw[a][sg * 2 + 0], dw[a][sg * 2 + 1]};
              wq[a][1] = u32x4{dw[a][sg * 2 + 1], dw[a][sg * 2 + 0], dw[a][sg * 2 + 1], dw[a][sg * 2 + 0]};
#else
              wq[a][0] = unpack_q4<AT>(dw[a][sg * 2 + 0]);
              wq[a][1] = unpack_q4<AT>(dw[a][sg * 2 + 1]);
#endif
              sc[a] = (float)((const AT*)&sr[a][slot])[sg];
              bb[a] = (float)((const AT*)&br[a][slot])[sg] - Magic<AT>::offs * sc[a];
            }
          }
#pragma unroll
          for (int a = 0; a < NA; ++a) {
#ifdef MI_ABL_NOFMA
            acc[a][mt] = mfma16<AT>(c0, wq[a][0], acc[a][mt]);
            acc[a][mt] = mfma16<AT>(c1, wq[a][1], acc[a][mt]);
#else
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = mfma16<AT>(c0, wq[a][0], d);
            d = mfma16<AT>(c1, wq[a][1], d);
            acc[a][mt].x = fmaf(sc[a], d.x, fmaf(bb[a], csx.x, acc[a][mt].x));
            acc[a][mt].y = fmaf(sc[a], d.y, fmaf(bb[a], csx.y, acc[a][mt].y));
            acc[a][mt].z = fmaf(sc[a], d.z, fmaf(bb[a], csx.z, acc[a][mt].z));
            acc[a][mt].w = fmaf(sc[a], d.w, fmaf(bb[a], csx.w, acc[a][mt].w));
#endif
          }
          c0 = n0; c1 = n1; csx = nsx;
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        u32x4 wq[NA][2];
        float sc[NA], bb[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#ifdef MI_ABL_NOUNPACK      // timing-only ablation builds (tools/debug/build_ablation_libs.sh): results are wrong on purpose
          wq[a][0] = u32x4{dw[a][sg * 2 + 0], dw[a][sg * 2 + 1], dw[a][sg * 2 + 0], dw[a][sg * 2 + 1]};
          wq[a][1] = u32x4{dw[a][sg * 2 + 1], dw[a][sg * 2 + 0], dw[a][sg * 2 + 1], dw[a][sg * 2 + 0]};
#else
          wq[a][0] = unpack_q4<AT>(dw[a][sg * 2 + 0]);
          wq[a][1] = unpack_q4<AT>(dw[a][sg * 2 + 1]);
#endif
          sc[a] = (float)((const AT*)&sr[a][slot])[sg];
          bb[a] = (float)((const AT*)&br[a][slot])[sg] - Magic<AT>::offs * sc[a];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int pb = ((i * 2 + sg) * 2) * 4 * MB + mt * 16;
#ifdef MI_ABL_NOSX
          const f32x4 sxv = f32x4{1.f, 1.f, 1.f, 1.f};
#else
          const f32x4 sxv = *(const f32x4*)(cur + NIMG * FRAG + ((i * 2 + sg) * MB + mt * 16 + g * 4) * 4);
#endif
          f32x4 dq[NA];
#pragma unroll
          for (int a = 0; a < NA; ++a) dq[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int img = 0; img < NIMG; ++img) {
            const u32x4 af0 = *(const u32x4*)(cur + img * FRAG + lane_off[0] + pb * 16);
            const u32x4 af1 = *(const u32x4*)(cur + img * FRAG + lane_off[1] + (pb + 4 * MB) * 16);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
#ifdef MI_ABL_NOFMA
              acc[a][mt] = mfma16<AT>(af0, wq[a][0], acc[a][mt]);
              acc[a][mt] = mfma16<AT>(af1, wq[a][1], acc[a][mt]);
#else
              dq[a] = mfma16<AT>(af0, wq[a][0], dq[a]);
              dq[a] = mfma16<AT>(af1, wq[a][1], dq[a]);
#endif
            }
          }
#ifdef MI_ABL_NOFMA
          continue;
#endif
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            const f32x4 d = dq[a];
            acc[a][mt].x = fmaf(sc[a], d.x, fmaf(bb[a], sxv.x, acc[a][mt].x));
            acc[a][mt].y = fmaf(sc[a], d.y, fmaf(bb[a], sxv.y, acc[a][mt].y));
            acc[a][mt].z = fmaf(sc[a], d.z, fmaf(bb[a], sxv.z, acc[a][mt].z));
            acc[a][mt].w = fmaf(sc[a], d.w, fmaf(bb[a], sxv.w, acc[a][mt].w));
          }
        }
      }
      }
    }
  };

  // ================= prologue: activations first (older in the vmcnt queue), then the first UK blocks
  unsigned char* cur = smem;
  unsigned char* nxt = smem + BUF;
  SK_STAMP(0);
  // norm hand-over: ALL the producer's partial sums in one round trip (512 threads, <= 2 loads each), issued FIRST so that
  // waiting for them leaves the activation and weight loads in flight.  (The former 16-thread loop paid one L2 / memory
  // round trip per 8 parts behind the weight loads, which made the hand-over a wash.)  Straight-line: no load under a branch.
  // thread t holds parts t / MB + (512 / MB) u of row t mod MB (table [part][sq_ld], this workgroup's rows at row0)
  constexpr int SQ_LD = NH ? SQ_PARTS_MAX * MB / (SK_NW * 64) : 1;
  constexpr int SQ_STEP = NH ? (SK_NW * 64) / MB : 1;
  float pv[SQ_LD];
  if constexpr (NH) {
    const float* sqsrc = norm ? p.sq_in : (const float*)p.x;
#pragma unroll
    for (int u = 0; u < SQ_LD; ++u) {
      const int part = tid / MB + SQ_STEP * u;
      pv[u] = sqsrc[(norm && part < p.sq_parts) ? part * p.sq_ld + row0 + tid % MB : 0];      // host: sq_parts <= SQ_PARTS_MAX
    }
  }
  load_x(c0);
#pragma unroll
  for (int u = 0; u < UK; ++u) issue(u, u_begin + u);
  // residual epilogue (<= 16 rows): h is fetched now -- no other workgroup of this launch writes these columns -- instead of
  // as one more dependent round trip at the very end of the last arriver's chain.  Straight-line (pointer select).
  constexpr bool HPRE = !SWIGLU && MT == 1;
  float hpre[HPRE ? MT * 4 : 1];
  if constexpr (HPRE) {
    const bool res = p.epi == EPI_RESID;
    const XT* hp = res ? (const XT*)p.resid : (const XT*)p.out;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        hpre[mt * 4 + r] = (float)hp[(size_t)(row0 + min(mt * 16 + g * 4 + r, Mloc - 1)) * p.ldo + tile * 16 + c16];
  }
  if constexpr (NH) {
    if (norm) {
      // a fixed tree: the thread's registers in order, the lanes of the wave that hold the same row (16 rows: +16 / +32;
      // 32 rows: +32), then the 8 waves through LDS (64 rows: wave w holds the parts 8 u + w of every row)
      float v = 0.f;
#pragma unroll
      for (int u = 0; u < SQ_LD; ++u) v += (tid / MB + SQ_STEP * u < p.sq_parts) ? pv[u] : 0.f;
      if constexpr (MB == 16) v += __shfl_xor(v, 16);
      if constexpr (MB <= 32) v += __shfl_xor(v, 32);
      if (lane < MBC) sqp_sh[wave * MBC + lane] = v;
      __syncthreads();
      if (tid < MB) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SK_NW; ++w) t += sqp_sh[w * MBC + tid];
        rs_sh[tid] = 1.0f / sqrtf(t / (float)p.K + p.eps);
      }
      __syncthreads();
    }
  }
  SK_STAMP(1);
  store_x(c0, cur);
#if defined(MI_SK_TRACE) && !defined(MI_SK_TRACE_ENTRY)
  if (lane == 0) p.trace[(size_t)(blockIdx.x & 1023) * 16 + 8 + wave] = wall_clock64();     // every wave: its share of the first chunk is written
#endif
  __syncthreads();
  SK_STAMP(2);

  // ================= the slice
  for (int c = c0; c < c1; c += CPI) {
#pragma unroll
    for (int ci = 0; ci < CPI; ++ci) {
      const int cc = c + ci;                      // (int4: the second chunk of a trip may lie past the slice)
      const int cn = min(cc + 1, c1 - 1);
      load_x(cn);
#pragma unroll
      for (int i = 0; i < UPC; ++i) {
        const int slot = (ci * UPC + i) % UK, unit = cc * UPC + i;
        if (unit < u_end) mfma_unit(slot, i, cur);
        __builtin_amdgcn_sched_barrier(0);
        issue(slot, unit + UK);
        __builtin_amdgcn_sched_barrier(0);
#ifndef MI_ABL_NOSTAGE
        if (i == UPC / 2) store_x(cn, nxt, cc + 1 < c1);   // (past the slice's end the last chunk is staged again, unused)
#endif
      }
      __syncthreads();                            // the next chunk's fragments are complete; this chunk's are free
      unsigned char* t = cur; cur = nxt; nxt = t;
    }
  }

  SK_STAMP(3);
  // ================= deferred RMSNorm: this slice's sum of x^2 per row, in a fixed order (8 lanes of a piece by DPP,
  // then the 4 k-groups of the chunk through LDS)
  __shared__ float rs_row[32];
  float slice_sq = 0.f;                          // threads < MB: the slice's sum for row tid
  if constexpr (X32) {
    if (dn) {
      float* sq4 = (float*)smem;                 // [4][MB] (the activation buffers are free: the slice loop ended on a barrier)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float v = lane8_sum(sqacc[i]);
        const int r = (tid + i * (SK_NW * 64)) >> 3;
        if ((tid & 7) == 0) sq4[(r / MB) * MB + (r % MB)] = v;
      }
      __syncthreads();
      if (tid < MB) slice_sq = (sq4[tid] + sq4[MB + tid]) + (sq4[2 * MB + tid] + sq4[3 * MB + tid]);
      if (p.ksplit > 1) {
        if (tid < MB) __hip_atomic_store(&p.sqws[((size_t)grp * p.ksplit + s) * MB + tid], slice_sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (tid < MB) rs_row[tid] = 1.0f / sqrtf(slice_sq / (float)p.kx + p.eps);
        __syncthreads();
      }
    }
  }

  // ================= split K: publish the partial tile; the last workgroup of the group adds them up.
  // Hand-off as in gemv_phase.h seam_arrive: partials are stored write-through (agent-scope atomics -> sc1, they do not
  // stay dirty in this XCD's L2), every wave drains its stores, the workgroup is counted, and the reader uses
  // agent-scope loads.  No release / acquire FENCES: on this part they write back / invalidate the whole L2 of the
  // XCD per wave, which made the launch several times slower than the streaming itself.
  if (p.ksplit > 1) {
    if (valid) {
      unsigned long long* wp = (unsigned long long*)(p.ws + (((size_t)((slab * p.ksplit + s) * p.ntiles + tile) * (NA * MT)) * 64 + lane) * 4);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[a][mt];
          unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
          unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
          __hip_atomic_store(wp + (size_t)(a * MT + mt) * 128, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(wp + (size_t)(a * MT + mt) * 128 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores have left
    __syncthreads();
    SK_STAMP(4);
    if (tid == 0) {
      unsigned* cp = &p.ctr[grp * p.nslab + slab];
      const unsigned old = __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)(p.ksplit - 1);
      if (last) __hip_atomic_store(cp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
      last_sh = last;
    }
    __syncthreads();
    SK_STAMP(5);
    if (!last_sh) return;
    if constexpr (X32) {
      if (dn) {                                  // the slices' sums of squares, in slice order
        float* sqs = (float*)smem;               // [ksplit][MB] (ksplit <= 16, MB <= 32: one load per thread, one round trip)
        if (tid < p.ksplit * MB)
          sqs[tid] = __hip_atomic_load(&p.sqws[(size_t)grp * p.ksplit * MB + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid < MB) {
          float tot = 0.f;
          for (int s2 = 0; s2 < p.ksplit; ++s2) tot += sqs[s2 * MB + tid];
          rs_row[tid] = 1.0f / sqrtf(tot / (float)p.kx + p.eps);
        }
        __syncthreads();
      }
    }
    if (valid) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      // slice order, whoever arrived last; CB slices' partials are fetched per round trip (they come from other XCDs,
      // i.e. from memory: with one slice per trip the o_proj / down_proj combines spent 3-4 trips here) at <= 32 rows
      if constexpr (NA * MT <= 2) {
        constexpr int CB = 8;
        for (int sb = 0; sb < p.ksplit; sb += CB) {
          u32x4 pvv[CB][NA][MT];
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            if (sb + j < p.ksplit) {            // (uniform; nothing else is in flight here)
              const float* rp = p.ws + (((size_t)((slab * p.ksplit + sb + j) * p.ntiles + tile) * (NA * MT)) * 64 + lane) * 4;
#pragma unroll
              for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) pvv[j][a][mt] = load16_agent(rp + (size_t)(a * MT + mt) * 256);
            }
          }
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            if (sb + j < p.ksplit) {
#pragma unroll
              for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                  const u32x4 v = pvv[j][a][mt];
                  acc[a][mt].x += __uint_as_float(v.x); acc[a][mt].y += __uint_as_float(v.y);
                  acc[a][mt].z += __uint_as_float(v.z); acc[a][mt].w += __uint_as_float(v.w);
                }
            }
          }
        }
      } else {
#pragma unroll 2
        for (int s2 = 0; s2 < p.ksplit; ++s2) {
          const float* rp = p.ws + (((size_t)((slab * p.ksplit + s2) * p.ntiles + tile) * (NA * MT)) * 64 + lane) * 4;
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const u32x4 v = load16_agent(rp + (size_t)(a * MT + mt) * 256);
              acc[a][mt].x += __uint_as_float(v.x); acc[a][mt].y += __uint_as_float(v.y);
              acc[a][mt].z += __uint_as_float(v.z); acc[a][mt].w += __uint_as_float(v.w);
            }
        }
      }
    }
  }
  SK_STAMP(6);
  const bool want_sq = NH && p.sq_out != nullptr;      // (uniform; then spare waves stay for the barriers below)
  if (!valid && !want_sq) return;

  // ================= epilogue: lane (c16, g) holds y[16 mt + 4 g + r][16 tile + c16]
  const int n = tile * 16 + c16;
  AT* out = (AT*)p.out;
  float hsq[NH ? MT * 4 : 1];
#pragma unroll
  for (int i = 0; i < (NH ? MT * 4 : 1); ++i) hsq[i] = 0.f;
  // residual epilogue above 16 rows: ALL of the lane's h values in one round trip.  Left to the loop below, every h load sat
  // behind the previous h store (same array: the compiler keeps the order) -- 16 dependent round trips at 64 rows, 6.5 us of
  // o_proj's 24.6 (stamps, Qwen3-14B int4, B = 64).
  constexpr bool HLATE = !SWIGLU && MT > 1;
  float hlate[HLATE ? MT * 4 : 1];
  if constexpr (HLATE) {
    if (p.epi == EPI_RESID) {                    // (uniform; nothing else is in flight here)
      const XT* hp = (const XT*)p.resid;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hlate[mt * 4 + r] = (float)hp[(size_t)(row0 + min(mt * 16 + g * 4 + r, Mloc - 1)) * p.ldo + n];
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ml = mt * 16 + g * 4 + r;
      if (ml >= Mloc || !valid) continue;
      const int m = row0 + ml;                   // the row in the caller's arrays
      float y0 = acc[0][mt][r];
      if constexpr (X32) {                        // float32 storage, run-time logical rounding (gemv_v1.hip's epilogue)
        float* o32 = (float*)p.out;
        const float rsm = dn ? rs_row[ml] : 1.0f;
        y0 *= rsm;
        if constexpr (SWIGLU) {
          const float gt = round_rt(y0, p.rnd), up = round_rt(acc[NA - 1][mt][r] * rsm, p.rnd);
          const float sig = round_rt(1.0f / (1.0f + expf(-gt)), p.rnd);
          const float sl = round_rt(gt * sig, p.rnd);
          o32[(size_t)m * p.ldo + n] = round_rt(sl * up, p.rnd);
        } else {
          float y = round_rt(y0, p.rnd);
          if (p.lora_t != nullptr) {             // LoRALinear in this mode: y + T(scale (x A) B), the term in float32 (App. A.6)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
              const int r0 = sl ? p.lora_row0_1 : p.lora_row0_0;
              const int ln = sl ? p.lora_n_1 : p.lora_n_0;
              const int rk = sl ? p.lora_rank_1 : p.lora_rank_0;
              const float* lb = sl ? p.lora_b1 : p.lora_b0;
              if (lb != nullptr && n >= r0 && n < r0 + ln) {
                const float* tt = p.lora_t + (size_t)m * p.lora_t_ld + sl * (p.lora_t_ld / 2);
                float z = lora_dot(tt, lb + (n - r0), ln, rk);
                z = round_rt((sl ? p.lora_scale_1 : p.lora_scale_0) * z, p.rnd);
                y = round_rt(y + z, p.rnd);
              }
            }
          }
          if (p.epi == EPI_RESID) {
            float* h = (float*)p.resid;
            float hv;
            if constexpr (HPRE) hv = hpre[mt * 4 + r];
            else if constexpr (HLATE) hv = hlate[mt * 4 + r];
            else hv = h[(size_t)m * p.ldo + n];
            h[(size_t)m * p.ldo + n] = round_rt(hv + y, p.rnd);
          } else {
            o32[(size_t)m * p.ldo + n] = y;     // EPI_STORE and EPI_STORE_F32 coincide
          }
        }
        continue;
      }
      if constexpr (SWIGLU) {
        const float gt = (float)(AT)y0, up = (float)(AT)acc[NA - 1][mt][r];
        const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
        const float sl = (float)(AT)(gt * sig);
        out[(size_t)m * p.ldo + n] = (AT)(sl * up);
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
        if (p.epi == EPI_STORE) out[(size_t)m * p.ldo + n] = (AT)y;
        else if (p.epi == EPI_STORE_F32) ((float*)p.out)[(size_t)m * p.ldo + n] = y;
        else {
          AT* h = (AT*)p.resid;
          float h0;
          if constexpr (HPRE) h0 = hpre[mt * 4 + r];
          else if constexpr (HLATE) h0 = hlate[mt * 4 + r];
          else h0 = (float)h[(size_t)m * p.ldo + n];
          const AT hv = (AT)(h0 + y);
          h[(size_t)m * p.ldo + n] = hv;
          if constexpr (NH) hsq[mt * 4 + r] = (float)hv * (float)hv;
        }
      }
    }
  if constexpr (NH) {
    if (want_sq) {
      // sum of h^2 per row over this group's 8 tiles x 16 columns, in a fixed order: [wave][row][column] through LDS,
      // a thread per (row, column) adds the 8 waves, a 16-lane butterfly adds the columns
      float* sq = (float*)smem;                   // (the activation buffers are free: the slice loop ended on a barrier)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sq[(wave * MB + mt * 16 + g * 4 + r) * 16 + c16] = hsq[mt * 4 + r];
      __syncthreads();
#pragma unroll
      for (int it = 0; it < (MB * 16 + SK_NW * 64 - 1) / (SK_NW * 64); ++it) {
        const int idx = tid + it * (SK_NW * 64);           // (MB * 16 is a multiple of 256: whole 16-lane groups are in or out)
        float v = 0.f;
        if (idx < MB * 16) {
#pragma unroll
          for (int w = 0; w < SK_NW; ++w) v += sq[(w * MB + (idx >> 4)) * 16 + (idx & 15)];
        }
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        if (idx < MB * 16 && (idx & 15) == 0) p.sq_out[(size_t)grp * p.sq_ld + row0 + (idx >> 4)] = v;
      }
    }
  }
  SK_STAMP(7);
}

// 16-row tiles of activations per workgroup: 1, 2, 3, 4, 6 or 8 (65..96 rows -> 6, 97..128 -> 8)
int skinny_mt(size_t rows) {
  const int mt = (int)((rows + 15) / 16);
  return mt == 5 ? 6 : mt == 7 ? 8 : mt;
}

struct SkinnyPlan { int ntiles, ngroups, nchunks, ksplit, mt, na, nslab; size_t ws_bytes; };

// Row slabs (SkinnyParams::nslab): int4 weights above 32 rows can run as 32-row slabs, concurrently in one launch.
// Measured (Qwen3-14B int4 + LoRA, 64 rows, same box): the linears with one accumulator stream (q|k|v, o, down: 112 VGPRs at
// 32 rows, two workgroups per CU) gain ~6 % as slabs; the SwiGLU pair does not (148 VGPRs at 32 rows = one workgroup per
// CU either way; capped at 128 it spills inside the loop): 76.9 us at 64 rows vs 82.9 (3 K slices) / 87.4 (2) / 92.1 (1) as
// slabs.  So: slabs for the single-stream linears above 32 rows and for everything above 96 rows (where the 128-row
// int4 SwiGLU instantiation would spill -- int4 used to stop at 96 rows).  MI_SKINNY_SLABS=0 / 2: never / always (A/B).
static int skinny_slabs_mode() {
  static const int v = [] { const char* e = getenv("MI_SKINNY_SLABS"); return e == nullptr ? 1 : atoi(e); }();
  return v;
}
static bool skinny_slabs_on() { return skinny_slabs_mode() != 0; }

// ksplit: time of the launch ~ rounds x (bytes one workgroup moves) / (rate one CU gets), with the partial
// tiles (written, then read once) counted as extra bytes.  A lone workgroup cannot pull more than ~40 GB/s
// (bytes in flight / latency), so too few workgroups are slow even though HBM is idle.
SkinnyPlan skinny_plan(const LinearW& W, const GemvCall& c, size_t rows) {
  SkinnyPlan pl{};
  pl.na = c.epi == EPI_SWIGLU ? 2 : 1;
  pl.mt = skinny_mt(rows);
  pl.nslab = 1;
  {
    const bool q4w = wk_is_quant(W.wk) && !(W.wk == WK_Q8_BF16 || W.wk == WK_Q8_F16);
    const bool want = skinny_slabs_mode() == 2 || (skinny_slabs_mode() == 1 && (pl.na == 1 || rows > 96));
    if (q4w && c.act != MI_F32 && rows > 32 && want) { pl.mt = 2; pl.nslab = (int)((rows + 31) / 32); }
    // (16-bit weights as two 16-row slabs at 17..32 rows -- Qwen3-14B gate|up has 136 tile groups for 256 CUs -- was measured
    // and dropped: Qwen3-14B gate|up 72 -> 145 us, Mistral-7B 54.2 -> 51.7 us with the step unchanged.)
  }
  pl.ntiles = (c.epi == EPI_SWIGLU ? c.pair_offset : W.N) / 16;
  pl.ngroups = (pl.ntiles + SK_NW - 1) / SK_NW;
  pl.nchunks = (W.K + SK_KC - 1) / SK_KC;
  const bool quant = wk_is_quant(W.wk), q8 = W.wk == WK_Q8_BF16 || W.wk == WK_Q8_F16;
  const double cus = gemv_cu_count(), bw = 6.3e12;
  const double bpe = q8 ? 1.0625 : quant ? 0.5625 : 2.0;
  const double unit_p = 2.0 * SK_NW * 16 * pl.na * pl.mt * 16 * 4;          // written + read
  // what ONE CU moves: r1 with one workgroup on it, r2 with two or more (quantised weights at <= 32 rows keep two
  // resident: <= 128 VGPRs, 17 KiB LDS).  int4: the loop is bound by the dequantisation -- measured 28 GB/s alone (Qwen3-14B
  // gate|up at ksplit 1: 136 workgroups streamed at 3.85 TB/s) and ~35 GB/s for a pair; that shape now splits 3 ways
  // (408 workgroups, 1880 -> 2117 tok/s).  Everything else keeps the rates the model was fitted with.
  const bool q4_small = quant && !q8 && pl.mt <= 2 && c.act != MI_F32;
  // int4 above 32 rows: ONE workgroup per CU (150-200 VGPRs) and its loop is bound by the work per weight block, which grows
  // with the row tiles (stamps, Qwen3-14B int4 at 64 rows: 9-13 GB/s per workgroup; gate|up ran unsplit on 136 CUs at
  // 1.7 TB/s).  With the rate the model was fitted with (40 GB/s) it priced a second round above the idle CUs.
  const bool q4_big = quant && !q8 && pl.mt > 2 && c.act != MI_F32;
  const double r_big = 28e9 / (1.0 + 0.45 * (pl.mt - 1));
  const double r1 = q4_small ? 28e9 : q4_big ? r_big : 40e9, r2 = q4_small ? 35.5e9 : r1;
  double best = 0.0;
  pl.ksplit = 1;
  for (int s = 1; s <= std::min(16, pl.nchunks); ++s) {
    const double units = (double)pl.ngroups * s * pl.nslab;
    const double unit_w = (double)SK_NW * 16 * pl.na * bpe * SK_KC * ((pl.nchunks + s - 1) / s), unit_pp = s > 1 ? unit_p : 0.0;
    const double unit = unit_w + unit_pp;
    const double k = std::ceil(units / cus);                               // workgroups on the busiest CU
    const double share = bw / std::min(units, cus);
    double t;
    // (row slabs: the slabs of a unit read the same weights at about the same time on one XCD, so memory sees them once;
    // two workgroups are resident per CU, so k counts pairs)
    if (q4_small && pl.nslab > 1) t = std::max(units / pl.nslab * unit_w / bw + units * unit_pp / bw, std::ceil(units / (2 * cus)) * 2 * unit / r2) + 0.2e-6 * s;
    else if (q4_small) t = std::max(units * unit / bw, k * unit / (k == 1.0 ? r1 : r2)) + 0.2e-6 * s;   // (+ the last arriver's read per slice)
    else if (q4_big) t = k * (unit_w / std::min(r1, share) + unit_pp / std::min(40e9, share));     // partial tiles move at the memory rate
    else t = k * std::max(std::min(units, cus) * unit / bw, unit / r1);
    if (s == 1 || t < best * 0.97) { best = t; pl.ksplit = s; }            // a larger split has to earn its partials
  }
  // A/B: MI_SKINNY_FORCE="N:K:ksplit,..." overrides the model for the linears named
  static const char* force = getenv("MI_SKINNY_FORCE");
  if (force != nullptr) {
    for (const char* q = force; *q;) {
      int fn = 0, fk = 0, fs = 0;
      if (sscanf(q, "%d:%d:%d", &fn, &fk, &fs) == 3 && fn == W.N && fk == W.K && fs >= 1) pl.ksplit = std::min(fs, std::min(16, pl.nchunks));
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
  }
  pl.ws_bytes = pl.ksplit > 1 ? (size_t)pl.nslab * pl.ksplit * pl.ntiles * pl.na * pl.mt * 1024 : 0;
  if (pl.ksplit > 1 && c.act == MI_F32 && c.pro == PRO_NORM)       // deferred RMSNorm: the K slices' row sums of squares
    pl.ws_bytes += (size_t)pl.ngroups * pl.ksplit * 16 * pl.mt * sizeof(float);
  return pl;
}

template <typename AT, int QB, int MT, bool SWIGLU>
int launch_k(const SkinnyParams& p, int grid, hipStream_t st) {
  auto kern = skinny_kernel<AT, QB, MT, SWIGLU>;
  const size_t lds = 2 * ((size_t)16 * MT * 512 + (QB ? 16 * MT * 16 : 0));
  // the opt-in belongs to the kernel object of the CURRENT device (one engine per GPU may live in one process:
  // server --devices), and several scheduler threads launch concurrently: one flag per device, set after the call
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 512 + 128 * 16)));
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(SK_NW * 64), lds, st, p);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename AT, int QB, bool SWIGLU>
int launch_mt(const SkinnyParams& p, int mt, int grid, hipStream_t st) {
  switch (mt) {
    case 1: return launch_k<AT, QB, 1, SWIGLU>(p, grid, st);    // int8: every decode step; 16-bit / int4: 9..16 rows
    case 2: return launch_k<AT, QB, 2, SWIGLU>(p, grid, st);
    case 3: return launch_k<AT, QB, 3, SWIGLU>(p, grid, st);
    case 4: return launch_k<AT, QB, 4, SWIGLU>(p, grid, st);
    case 6: return launch_k<AT, QB, 6, SWIGLU>(p, grid, st);
    case 8: return launch_k<AT, QB, 8, SWIGLU>(p, grid, st);
  }
  return fail(MI_ERR_INVALID, "gemm_skinny: 1..128 rows");
}

template <typename AT>
int launch_at(const SkinnyParams& p, int qb, bool swiglu, int mt, int grid, hipStream_t st) {
  if (qb == 4) return swiglu ? launch_mt<AT, 4, true>(p, mt, grid, st) : launch_mt<AT, 4, false>(p, mt, grid, st);
  if (qb == 8) return swiglu ? launch_mt<AT, 8, true>(p, mt, grid, st) : launch_mt<AT, 8, false>(p, mt, grid, st);
  return swiglu ? launch_mt<AT, 0, true>(p, mt, grid, st) : launch_mt<AT, 0, false>(p, mt, grid, st);
}

}  // namespace

// 16-bit / int4 weights: gemv_mfma.hip serves up to 8 rows, this kernel 9..64 (its 16-row instantiation beats the
// M <= 16 form of gemv_mfma.hip, which stages all of x per workgroup: Mistral-7B bf16 B = 16 3557 -> 3835 tok/s, int4
// 4181 -> 5973, Qwen3-14B bf16 1401 -> 2386; at 8 rows and below gemv_mfma.hip wins, 3.59 vs 3.95 ms/step).
// MI_SKINNY_MIN_ROWS moves the hand-over (A/B).
static int skinny_min_rows() {
  static const int v = [] { const char* e = getenv("MI_SKINNY_MIN_ROWS"); return e ? std::max(1, atoi(e)) : 9; }();
  return v;
}

bool gemm_skinny_supported(const LinearW& W, const GemvCall& c, size_t rows) {
  if (c.force_v1 || W.layout != 1) return false;
  // float32 activations on 16-bit dense weights (PagedKVCache mode): the 16-row instantiation with x split three ways
  if (c.act == MI_F32) {
    static const bool x32_ok = getenv("MI_SKINNY_NO_F32") == nullptr;
    const int n32 = c.epi == EPI_SWIGLU ? c.pair_offset : W.N;
    const bool wok = W.wk == WK_BF16 || (W.group == 64 && ((W.wk == WK_Q4_BF16 && W.K % 128 == 0) || (W.wk == WK_Q8_BF16 && W.K % 64 == 0)));
    const bool lora = W.lora_b[0] != nullptr || W.lora_b[1] != nullptr;      // (the term is added in the plain-store epilogue)
    return x32_ok && wok && rows >= 1 && rows <= 32 && W.K % 32 == 0 && c.ldx % 4 == 0 && n32 % 16 == 0 &&
           (!lora || c.epi == EPI_STORE);
  }
  if (c.rnd != RND_NONE) return false;
  const bool q8 = ((W.wk == WK_Q8_BF16 && c.act == MI_BF16) || (W.wk == WK_Q8_F16 && c.act == MI_F16)) && W.group == 64 &&
                  W.K % 64 == 0;                         // int8 has no M <= 16 kernel of its own: every decode step runs here
  // int4: this kernel also wins below 9 rows, on every linear (M = 8, Mistral-7B shapes, us: q|k|v 9.2 vs 12.4, o 9.8 vs
  // 9.8, gate|up 19.0 vs 24.3, down 14.8 vs 23.7, lm_head 17.7 vs 24.5 -- more than the two RMSNorm launches it adds)
  static const bool q4_small = getenv("MI_SKINNY_Q4_MIN_ROWS") == nullptr;   // A/B: set = hand-over at skinny_min_rows() as for 16-bit
  if (rows < 1 || rows > 128) return false;
  if (wk_is_quant(W.wk) && !q8 && rows > (skinny_slabs_on() ? 128u : 96u)) return false;   // (the 128-row int4 SwiGLU instantiation would spill registers; as 32-row slabs it runs)
  // 16-bit weights below the hand-over: only the linears without a norm in front (o_proj, down_proj) whose K leaves
  // gemv_mfma a short last activation chunk (K mod 4096 in 1..1024: Qwen3-14B's 5120 and 17408) -- M = 8, us: o 14.3 vs
  // 16.2, down 34.5 vs 45.5, in the bench 1134 -> 1236 tok/s; no RMSNorm launch is added.  (Mistral-7B's down_proj,
  // K = 14336 = 3.5 chunks, measured neutral: 2213 vs 2203 tok/s, and stays on gemv_mfma.)
  static const bool ragged_small = getenv("MI_SKINNY_NO_RAGGED_K") == nullptr;
  const int rem = W.K % 4096;
  const bool small_ok = q8 || (q4_small && wk_is_quant(W.wk)) ||
                        (ragged_small && !wk_is_quant(W.wk) && c.pro == PRO_NONE && W.K > 4096 && rem > 0 && rem <= 1024);
  if ((int)rows < skinny_min_rows() && !small_ok) return false;
  const bool dense = (W.wk == WK_BF16 && c.act == MI_BF16) || (W.wk == WK_F16 && c.act == MI_F16);
  const bool q4 = ((W.wk == WK_Q4_BF16 && c.act == MI_BF16) || (W.wk == WK_Q4_F16 && c.act == MI_F16)) && W.group == 64 &&
                  W.K % 128 == 0;
  if (!dense && !q4 && !q8) return false;
  if (W.K % 32 != 0 || c.ldx % 8 != 0) return false;
  const int n = c.epi == EPI_SWIGLU ? c.pair_offset : W.N;
  return n % 16 == 0;
}

size_t gemm_skinny_ws_bytes(const LinearW& W, const GemvCall& c, size_t rows) {
  if (gemm_q4_supported(W, c, rows)) return gemm_q4_ws_bytes(W, c, rows);
  return skinny_plan(W, c, rows).ws_bytes;
}
int gemm_skinny_groups(const LinearW& W, const GemvCall& c, size_t rows) {      // arrival counters the launch needs
  if (gemm_q4_supported(W, c, rows)) return gemm_q4_groups(W, c, rows);
  const SkinnyPlan pl = skinny_plan(W, c, rows);
  return pl.ngroups * pl.nslab;
}

// c.pro must be PRO_NONE (normalise first); `ws` holds gemm_skinny_ws_bytes(), `ctr` gemm_skinny_groups() zeroed words
#ifdef MI_SK_TRACE
namespace {
struct SkTraceRec { int launch, N, K, ksplit, grid, epi, M, qb, pro, act, mt, pad; };
constexpr int TR_LAUNCHES = 4096, TR_WG = 1024;
unsigned long long* tr_buf = nullptr;
SkTraceRec tr_rec[TR_LAUNCHES];
long tr_count = 0;
}
// the stamp slab of one launch of ANOTHER kernel family (gemv_mfma.hip); `kind` goes to the record's qb field (-1 = gemv_mfma)
unsigned long long* dbg_trace_slot(int N, int K, int grid, int epi, int M, int kind, int pro, int act) {
  if (!tr_buf) {
    if (hipMalloc(&tr_buf, (size_t)TR_LAUNCHES * TR_WG * 128) != hipSuccess) return nullptr;
    hipMemset(tr_buf, 0, (size_t)TR_LAUNCHES * TR_WG * 128);
  }
  tr_rec[tr_count % TR_LAUNCHES] = SkTraceRec{(int)tr_count, N, K, 1, std::min(grid, TR_WG), epi, M, kind, pro, act, 1, 0};
  unsigned long long* p = tr_buf + (size_t)(tr_count % TR_LAUNCHES) * TR_WG * 16;
  ++tr_count;
  return p;
}
extern "C" int mi_debug_sk_trace_dump(const char* path) {
  if (!tr_buf) return 1;
  hipDeviceSynchronize();
  const long n = std::min<long>(tr_count, TR_LAUNCHES);
  FILE* f = fopen(path, "wb");
  if (!f) return 2;
  std::vector<unsigned long long> h((size_t)TR_WG * 16);
  fwrite(&n, sizeof(long), 1, f);
  for (long i = tr_count - n; i < tr_count; ++i) {
    const SkTraceRec& r = tr_rec[i % TR_LAUNCHES];
    hipMemcpy(h.data(), tr_buf + (size_t)(i % TR_LAUNCHES) * TR_WG * 16, (size_t)r.grid * 128, hipMemcpyDeviceToHost);
    fwrite(&r, sizeof(r), 1, f);
    fwrite(h.data(), 8, (size_t)r.grid * 16, f);
  }
  fclose(f);
  return 0;
}
#endif

int launch_gemm_skinny(const LinearW& W, const GemvCall& c, size_t rows, hipStream_t st, void* ws, unsigned* ctr, int ksplit) {
  if (ksplit <= 0 && gemm_q4_supported(W, c, rows)) return launch_gemm_q4(W, c, rows, st, ws, ctr);     // int4 above 16 rows (gemm_q4.hip)
  const bool defer_norm = c.pro == PRO_NORM && c.act == MI_F32 && c.rnd == RND_NONE && c.norm_w != nullptr;
  if (c.pro != PRO_NONE && !defer_norm && !(gemm_skinny_handover_ld(W, c, rows) > 0 && c.sq_in != nullptr && c.sq_parts > 0))
    return fail(MI_ERR_INVALID, "gemm_skinny: normalise the activations first (or hand over the row sums of squares)");
  if (c.sq_in != nullptr && c.sq_parts > SQ_PARTS_MAX) return fail(MI_ERR_INVALID, "gemm_skinny: norm hand-over from more than 64 tile groups");
  SkinnyPlan pl = skinny_plan(W, c, rows);
  if (ksplit > 0) {
    pl.ksplit = std::min(ksplit, pl.nchunks);
    if (defer_norm) pl.ksplit = std::min(pl.ksplit, 16);
    pl.ws_bytes = pl.ksplit > 1 ? (size_t)pl.nslab * pl.ksplit * pl.ntiles * pl.na * pl.mt * 1024 : 0;
    if (pl.ksplit > 1 && defer_norm) pl.ws_bytes += (size_t)pl.ngroups * pl.ksplit * 16 * pl.mt * sizeof(float);
  }
  if (pl.ksplit > 1 && (ws == nullptr || ctr == nullptr)) return fail(MI_ERR_INVALID, "gemm_skinny: workspace missing");
  SkinnyParams p{};
  p.x = c.x; p.ldx = c.ldx; p.M = (int)rows;
  p.w = W.w; p.N = W.N; p.K = W.K;
  p.kx = (c.kx > 0 && c.act == MI_F32) ? c.kx : W.K;
  if (p.kx != W.K && (2 * p.kx != W.K || p.kx % SK_KC != 0)) return fail(MI_ERR_INVALID, "gemm_skinny: a [hi | lo] matrix has 2 kx columns, kx a multiple of 256");
  p.epi = c.epi; p.out = c.out; p.ldo = c.ldo; p.resid = c.resid; p.pair_offset = c.epi == EPI_SWIGLU ? c.pair_offset : 0;
  const bool nh = (pl.mt == 1 || (pl.mt == 2 && !wk_is_quant(W.wk))) && c.act != MI_F32;
  p.sq_out = (nh && c.epi == EPI_RESID) ? c.sq_out : nullptr;
  p.sq_in = (nh && c.pro == PRO_NORM) ? c.sq_in : nullptr; p.sq_parts = c.sq_parts;
  p.sq_ld = pl.nslab * 16 * pl.mt;
  p.norm_w = c.norm_w; p.eps = c.eps; p.rnd = c.rnd;
  p.ntiles = pl.ntiles; p.ksplit = pl.ksplit; p.nchunks = pl.nchunks;
  p.nslab = pl.nslab; p.nunits = pl.ngroups * pl.ksplit;
  p.ws = (float*)ws; p.ctr = ctr;
  p.defer_norm = defer_norm ? 1 : 0;
  p.sqws = (defer_norm && pl.ksplit > 1) ? (float*)((char*)ws + (size_t)pl.ksplit * pl.ntiles * pl.na * pl.mt * 1024) : nullptr;
  p.lora_t = c.lora_t; p.lora_t_ld = c.lora_t_ld;
  p.lora_b0 = W.lora_b[0]; p.lora_b1 = W.lora_b[1];
  p.lora_row0_0 = W.lora_row0[0]; p.lora_n_0 = W.lora_n[0]; p.lora_rank_0 = W.lora_rank[0]; p.lora_scale_0 = W.lora_scale[0];
  p.lora_row0_1 = W.lora_row0[1]; p.lora_n_1 = W.lora_n[1]; p.lora_rank_1 = W.lora_rank[1]; p.lora_scale_1 = W.lora_scale[1];
  const int qb = (W.wk == WK_Q8_BF16 || W.wk == WK_Q8_F16) ? 8 : wk_is_quant(W.wk) ? 4 : 0;
  const bool sw = c.epi == EPI_SWIGLU;
  const int grid = pl.nslab > 1 ? (pl.ngroups * pl.ksplit + 7) / 8 * 8 * pl.nslab : pl.ngroups * pl.ksplit;
#ifdef MI_SK_TRACE
  if (!tr_buf) { MI_HIP(hipMalloc(&tr_buf, (size_t)TR_LAUNCHES * TR_WG * 128)); MI_HIP(hipMemset(tr_buf, 0, (size_t)TR_LAUNCHES * TR_WG * 128)); }
  tr_rec[tr_count % TR_LAUNCHES] = SkTraceRec{(int)tr_count, W.N, W.K, pl.ksplit, std::min(grid, TR_WG), c.epi, (int)rows, qb, c.pro, c.act, pl.mt, 0};
  p.trace = tr_buf + (size_t)(tr_count % TR_LAUNCHES) * TR_WG * 16;
  ++tr_count;
#endif
  if (c.act == MI_F32) {
    p.sq_out = nullptr; p.sq_in = nullptr;
    auto launch32 = [&](auto kern) -> int {
      const size_t lds = 2 * ((size_t)3 * 16 * pl.mt * 512 + (qb ? 16 * pl.mt * 16 : 0));
      MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (3 * 32 * 512 + 32 * 16)));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(SK_NW * 64), lds, st, p);
      MI_HIP(hipGetLastError());
      return MI_OK;
    };
#define GO32(QBV) do { \
      if (pl.mt == 1) return sw ? launch32(skinny_kernel<bf16, QBV, 1, true, true>) : launch32(skinny_kernel<bf16, QBV, 1, false, true>); \
      return sw ? launch32(skinny_kernel<bf16, QBV, 2, true, true>) : launch32(skinny_kernel<bf16, QBV, 2, false, true>); } while (0)
    if (qb == 4) GO32(4);
    if (qb == 8) GO32(8);
    GO32(0);
#undef GO32
  }
  return c.act == MI_BF16 ? launch_at<bf16>(p, qb, sw, pl.mt, grid, st) : launch_at<f16>(p, qb, sw, pl.mt, grid, st);
}

int gemm_skinny_ksplit(const LinearW& W, const GemvCall& c, size_t rows) {
  if (gemm_q4_supported(W, c, rows)) return gemm_q4_ksplit(W, c, rows);
  return skinny_plan(W, c, rows).ksplit;
}

// RMSNorm hand-over between two launches: row stride of the [tile groups][ld] table of sums of squares that this call would
// write (residual epilogue) or read (PRO_NORM) -- its rows padded to whole workgroups -- or 0 when its instantiation
// (more than 32 rows per workgroup, float32 activations) keeps the separate norm launch.  Producer and consumer of one
// hand-over must agree on it.
int gemm_skinny_handover_ld(const LinearW& W, const GemvCall& c, size_t rows) {
  if (c.act == MI_F32) return 0;
  if (gemm_q4_supported(W, c, rows)) return 0;
  const SkinnyPlan pl = skinny_plan(W, c, rows);
  if (pl.mt != 1 && !(pl.mt == 2 && !wk_is_quant(W.wk))) return 0;
  return pl.nslab * 16 * pl.mt;
}
int gemm_skinny_tile_groups(const LinearW& W, const GemvCall& c, size_t rows) { return skinny_plan(W, c, rows).ngroups; }

}  // namespace mi
