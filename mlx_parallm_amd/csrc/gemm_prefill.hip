// gemm_prefill.hip -- MFMA tile GEMM for the prefill call of generate_step (L > 1 tokens per row):
//   C[M][N] = X[M][K] . W[N][K]^T   (nn.Linear on (B, L, H) activations: llama.py:93,143,165)
// plus the row-wise RMSNorm that feeds it (llama.py:187,189) and the residual / SwiGLU epilogues.
// 16-bit activations, dense 16-bit weights in the tile-major layout of repack.hip.  The decode
// kernels (gemv_mfma.hip) stream W once per <= 16 rows; here a 128 x 128 output tile re-uses
// every weight fragment over 128 rows, which makes the call MFMA-bound instead of HBM-bound.
//
// Block = 256 threads = 4 waves as 2 (M) x 2 (N); wave tile 64 x 64 = 4 x 4 MFMA tiles
// (v_mfma_f32_16x16x32, 64 accumulator registers); K step 64.
//   * X tile (128 x 64) goes global -> registers -> LDS, rows padded to 144 B so that the 16 lanes
//     of an A-fragment read (16 consecutive rows, one 16-B chunk each) hit 16 distinct bank groups;
//     double-buffered, the next tile's global loads are issued before the current tile's MFMAs.
//   * W fragments come straight from global memory in MFMA order: in the tile-major layout the
//     fragment of (16 rows, 32 k) is one contiguous KiB, lane l at 16 l -- no LDS, no transpose.
//     The next step's 8 fragments are prefetched into a second register set.
//   * SwiGLU: a wave's four N tiles are two gate tiles and the two matching up tiles, so
//     silu(gate) * up happens in registers.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace mi {

int gemv_cu_count();      // gemv_mfma.hip

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef MI_GEMM_VARIANT
#define MI_GEMM_VARIANT 0       // A/B experiments (tools/debug/gemm_ab.py); 0 = the shipped kernel
#endif
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDA = 72;   // LDS row stride in elements (144 B)

template <typename T>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct GemmParams {
  const void* x; int ldx; int M;
  const void* w; int N, K;          // W tile-major; N = rows of W (SWIGLU: gate rows = N/2); K = its reduction length
  int ka;                           // reduction length of x: K, or K/2 when W is a [hi | lo] pair (x is walked twice)
  int kw;                           // reduction length of W (its row-tile stride): K, or K/3 when x is a [hi | mid | lo]
                                    // triple of float32 activations (W is walked three times).  k step t multiplies
                                    // x[.., t mod ka] with W[.., t mod kw]; with ka = 3 K0, kw = 2 K0 the 6 K0 steps meet
                                    // every (x part, W part) pair once.
  int out32;                        // outputs / residual stream are float32 (PagedKVCache mode), no 16-bit rounding
  int epi; void* out; int ldo; void* resid; int pair_offset;
  // 128 x 128 tile only: K split over gridDim.z workgroups when the (M, N) grid alone leaves most CUs idle (a prompt of a
  // few hundred tokens against N = 4096: 32 x 2..8 blocks); slice z leaves its float32 partial tile in ws[z][M][N] and
  // splitk_epilogue_kernel adds the slices in order and applies the epilogue
  int ksplit; float* ws;
};

// the epilogues in float32 storage (PagedKVCache mode after layer 0: every op produces a float32 array)
__device__ __forceinline__ void epi32_swiglu(const GemmParams& p, int m, int n, float gt, float up) {
  const float sig = 1.0f / (1.0f + expf(-gt));
  const float sl = gt * sig;
  ((float*)p.out)[(size_t)m * p.ldo + n] = sl * up;
}
__device__ __forceinline__ void epi32_plain(const GemmParams& p, int m, int n, float y) {
  if (p.epi == EPI_STORE) ((float*)p.out)[(size_t)m * p.ldo + n] = y;
  else { float* h = (float*)p.resid; h[(size_t)m * p.ldo + n] = h[(size_t)m * p.ldo + n] + y; }
}

// grid: (n blocks, m blocks).  SWIGLU: a block covers 64 gate columns + the 64 matching up columns.
template <typename AT, bool SWIGLU>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) AT As[2][BM][LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, g = lane >> 4;
  // blockIdx.x = M block (fastest): the blocks that run together share one N block, i.e. the same
  // W tiles, which then stay in the XCD's L2 instead of being re-fetched per block
  const int bm = blockIdx.x, bn = blockIdx.y;
  const int m0 = bm * BM;
  const int nk_all = p.K / BK, nkw = p.kw / BK;
  const int ks0 = (int)(((long)blockIdx.z * nk_all) / p.ksplit), ks1 = (int)(((long)(blockIdx.z + 1) * nk_all) / p.ksplit);
  const int nk = ks1;                               // (the loop runs over [ks0, ks1); "nk" bounds the prefetches)
  const AT* x = (const AT*)p.x;

  // W tiles (16 rows each) of this wave's four N tiles
  int wtile[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    if constexpr (!SWIGLU) wtile[nt] = (bn * BN + wn * 64 + nt * 16) / 16;
    else wtile[nt] = (bn * 64 + wn * 32 + (nt & 1) * 16 + (nt >> 1) * p.pair_offset) / 16;
  }
  const int ntiles_w = p.N / 16;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- X tile loads: 128 rows x 8 chunks of 16 B; thread -> chunks tid, tid+256, ...
  // (straight-line: rows past M read row M - 1 and are zeroed on the way into LDS -- see gemm_tile256_kernel)
  u32x4 areg[4];
  const AT* arow[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, kq = c & 7;
    aok[i] = m0 + row < p.M;
    arow[i] = x + (size_t)min(m0 + row, p.M - 1) * p.ldx + kq * 8;
  }
  auto load_a = [&](int ks) {
    const int ka = (ks * BK) % p.ka;
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[i] = *(const u32x4*)(arow[i] + ka);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = c >> 3, kq = c & 7;
      *(u32x4*)&As[buf][row][kq * 8] = aok[i] ? areg[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  u32x4 breg[3][4][2];     // [set][n tile][k block]; W fragments are fetched two K steps ahead
  auto load_b = [&](int set, int ks) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int t = min(wtile[nt], ntiles_w - 1);
        const char* blk = (const char*)p.w + ((size_t)t * (p.kw / 32) + (size_t)((ks % nkw) * 2 + kb)) * 1024;
        breg[set][nt][kb] = *(const u32x4*)(blk + lane * 16);
      }
  };

  load_a(ks0);
  load_b(0, ks0);
  if (ks0 + 1 < nk) load_b(1, ks0 + 1);
  store_a(0);
  __syncthreads();

  // the K loop is unrolled by 3 so that the register set of each step is a compile-time index
  auto step = [&](int ks, auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    const int cur = (ks - ks0) & 1;
    load_a(min(ks + 1, nk - 1));
    load_b((SET + 2) % 3, min(ks + 2, nk - 1));
    __builtin_amdgcn_sched_barrier(0);           // the loads stay ahead of this step's MFMAs
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      u32x4 af[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = *(const u32x4*)&As[cur][wm * 64 + mt * 16 + c16][kb * 32 + g * 8];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<AT>(af[mt], breg[SET][nt][kb], acc[mt][nt]);
    }
    if (ks + 1 < nk) {
      store_a(cur ^ 1);       // buffer cur^1 was last read one step ago; the barrier of that step separates
      __syncthreads();
    }
  };
  for (int ks = ks0; ks < nk; ks += 3) {
    step(ks, std::integral_constant<int, 0>{});
    if (ks + 1 < nk) step(ks + 1, std::integral_constant<int, 1>{});
    if (ks + 2 < nk) step(ks + 2, std::integral_constant<int, 2>{});
  }

  if constexpr (!SWIGLU) {
    if (p.ksplit > 1) {                            // this K slice's partial tile, float32, for splitk_epilogue_kernel
      float* wz = p.ws + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + mt * 16 + 4 * g + r;
          if (m >= p.M) continue;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int n = bn * BN + wn * 64 + nt * 16 + c16;
            if (n < p.N) wz[(size_t)m * p.N + n] = acc[mt][nt][r];
          }
        }
      return;
    }
  }

  // ---- epilogue: lane (c16, g) holds C[m = 4g + r][n = c16] of every 16 x 16 tile
  AT* out = (AT*)p.out;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 64 + mt * 16 + 4 * g + r;
      if (m >= p.M) continue;
      if constexpr (SWIGLU) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = bn * 64 + wn * 32 + j * 16 + c16;
          if (n >= p.pair_offset) continue;
          if (p.out32) { epi32_swiglu(p, m, n, acc[mt][j][r], acc[mt][j + 2][r]); continue; }
          const float gt = (float)(AT)acc[mt][j][r], up = (float)(AT)acc[mt][j + 2][r];
          const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
          const float sl = (float)(AT)(gt * sig);
          out[(size_t)m * p.ldo + n] = (AT)(sl * up);
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int n = bn * BN + wn * 64 + nt * 16 + c16;
          if (n >= p.N) continue;
          if (p.out32) { epi32_plain(p, m, n, acc[mt][nt][r]); continue; }
          const float y = (float)(AT)acc[mt][nt][r];
          if (p.epi == EPI_STORE) out[(size_t)m * p.ldo + n] = (AT)y;
          else {
            AT* h = (AT*)p.resid;
            h[(size_t)m * p.ldo + n] = (AT)((float)h[(size_t)m * p.ldo + n] + y);
          }
        }
      }
    }
}

// the K slices of gemm_tile_kernel, added in slice order, + the epilogue (EPI_STORE / EPI_RESID); 4 columns per thread
template <typename AT>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmParams p) {
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total4 = (size_t)p.M * p.N / 4;
  if (i4 >= total4) return;
  const size_t e0 = i4 * 4;
  const int m = (int)(e0 / p.N), n = (int)(e0 % p.N);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.ksplit; ++z) a += *(const f32x4*)(p.ws + (size_t)z * p.M * p.N + e0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (p.out32) { epi32_plain(p, m, n + j, a[j]); continue; }
    const float y = (float)(AT)a[j];
    if (p.epi == EPI_STORE) ((AT*)p.out)[(size_t)m * p.ldo + n + j] = (AT)y;
    else { AT* h = (AT*)p.resid; h[(size_t)m * p.ldo + n + j] = (AT)((float)h[(size_t)m * p.ldo + n + j] + y); }
  }
}

constexpr int BM2 = 256, BN2 = 256;

// the epilogue of the 256 x 256 tiles (2 x 4 waves, 128 x 64 per wave): lane (c16, g) holds C[m = 4g + r][n = c16] of
// every 16 x 16 tile
template <typename AT, bool SWIGLU>
__device__ __forceinline__ void tile256_epilogue(const GemmParams& p, f32x4 (&acc)[8][4], int m0, int bn, int wm, int wn, int c16, int g) {
  AT* out = (AT*)p.out;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 128 + mt * 16 + 4 * g + r;
      if (m >= p.M) continue;
      if constexpr (SWIGLU) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = bn * 128 + wn * 32 + j * 16 + c16;
          if (n >= p.pair_offset) continue;
          if (p.out32) { epi32_swiglu(p, m, n, acc[mt][j][r], acc[mt][j + 2][r]); continue; }
          const float gt = (float)(AT)acc[mt][j][r], up = (float)(AT)acc[mt][j + 2][r];
          const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
          const float sl = (float)(AT)(gt * sig);
          out[(size_t)m * p.ldo + n] = (AT)(sl * up);
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int n = bn * BN2 + wn * 64 + nt * 16 + c16;
          if (n >= p.N) continue;
          if (p.out32) { epi32_plain(p, m, n, acc[mt][nt][r]); continue; }
          const float y = (float)(AT)acc[mt][nt][r];
          if (p.epi == EPI_STORE) out[(size_t)m * p.ldo + n] = (AT)y;
          else {
            AT* h = (AT*)p.resid;
            h[(size_t)m * p.ldo + n] = (AT)((float)h[(size_t)m * p.ldo + n] + y);
          }
        }
      }
    }
}

// The residual epilogue (h += y in place) with the loads of h batched: written as in tile256_epilogue every load of
// h sits behind the previous store to h (same array: the compiler keeps the order) -- 128 dependent round trips per
// wave, ~15 us per 256 x 256 block, 13 % of o_proj at 8 x 1024 rows.  Here the 16 values of an M tile are fetched in one
// round trip, then added and stored.
template <typename AT, typename HT>        // HT: storage of the residual stream (AT, or float in the PagedKVCache mode: h += y unrounded)
__device__ __forceinline__ void tile256_epilogue_resid(const GemmParams& p, f32x4 (&acc)[8][4], int m0, int bn, int wm, int wn, int c16, int g) {
  HT* h = (HT*)p.resid;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    HT hv[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int m = min(m0 + wm * 128 + mt * 16 + 4 * g + r, p.M - 1);
        const int n = min(bn * BN2 + wn * 64 + nt * 16 + c16, p.N - 1);
        hv[r][nt] = h[(size_t)m * p.ldo + n];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int m = m0 + wm * 128 + mt * 16 + 4 * g + r;
        const int n = bn * BN2 + wn * 64 + nt * 16 + c16;
        if (m >= p.M || n >= p.N) continue;
        if constexpr (std::is_same<HT, float>::value) h[(size_t)m * p.ldo + n] = hv[r][nt] + acc[mt][nt][r];
        else h[(size_t)m * p.ldo + n] = (AT)((float)hv[r][nt] + (float)(AT)acc[mt][nt][r]);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 in M x 4 in N, 128 x 64 per wave = 8 x 4 MFMA tiles), BOTH operands staged in
// LDS.  Why: with W fragments read per wave from global memory the 128 x 128 kernel moves 48 KB from L2 per
// 2.1 MFLOP (44 flop/B) and sits on the L2 -> CU bandwidth (~19 TB/s at its 840 TFLOP/s); here a K step moves
// 64 KB per 8.4 MFLOP (131 flop/B).  The W block of a K step keeps its tile-major order in LDS -- a fragment is one
// contiguous KiB, read conflict-free -- so the copy global -> registers -> LDS is a plain 16-byte lane-linear move.
// Double-buffered: the next step's 8 loads per thread are in flight during the current step's 64 MFMAs per wave.
// SWIGLU: the block covers 128 gate columns and the 128 matching up columns; a wave's four N tiles are two gate
// tiles + their up tiles.
constexpr int A2_BYTES = BM2 * LDA * 2;                 // one X image (rows padded to 144 B)
constexpr int B2_BYTES = BN2 * BK * 2;                  // one W image: 16 row tiles x 2 k blocks x 1 KiB
constexpr int LDS2_BYTES = 2 * (A2_BYTES + B2_BYTES);
constexpr int AFR = 4;                                  // A fragments in flight from LDS (ring): item i + 3 is read while item i multiplies
constexpr int SYNC_AT = 13;                             // the item in front of which the K step's barrier sits
constexpr int STORE_AT = 11;                            // the item (of 16 per K step) after which the next tile is written to LDS

// BD ("B direct", A/B variant behind MI_GEMM_B_DIRECT): the W fragments do not go through LDS -- in the tile-major
// layout a fragment is one contiguous KiB, so each wave loads the 8 fragments of its four N tiles for the NEXT K step
// straight into a second register set (two sets in turn, the K loop is unrolled by two); LDS then carries only the X
// tile (16 instead of 24 fragment reads per wave and step, half the stores).  Measured 2 % SLOWER than both operands
// in LDS (68.5 k vs 69.9 k prefill tok/s): the LDS pipe was not the limit, the instruction order was (below).
template <typename AT, bool SWIGLU, bool BD>
__global__ __launch_bounds__(512) void gemm_tile256_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;              // 2 x 4 waves
  const int c16 = lane & 15, g = lane >> 4;
  const int bm = blockIdx.x, bn = blockIdx.y;           // M fastest: co-scheduled blocks share the W tiles in L2
  const int m0 = bm * BM2;
  const int nk = p.K / BK, nkw = p.kw / BK;
  const AT* x = (const AT*)p.x;
  const int ntiles_w = p.N / 16;
  auto a_img = [&](int buf) { return (AT*)(smem2 + (size_t)buf * A2_BYTES); };
  auto b_img = [&](int buf) { return smem2 + 2 * (size_t)A2_BYTES + (size_t)buf * B2_BYTES; };

  // the 16 W row tiles of this block, in LDS order: tile slot s (0..15)
  auto w_tile_of = [&](int s) -> int {
    int t;
    if constexpr (!SWIGLU) t = (bn * BN2) / 16 + s;
    else t = (bn * 128 + (s >> 2) * 32 + (s & 1) * 16 + ((s >> 1) & 1) * p.pair_offset) / 16;   // wave wn = s>>2: gate, gate, up, up
    return min(t, ntiles_w - 1);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- global -> registers: X 256 rows x 8 chunks = 2048 pieces, W 16 tiles x 2 k blocks x 64 lanes = 2048 pieces.
  // STRAIGHT-LINE loads (no exec mask, no branch): a load under a condition makes hipcc drain the vmcnt queue at the
  // join -- here that was an exposed L2 round trip per K step.  Rows past M read row M - 1 and are zeroed on the
  // way into LDS; the step after the last re-reads the last step.
  constexpr int NB = BD ? 1 : 4;
  u32x4 areg[4], breg[NB];
  const AT* arow[4];
  const char* brow[NB];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 512 * i, row = c >> 3, kq = c & 7;
    aok[i] = m0 + row < p.M;
    arow[i] = x + (size_t)min(m0 + row, p.M - 1) * p.ldx + kq * 8;
    if constexpr (!BD) {
      const int s = c >> 7, kb = (c >> 6) & 1, l = c & 63;
      brow[i] = (const char*)p.w + ((size_t)w_tile_of(s) * (p.kw / 32) + (size_t)kb) * 1024 + l * 16;
    }
  }
  const char* bptr[4];                                   // BD: this wave's four N tiles, lane-linear
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bptr[nt] = (const char*)p.w + (size_t)w_tile_of(wn * 4 + nt) * (p.kw / 32) * 1024 + lane * 16;
  auto load_a = [&](int ks) {
    const int ka = (ks * BK) % p.ka;
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[i] = *(const u32x4*)(arow[i] + ka);
    if constexpr (!BD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) breg[i] = *(const u32x4*)(brow[i] + (size_t)(ks % nkw) * 2048);
    }
  };
  auto load_bd = [&](int ks, u32x4 (&dst)[2][4]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dst[kb][nt] = *(const u32x4*)(bptr[nt] + (size_t)((ks % nkw) * 2 + kb) * 1024);
  };
  auto store_ab = [&](int buf) {
    AT* A = a_img(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 512 * i, row = c >> 3, kq = c & 7;
      *(u32x4*)&A[row * LDA + kq * 8] = aok[i] ? areg[i] : u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (!BD) {
      unsigned char* B = b_img(buf);
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(B + (size_t)(tid + 512 * i) * 16) = breg[i];     // (slot, kb, lane) order = piece index
    }
  };

  u32x4 bset0[2][4], bset1[2][4];                        // BD: the W fragments of the current / the next K step
  // The fragment pipeline runs ACROSS the K steps: 16 (k block, M tile) items of 4 MFMAs per step; the A fragment of
  // item i + 3 is read from LDS while item i multiplies (ring of AFR registers; 16 % AFR == 0, so the ring index carries
  // over), and for the last three items of a step "item i + 3" is item 0..2 of the NEXT step, read from the other
  // buffer.  That buffer is complete once every wave has stored its share (after item STORE_AT) -- so the step's one
  // barrier sits in front of item SYNC_AT, not at the end of the step, and the wait for the first fragments of a
  // step hides behind the previous step's last 12 MFMAs instead of stopping every wave of the workgroup at once.
  u32x4 bf[2][4], af[AFR];
  auto a_frag = [&](const AT* A, int it) {
    return *(const u32x4*)&A[(wm * 128 + (it & 7) * 16 + c16) * LDA + (it >> 3) * 32 + g * 8];
  };
  auto b_frags = [&](const unsigned char* B, int kb, u32x4 (&dst)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dst[nt] = *(const u32x4*)(B + (size_t)(((wn * 4 + nt) * 2 + kb) * 64 + lane) * 16);
  };
  // one K step: `bc` holds this step's W fragments (BD), `bn` receives the next step's
  auto step = [&](int ks, u32x4 (&bc)[2][4], u32x4 (&bn)[2][4]) {
    const int cur = ks & 1;
    load_a(min(ks + 1, nk - 1));
    if constexpr (BD) load_bd(min(ks + 1, nk - 1), bn);
    __builtin_amdgcn_sched_barrier(0);           // keep the loads HERE, a whole step of MFMAs ahead of their use (hipcc
                                                 // otherwise sinks them next to the LDS stores at the bottom of the step)
    const AT* A = a_img(cur);
    const AT* An = a_img(cur ^ 1);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (it == SYNC_AT) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                         // the other buffer holds the next step's tile; this step's reads are done
        __builtin_amdgcn_sched_barrier(0);
      }
      const int nx = it + AFR - 1;
      af[nx % AFR] = nx < 16 ? a_frag(A, nx) : a_frag(An, nx - 16);
      if constexpr (!BD) {
        if (it == 3) b_frags(b_img(cur), 1, bf[1]);
        if (it == SYNC_AT) b_frags(b_img(cur ^ 1), 0, bf[0]);     // (bf[0] is dead since item 7)
      }
#if MI_GEMM_VARIANT == 1
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[it & 7][nt] = mfma16<AT>(af[it % AFR], BD ? bc[it >> 3][nt] : bf[it >> 3][nt], acc[it & 7][nt]);
#if MI_GEMM_VARIANT == 1
      __builtin_amdgcn_s_setprio(0);
#endif
      // the order is imposed on hipcc's scheduler (sched_group_barrier: 0x100 = LDS read, 0x008 = MFMA); left alone
      // it bunches the fragment reads and waits for them in front of single MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (!BD && (it == 3 || it == SYNC_AT)) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      if (it == STORE_AT) {
        // the next step's tile goes into the other buffer (its readers finished before the last barrier): the loads
        // have had three quarters of a step to arrive
        __builtin_amdgcn_sched_barrier(0);
        store_ab(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  load_a(0);
  if constexpr (BD) load_bd(0, bset0);
  store_ab(0);
  __syncthreads();
  // static priority for the younger wave of every SIMD (waves 4..7 lose every issue arbitration by age otherwise):
  // prefill 72.9 k -> 73.5 k tok/s, two runs each on one box.  (Priority flipped around every MFMA group: -1 %.)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  if constexpr (!BD) b_frags(b_img(0), 0, bf[0]);
#pragma unroll
  for (int i = 0; i < AFR - 1; ++i) af[i] = a_frag(a_img(0), i);
  for (int ks = 0; ks < nk; ks += 2) {
    step(ks, bset0, bset1);
    if (ks + 1 < nk) step(ks + 1, bset1, bset0);
  }

  if constexpr (!SWIGLU) {
    if (p.epi == EPI_RESID) {
      if (p.out32) tile256_epilogue_resid<AT, float>(p, acc, m0, bn, wm, wn, c16, g);
      else tile256_epilogue_resid<AT, AT>(p, acc, m0, bn, wm, wn, c16, g);
      return;
    }
  }
  tile256_epilogue<AT, SWIGLU>(p, acc, m0, bn, wm, wn, c16, g);
}


// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, both operands staged by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, the
// copies of a K tile stay in flight across barriers for three to four phases, and the two waves of a SIMD take the matrix
// core in turn.  Same wave grid and accumulator layout as gemm_tile256_kernel (2 x 4 waves, 128 x 64 per wave,
// acc[8][4]), same MFMA chain per accumulator (so the same float32 sums, bit for bit), same epilogue.
// Mistral-7B linears over 8 x 1024 rows, random bf16 operands, interleaved rounds in one process (tools/debug/
// gemm_prefill_ab.py): 1.15 / 1.03 / 1.15 / 1.14 PFLOP/s (q|k|v, o, gate|up, down) on the register-staged tile,
// 1.38 / 1.17 / 1.35 / 1.38 here.
//
// LDS: two K-tile buffers of 64 KiB, each four UNITS of 16 KiB -- a unit is what all 8 waves stage together in one
// phase (two 1-KiB DMA instructions per wave):
//     A-h0 / A-h1   rows {wm * 128 + h * 64 + [0, 64)} of the X tile (128 rows x 128 B), image row ir = wm * 64 + r
//     B-k0 / B-k1   the 16 W row tiles of k block kb: 16 fragments of 1 KiB in tile-major order (a fragment read is one
//                   contiguous KiB: conflict-free as it lies)
// A image: the DMA writes lane-linear (base + 16 lane), so the layout is chosen through the SOURCE address of each lane.
// Image rows are paired into 256-byte bank rows; piece (ir, q) (q = 16-byte chunk 0..7 of the row's 128 bytes) lies at
//     (ir >> 1) * 256 + ((((ir & 1) << 3) | q) ^ ((ir >> 1) & 7)) * 16
// -- every lane group of a ds_read_b128 fragment read ({0-3, 12-15, 20-27}, ... : 16 rows x one or two q) hits 16
// different 16-byte columns (checked exhaustively), and the 8 lanes that fill a row fetch its whole 128-byte line.
//
// Schedule (ping-pong).  A K tile is four phases of 16 MFMAs per wave: (k block, row half) = (0,0) (0,1) (1,0) (1,1).  A
// phase is  [reads of ITS fragments, stage one unit, s_waitcnt lgkmcnt(0), s_waitcnt vmcnt(N)] s_barrier [16 MFMAs] s_barrier;
// waves 4..7 -- the second wave of every SIMD -- enter through one extra barrier (and waves 0..3 leave through one), so
// every s_barrier pairs the end of one group's read section with the end of the other's MFMA section: while one wave of a
// SIMD reads and issues its DMA, the other has the matrix core.  Reads come BEFORE the DMA issue: an LDS-DMA instruction
// holds the wave's issue for 60..180 cycles (the CU's one address unit serves four waves at once), the fragments arrive
// meanwhile.
//   stage:  j=0: A-h1(t+1)   j=1: B-k0(t+2)   j=2: B-k1(t+1)   j=3: A-h0(t+2)
// Write-after-read: the reads a group issues in phase Q have completed at that group's next barrier (lgkmcnt(0)), both
// groups' two barriers after phase Q began; a unit last read in phase Q is restaged in phase Q + 1 at the earliest
// (A-h0: read (t,0) (t,2) -> (t,3); A-h1: (t,1) (t,3) -> (t+1,0); B-k0: (t,0) -> (t,1); B-k1: (t,2) -> (t+1,2)).
// Read-after-write: a wave's DMA has landed when its counted vmcnt wait says so, everybody's when both groups have passed
// the barrier behind that wait -- waited for in the read section of phase W, first read in phase W + 1:
//     unit          staged in     waited for in   first read in   DMAs issued since (may stay in flight)
//     A-h1(t+1)     (t, j=0)      (t+1, j=0)      (t+1, j=1)      4 units -> vmcnt(8)
//     B-k1(t+1)     (t, j=2)      (t+1, j=1)      (t+1, j=2)      3 units -> vmcnt(6)
//     B-k0(t+2)     (t, j=1)      (t+1, j=3)      (t+2, j=0)      (in order: landed before A-h0(t+2))
//     A-h0(t+2)     (t, j=3)      (t+1, j=3)      (t+2, j=0)      4 units -> vmcnt(8)
// (vmcnt counts in order, and these DMAs are the loop's only vector-memory instructions.)  Past the last tile the stages
// repeat the last tile's sources -- straight-line, the counts stay what the table says -- into units nobody reads again.
// Measured and dropped (same tool, same box; the numbers are q|k|v / down): lockstep phases with the next phase's
// fragments read one phase ahead and ONE barrier per phase 1.26 / 1.25 PFLOP/s; this ping-pong with the DMA issued in
// front of the reads 1.29 / 1.30; the DMA issued at the head of the MFMA section 1.25 / 1.24; the units going round
// rings of five slots (all 160 KiB, every unit staged a tile earlier: five to seven phases in flight) 1.26 / 1.27 -- the
// copies' latency is covered with two buffers, what counts is the length of the read section; no s_setprio: the same.
constexpr int DMA_BUF = 65536, DMA_UNIT = 16384;

__device__ __forceinline__ void dma_kib(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;

template <typename AT, bool SWIGLU>
__global__ __launch_bounds__(512) void gemm_dma256_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;              // 2 x 4 waves
  const int c16 = lane & 15, g = lane >> 4;
  const int bm = blockIdx.x, bn = blockIdx.y;
  const int m0 = bm * BM2;
  // K split over gridDim.z workgroups (a prompt of 256..4095 rows against N = 4096: the (M, N) grid alone leaves most CUs
  // idle): slice z multiplies the K tiles [tb, tb + nk) and leaves its float32 partial tile in ws[z][M][N];
  // splitk_epilogue_kernel adds the slices in order and applies the epilogue (host: >= 2 tiles per slice)
  const int nk_all = p.K / BK, nkw = p.kw / BK;
  const int tb = (int)(((long)blockIdx.z * nk_all) / gridDim.z);
  const int nk = (int)(((long)(blockIdx.z + 1) * nk_all) / gridDim.z) - tb;
  const int ntiles_w = p.N / 16;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem2;

  auto w_tile_of = [&](int s) -> int {                   // the 16 W row tiles of this block, in LDS order (slot s)
    int t;
    if constexpr (!SWIGLU) t = (bn * BN2) / 16 + s;
    else t = (bn * 128 + (s >> 2) * 32 + (s & 1) * 16 + ((s >> 1) & 1) * p.pair_offset) / 16;   // wave wn = s>>2: gate, gate, up, up
    return min(t, ntiles_w - 1);
  };

  // ---- DMA sources.  A unit h, instruction i (0 / 1) of this wave = image rows (2 wave + i) * 8 + [0, 8) of the unit:
  // lane -> bank row R = lane >> 4 of the instruction's four, column col = lane & 15; row in pair = col >> 3, and the
  // chunk q it must fetch so that the linear write realises the swizzle: q = (col & 7) ^ (R & 7), R & 7 = 4 i + (lane >> 4).
  const char* asrc[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rl = lane >> 4, col = lane & 15;
      const int ir = (2 * wave + i) * 8 + 2 * rl + (col >> 3);            // image row of the unit: wm' * 64 + r
      const int q = (col & 7) ^ (4 * i + rl);
      const int row = (ir >> 6) * 128 + h * 64 + (ir & 63);                 // row of the 256-row tile
      asrc[h][i] = (const char*)p.x + ((size_t)min(m0 + row, p.M - 1) * p.ldx + q * 8) * sizeof(AT);
    }
  const char* bsrc[2];                                                     // B unit: fragment slots 2 wave, 2 wave + 1
#pragma unroll
  for (int i = 0; i < 2; ++i) bsrc[i] = (const char*)p.w + (size_t)w_tile_of(2 * wave + i) * (p.kw / 32) * 1024 + lane * 16;

  // unit A-h / B-kb of tile t (sources past the last tile repeat it) -> the 16 KiB at LDS byte offset `off`
  auto stage_a_at = [&](int t, int h, unsigned off) {
    const int ts = tb + min(t, nk - 1);
    const size_t koff = (size_t)((ts * BK) % p.ka) * sizeof(AT);
    const unsigned dst = lds0 + off + wave * 2048;
    dma_kib(asrc[h][0] + koff, dst);
    dma_kib(asrc[h][1] + koff, dst + 1024);
  };
  auto stage_b_at = [&](int t, int kb, unsigned off) {
    const int ts = tb + min(t, nk - 1);
    const size_t koff = (size_t)((ts % nkw) * 2 + kb) * 1024;
    const unsigned dst = lds0 + off + wave * 2048;
    dma_kib(bsrc[0] + koff, dst);
    dma_kib(bsrc[1] + koff, dst + 1024);
  };
  // ---- fragment reads.  A (kb, mt4) of the unit at `off`: image row wm * 64 + mt4 * 16 + c16, chunk kb * 4 + g
  const unsigned a_lane = lds0 + (unsigned)((wm * 32 + (c16 >> 1)) * 256 + (((((c16 & 1) << 3) | g) ^ (c16 >> 1)) * 16));
  const unsigned b_lane = lds0 + (unsigned)(wn * 4 * 1024 + lane * 16);
  auto a_frags_at = [&](unsigned off, int kb, u32x4 (&dst)[4]) {
    const unsigned base = (a_lane ^ (unsigned)(kb << 6)) + off;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) dst[mt] = *(const lds_u32x4_t*)(size_t)(base + mt * 8 * 256);
  };
  auto b_frags_at = [&](unsigned off, u32x4 (&dst)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dst[nt] = *(const lds_u32x4_t*)(size_t)(b_lane + off + nt * 1024);
  };
  // tile t in buffer t & 1 = [A-h0][A-h1][B-k0][B-k1]
  auto stage_a = [&](int t, int h) { stage_a_at(t, h, (t & 1) * DMA_BUF + h * DMA_UNIT); };
  auto stage_b = [&](int t, int kb) { stage_b_at(t, kb, (t & 1) * DMA_BUF + (2 + kb) * DMA_UNIT); };
  auto a_frags = [&](int t, int kb, int h, u32x4 (&dst)[4]) { a_frags_at((t & 1) * DMA_BUF + h * DMA_UNIT, kb, dst); };
  auto b_frags = [&](int t, int kb, u32x4 (&dst)[4]) { b_frags_at((t & 1) * DMA_BUF + (2 + kb) * DMA_UNIT, dst); };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[1][4], bf[2][4];

  auto mma = [&](int h, const u32x4 (&a)[4], const u32x4 (&b)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[h * 4 + mt][nt] = mfma16<AT>(a[mt], b[nt], acc[h * 4 + mt][nt]);
  };

  {
    // prologue: tile 0 whole and what the phases of tiles -2 / -1 would have staged, in their order
    stage_b(0, 0); stage_a(0, 0); stage_a(0, 1); stage_b(1, 0); stage_b(0, 1); stage_a(1, 0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (wave >= 4) asm volatile("s_barrier" ::: "memory");
#define PP_READS_DONE(VM) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
      if (VM >= 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM < 0 ? 0 : VM) : "memory"); \
      asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define PP_MMA_DONE() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); asm volatile("s_barrier" ::: "memory"); \
      __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_ORDER() __builtin_amdgcn_sched_barrier(0)
    for (int t = 0; t < nk; ++t) {
      b_frags(t, 0, bf[0]); a_frags(t, 0, 0, af[0]); PP_ORDER(); stage_a(t + 1, 1);
      PP_READS_DONE(8);
      mma(0, af[0], bf[0]);
      PP_MMA_DONE();
      a_frags(t, 0, 1, af[0]); PP_ORDER(); stage_b(t + 2, 0);
      PP_READS_DONE(6);
      mma(1, af[0], bf[0]);
      PP_MMA_DONE();
      b_frags(t, 1, bf[1]); a_frags(t, 1, 0, af[0]); PP_ORDER(); stage_b(t + 1, 1);
      PP_READS_DONE(-1);
      mma(0, af[0], bf[1]);
      PP_MMA_DONE();
      a_frags(t, 1, 1, af[0]); PP_ORDER(); stage_a(t + 2, 0);
      PP_READS_DONE(8);
      mma(1, af[0], bf[1]);
      PP_MMA_DONE();
    }
#undef PP_ORDER
#undef PP_READS_DONE
#undef PP_MMA_DONE
    if (wave < 4) asm volatile("s_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA of this workgroup may land after it has left the CU

  if constexpr (!SWIGLU) {
    if (gridDim.z > 1) {                                   // this slice's partial tile, float32
      float* wz = p.ws + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 128 + mt * 16 + 4 * g + r;
          if (m >= p.M) continue;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int n = bn * BN2 + wn * 64 + nt * 16 + c16;
            if (n < p.N) wz[(size_t)m * p.N + n] = acc[mt][nt][r];
          }
        }
      return;
    }
  }
  if constexpr (!SWIGLU) {
    if (p.epi == EPI_RESID) {
      if (p.out32) tile256_epilogue_resid<AT, float>(p, acc, m0, bn, wm, wn, c16, g);
      else tile256_epilogue_resid<AT, AT>(p, acc, m0, bn, wm, wn, c16, g);
      return;
    }
  }
  tile256_epilogue<AT, SWIGLU>(p, acc, m0, bn, wm, wn, c16, g);
}


// ---------------------------------------------------------------------------------------------------------------
// The same 256 x 256 x 64 tile on v_mfma_f32_32x32x16: per wave and K step 32 MFMAs of 32 cycles instead of 64 of 16.
// Why: a 16x16x32 MFMA leaves 8 of its 16 cycles for other instructions to issue, the loop has ~135 of them per 64 MFMAs
// (24 LDS reads, 8 + 8 staging loads / stores, ~45 address VALU, the waits) -- more than one wave's MFMA shadow holds;
// a 32x32x16 MFMA leaves 24 of 32 cycles, i.e. 6 issue slots per MFMA and ~190 per K step.  Same staging, same LDS images
// (the A fragment of 32 rows x 16 k reads rows at the padded 144-byte stride: the 16 lanes of a read group hit 16
// distinct 16-byte slots; the B fragment of 32 W rows x 16 k is two 256-byte runs of the tile-major image, 2 KiB apart:
// conflict-free), same barrier placement.  Item (ks16, mt) = 2 MFMAs (the wave's two 32-column N tiles); SwiGLU: N tile 0
// is the gate tile, N tile 1 the matching up tile.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

#ifdef MI_GEMM_M32_BUILD      // a debug build target (hipcc -DMI_GEMM_M32_BUILD, run with MI_GEMM_M32=1): measured 7 % slower than the 16 x 16 x 32 form, not in the shipped library
template <typename AT, bool SWIGLU>
__global__ __launch_bounds__(512) void gemm_tile256_m32_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;              // 2 x 4 waves
  const int c32 = lane & 31, h2 = lane >> 5;
  const int bm = blockIdx.x, bn = blockIdx.y;
  const int m0 = bm * BM2;
  const int nk = p.K / BK, nkw = p.kw / BK;
  const AT* x = (const AT*)p.x;
  const int ntiles_w = p.N / 16;
  auto a_img = [&](int buf) { return (AT*)(smem2 + (size_t)buf * A2_BYTES); };
  auto b_img = [&](int buf) { return smem2 + 2 * (size_t)A2_BYTES + (size_t)buf * B2_BYTES; };
  auto w_tile_of = [&](int s) -> int {
    int t;
    if constexpr (!SWIGLU) t = (bn * BN2) / 16 + s;
    else t = (bn * 128 + (s >> 2) * 32 + (s & 1) * 16 + ((s >> 1) & 1) * p.pair_offset) / 16;
    return min(t, ntiles_w - 1);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 areg[4], breg[4];
  const AT* arow[4];
  const char* brow[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 512 * i, row = c >> 3, kq = c & 7;
    aok[i] = m0 + row < p.M;
    arow[i] = x + (size_t)min(m0 + row, p.M - 1) * p.ldx + kq * 8;
    const int s = c >> 7, kb = (c >> 6) & 1, l = c & 63;
    brow[i] = (const char*)p.w + ((size_t)w_tile_of(s) * (p.kw / 32) + (size_t)kb) * 1024 + l * 16;
  }
  auto load_ab = [&](int ks) {
    const int ka = (ks * BK) % p.ka;
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[i] = *(const u32x4*)(arow[i] + ka);
#pragma unroll
    for (int i = 0; i < 4; ++i) breg[i] = *(const u32x4*)(brow[i] + (size_t)(ks % nkw) * 2048);
  };
  auto store_ab = [&](int buf) {
    AT* A = a_img(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 512 * i, row = c >> 3, kq = c & 7;
      *(u32x4*)&A[row * LDA + kq * 8] = aok[i] ? areg[i] : u32x4{0u, 0u, 0u, 0u};
    }
    unsigned char* B = b_img(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(B + (size_t)(tid + 512 * i) * 16) = breg[i];
  };

  // item it = (ks16 = it >> 2, mt = it & 3): A fragment of rows wm*128 + 32 mt + c32, k = 16 ks16 + 8 h2
  u32x4 af[AFR], bf[4][2];
  auto a_frag = [&](const AT* A, int it) {
    return *(const u32x4*)&A[(wm * 128 + (it & 3) * 32 + c32) * LDA + (it >> 2) * 16 + h2 * 8];
  };
  auto b_frags = [&](const unsigned char* B, int ks16, u32x4 (&dst)[2]) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int slot = wn * 4 + nt * 2 + (c32 >> 4);
      dst[nt] = *(const u32x4*)(B + (size_t)(((slot * 2 + (ks16 >> 1)) * 64) + (2 * (ks16 & 1) + h2) * 16 + (c32 & 15)) * 16);
    }
  };
  auto step = [&](int ks) {
    const int cur = ks & 1;
    load_ab(min(ks + 1, nk - 1));
    __builtin_amdgcn_sched_barrier(0);
    const AT* A = a_img(cur);
    const AT* An = a_img(cur ^ 1);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (it == SYNC_AT) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
      }
      const int nx = it + AFR - 1;
      af[nx % AFR] = nx < 16 ? a_frag(A, nx) : a_frag(An, nx - 16);
      if (it == 1) b_frags(b_img(cur), 1, bf[1]);
      if (it == 5) b_frags(b_img(cur), 2, bf[2]);
      if (it == 9) b_frags(b_img(cur), 3, bf[3]);
      if (it == SYNC_AT) b_frags(b_img(cur ^ 1), 0, bf[0]);        // (bf[0] is dead since item 3)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[it & 3][nt] = mfma32<AT>(af[it % AFR], bf[it >> 2][nt], acc[it & 3][nt]);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (it == 1 || it == 5 || it == 9 || it == SYNC_AT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (it == STORE_AT) {
        __builtin_amdgcn_sched_barrier(0);
        store_ab(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  load_ab(0);
  store_ab(0);
  __syncthreads();
  b_frags(b_img(0), 0, bf[0]);
#pragma unroll
  for (int i = 0; i < AFR - 1; ++i) af[i] = a_frag(a_img(0), i);
  for (int ks = 0; ks < nk; ++ks) step(ks);

  // ---- epilogue: lane (c32, h2), register r holds C[m = (r & 3) + 8 (r >> 2) + 4 h2][n = c32] of every 32 x 32 tile
  AT* out = (AT*)p.out;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
      if (m >= p.M) continue;
      if constexpr (SWIGLU) {
        const int n = bn * 128 + wn * 32 + c32;
        if (n >= p.pair_offset) continue;
        if (p.out32) { epi32_swiglu(p, m, n, acc[mt][0][r], acc[mt][1][r]); continue; }
        const float gt = (float)(AT)acc[mt][0][r], up = (float)(AT)acc[mt][1][r];
        const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
        const float sl = (float)(AT)(gt * sig);
        out[(size_t)m * p.ldo + n] = (AT)(sl * up);
      } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int n = bn * BN2 + wn * 64 + nt * 32 + c32;
          if (n >= p.N) continue;
          if (p.out32) { epi32_plain(p, m, n, acc[mt][nt][r]); continue; }
          const float y = (float)(AT)acc[mt][nt][r];
          if (p.epi == EPI_STORE) out[(size_t)m * p.ldo + n] = (AT)y;
          else {
            AT* hh = (AT*)p.resid;
            hh[(size_t)m * p.ldo + n] = (AT)((float)hh[(size_t)m * p.ldo + n] + y);
          }
        }
      }
    }
}
#endif  // MI_GEMM_M32_BUILD

// out[row][:] = w * cast_T(x32 * rsqrt(mean(x32^2) + eps))   (nn.RMSNorm); one wave per row
template <typename AT>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const AT* x, int ldx, const AT* w, AT* out, int ldo,
                                                           int rows, int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const AT* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  for (int k = lane * 8; k < H; k += 512) {
    const u32x4 v = *(const u32x4*)(xr + k);
    const AT* e = (const AT*)&v;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = (float)e[j]; ss = fmaf(f, f, ss); }
  }
  ss = wave_sum(ss);
  const float rs = 1.0f / sqrtf(ss / (float)H + eps);
  for (int k = lane * 8; k < H; k += 512) {
    u32x4 v = *(const u32x4*)(xr + k);
    const u32x4 wv = *(const u32x4*)(w + k);
    AT* e = (AT*)&v;
    const AT* we = (const AT*)&wv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const AT xn = (AT)((float)e[j] * rs);
      e[j] = (AT)((float)xn * (float)we[j]);
    }
    *(u32x4*)(out + (size_t)row * ldo + k) = v;
  }
}

// The same operator with one 256-thread workgroup per row and the row held in registers (NP 16-byte pieces per
// thread): for the few rows of a decode step (gemm_skinny.hip), where one wave per row means a handful of
// workgroups walking their rows in dependent L2 round trips (9 us at 32 x 4096; this one: one round trip).
template <typename AT, int NP>
__global__ __launch_bounds__(256) void rmsnorm_row_block_kernel(const AT* x, int ldx, const AT* w, AT* out, int ldo, int H,
                                                                float eps) {
  __shared__ float part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const AT* xr = x + (size_t)row * ldx;
  u32x4 v[NP], wv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int k = (tid + i * 256) * 8;
    const int kc = k < H ? k : 0;
    v[i] = *(const u32x4*)(xr + kc);
    wv[i] = *(const u32x4*)(w + kc);
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if ((tid + i * 256) * 8 < H) {
      const AT* e = (const AT*)&v[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)e[j]; ss = fmaf(f, f, ss); }
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  const float rs = 1.0f / sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int k = (tid + i * 256) * 8;
    if (k < H) {
      AT* e = (AT*)&v[i];
      const AT* we = (const AT*)&wv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const AT xn = (AT)((float)e[j] * rs);
        e[j] = (AT)((float)xn * (float)we[j]);
      }
      *(u32x4*)(out + (size_t)row * ldo + k) = v[i];
    }
  }
}

// float32 storage with the run-time logical rounding (PagedKVCache mode): out = rnd(w * rnd(x * rsqrt(mean(x^2) + eps)))
__global__ __launch_bounds__(256) void rmsnorm_row_block_f32_kernel(const float* x, int ldx, const float* w, float* out, int ldo,
                                                                    int H, float eps, int rnd) {
  __shared__ float part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  for (int k = tid * 4; k < H; k += 1024) {
    const f32x4 v = *(const f32x4*)(xr + k);
    ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  const float rs = 1.0f / sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)H + eps);
  for (int k = tid * 4; k < H; k += 1024) {
    const f32x4 v = *(const f32x4*)(xr + k), wv = *(const f32x4*)(w + k);
    f32x4 o;
    o.x = round_rt(round_rt(v.x * rs, rnd) * wv.x, rnd);
    o.y = round_rt(round_rt(v.y * rs, rnd) * wv.y, rnd);
    o.z = round_rt(round_rt(v.z * rs, rnd) * wv.z, rnd);
    o.w = round_rt(round_rt(v.w * rs, rnd) * wv.w, rnd);
    *(f32x4*)(out + (size_t)row * ldo + k) = o;
  }
}


// float32 activations in front of the tile GEMM (PagedKVCache mode): x (optionally RMS-normalised first, in float32) is
// split EXACTLY into three 16-bit terms, x = hi + mid + lo (hi = T(x), mid = T(x - hi), lo = T(x - hi - mid): 24
// mantissa bits), stored side by side as one row [hi | mid | lo] of 3 K elements.  The GEMM then walks W three times
// (GemmParams::kw): every product is exact in the float32 accumulator, so the result is a float32 dot product in
// another summation order -- the same arithmetic as the decode step's skinny_kernel<.., X32>.  One workgroup per row.
// terms = 2 (round 4, dense bf16 weights): [hi | lo] only, 16+ mantissa bits of x, TWO walks of W instead of three.  What
// that costs in accuracy was measured on the CPU before it was built (oracle/numerics.py X_SPLIT2): the
// float32-accumulating variants' deviation from the exact oracle does not move (mean 1.91352e-4 vs 1.91362e-4).
template <typename AT>
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* x, int ldx, const float* norm_w, float eps, AT* out,
                                                          int K, int terms) {
  __shared__ float part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * ldx;
  float rs = 1.0f;
  if (norm_w != nullptr) {
    float ss = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
      const f32x4 v = *(const f32x4*)(xr + k);
      ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    rs = 1.0f / sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)K + eps);
  }
  AT* o = out + (size_t)row * terms * K;
  if ((K & 7) == 0) {                                 // (uniform) 8 elements per thread and trip: 16-byte stores of every term
    for (int k = tid * 8; k < K; k += 2048) {
      f32x4 v[2] = {*(const f32x4*)(xr + k), *(const f32x4*)(xr + k + 4)};
      if (norm_w != nullptr) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 wv = *(const f32x4*)(norm_w + k + 4 * h);
          v[h].x = (v[h].x * rs) * wv.x; v[h].y = (v[h].y * rs) * wv.y; v[h].z = (v[h].z * rs) * wv.z; v[h].w = (v[h].w * rs) * wv.w;
        }
      }
      AT hi[8], mid[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = v[j >> 2][j & 3];
        hi[j] = (AT)xv;
        const float r1 = xv - (float)hi[j];
        mid[j] = (AT)r1;
        lo[j] = (AT)(r1 - (float)mid[j]);
      }
      *(uint4*)(o + k) = *(const uint4*)hi;
      *(uint4*)(o + K + k) = *(const uint4*)mid;
      if (terms == 3) *(uint4*)(o + 2 * K + k) = *(const uint4*)lo;      // (uniform)
    }
    return;
  }
  for (int k = tid * 4; k < K; k += 1024) {
    f32x4 v = *(const f32x4*)(xr + k);
    if (norm_w != nullptr) {
      const f32x4 wv = *(const f32x4*)(norm_w + k);
      v.x = (v.x * rs) * wv.x; v.y = (v.y * rs) * wv.y; v.z = (v.z * rs) * wv.z; v.w = (v.w * rs) * wv.w;
    }
    AT hi[4], mid[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = (AT)v[j];
      const float r1 = v[j] - (float)hi[j];
      mid[j] = (AT)r1;
      lo[j] = (AT)(r1 - (float)mid[j]);
    }
    *(uint2*)(o + k) = *(const uint2*)hi;
    *(uint2*)(o + K + k) = *(const uint2*)mid;
    if (terms == 3) *(uint2*)(o + 2 * K + k) = *(const uint2*)lo;      // (uniform)
  }
}

template <typename AT>
int launch_rmsnorm_block(const void* x, int ldx, const void* w, void* out, int ldo, int rows, int H, float eps, hipStream_t st) {
  const int np = (H / 8 + 255) / 256;
  const dim3 grid(rows), block(256);
#define GO(NP) hipLaunchKernelGGL((rmsnorm_row_block_kernel<AT, NP>), grid, block, 0, st, (const AT*)x, ldx, (const AT*)w, (AT*)out, ldo, H, eps)
  switch (np) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    default: return fail(MI_ERR_UNSUPPORTED, "rmsnorm (one workgroup per row): hidden size above 8192");
  }
#undef GO
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace

bool gemm_prefill_supported(const LinearW& W, const GemvCall& c, size_t rows) {
  if (c.force_v1 || c.rnd != RND_NONE || W.layout != 1) return false;
  if (rows < 32) return false;
  const bool dense = (W.wk == WK_BF16 && c.act == MI_BF16) || (W.wk == WK_F16 && c.act == MI_F16);
  const bool q4 = ((W.wk == WK_Q4_BF16 && c.act == MI_BF16) || (W.wk == WK_Q4_F16 && c.act == MI_F16)) && W.group == 64 &&
                  W.K % 128 == 0;                       // through a [hi | lo] 16-bit copy (launch_dequant_q4_hilo)
  const bool q8 = ((W.wk == WK_Q8_BF16 && c.act == MI_BF16) || (W.wk == WK_Q8_F16 && c.act == MI_F16)) && W.group == 64;
  // float32 activations on a bf16 model (PagedKVCache mode): x goes through launch_split3_rows first
  static const bool f32_ok = getenv("MI_GEMM_NO_F32") == nullptr;
  const bool x32 = f32_ok && c.act == MI_F32 && (W.wk == WK_BF16 || (W.wk == WK_Q4_BF16 && W.group == 64 && W.K % 128 == 0)) &&
                   c.ldx % 4 == 0 && W.K % 4 == 0;
  if (!dense && !q4 && !q8 && !x32) return false;
  if (W.K % BK != 0 || (!x32 && c.ldx % 8 != 0)) return false;
  if (c.epi == EPI_STORE_F32) return false;
  // an adapted matrix: the caller adds the LoRA term to the stored output afterwards (launch_lora_up_add)
  if ((W.lora_b[0] != nullptr || W.lora_b[1] != nullptr) && c.epi != EPI_STORE) return false;
  const int n = c.epi == EPI_SWIGLU ? c.pair_offset : W.N;
  return n % 16 == 0 && (c.epi != EPI_SWIGLU || c.pair_offset % 16 == 0);
}

int launch_rmsnorm_rows(const void* x, int ldx, const void* w, void* out, int ldo, int rows, int H, float eps, int act,
                        hipStream_t st, bool block_per_row, int rnd) {
  if (H % 8 != 0) return fail(MI_ERR_UNSUPPORTED, "rmsnorm_rows: hidden size must be a multiple of 8");
  if (act == MI_F32) {
    hipLaunchKernelGGL(rmsnorm_row_block_f32_kernel, dim3(rows), dim3(256), 0, st, (const float*)x, ldx, (const float*)w,
                       (float*)out, ldo, H, eps, rnd);
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  if (block_per_row && H <= 8192) {
    if (act == MI_BF16) return launch_rmsnorm_block<bf16>(x, ldx, w, out, ldo, rows, H, eps, st);
    if (act == MI_F16) return launch_rmsnorm_block<f16>(x, ldx, w, out, ldo, rows, H, eps, st);
  }
  const dim3 grid((rows + 3) / 4), block(256);
  if (act == MI_BF16)
    hipLaunchKernelGGL(rmsnorm_rows_kernel<bf16>, grid, block, 0, st, (const bf16*)x, ldx, (const bf16*)w, (bf16*)out, ldo, rows, H, eps);
  else if (act == MI_F16)
    hipLaunchKernelGGL(rmsnorm_rows_kernel<f16>, grid, block, 0, st, (const f16*)x, ldx, (const f16*)w, (f16*)out, ldo, rows, H, eps);
  else return fail(MI_ERR_UNSUPPORTED, "rmsnorm_rows: 16-bit activations only");
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// c.pro must be PRO_NONE here (the caller runs launch_rmsnorm_rows first); rows = total rows of x.
// int4 weights: `scratch` (>= dequant_hilo_bytes(N, K)) receives the [hi | lo] copy first.
size_t split3_bytes(size_t rows, int K) { return rows * 3 * (size_t)K * 2; }

int launch_split3_rows(const void* x, int ldx, const void* norm_w, float eps, void* out, int rows, int K, hipStream_t st,
                       int terms) {
  if (K % 4 != 0 || ldx % 4 != 0) return fail(MI_ERR_UNSUPPORTED, "split3_rows: K and the row stride must be multiples of 4");
  if (terms != 2 && terms != 3) return fail(MI_ERR_INVALID, "split3_rows: two or three terms");
  hipLaunchKernelGGL(split3_rows_kernel<bf16>, dim3(rows), dim3(256), 0, st, (const float*)x, ldx, (const float*)norm_w, eps,
                     (bf16*)out, K, terms);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_gemm_prefill(const LinearW& W, const GemvCall& c, size_t rows, hipStream_t st, void* scratch, void* splitk_ws,
                        size_t splitk_cap) {
  if (c.pro != PRO_NONE) return fail(MI_ERR_INVALID, "gemm_prefill: normalise the rows first");
  GemmParams p{};
  p.ksplit = 1;
  p.x = c.x; p.ldx = c.ldx; p.M = (int)rows; p.w = W.w; p.N = W.N; p.K = W.K; p.ka = W.K; p.kw = W.K;
  if (wk_is_quant(W.wk)) {
    if (scratch == nullptr) return fail(MI_ERR_INVALID, "gemm_prefill: int4 weights need a dequantisation scratch buffer");
    MI_TRY(launch_dequant_q4_hilo(W, scratch, st));
    p.w = scratch; p.K = 2 * W.K; p.kw = 2 * W.K;
  }
  const bool x32 = c.act == MI_F32;        // c.x is the [hi | mid | lo] bf16 image of launch_split3_rows, c.ldx = 3 K
  if (x32) {
    // c.kx > 0: W is the persistent [hi | lo] bf16 copy of an f16 matrix (2 kx columns, launch_f16_to_hilo) -- the same
    // walk as over an int4 matrix's on-the-fly [hi | lo] copy: x = [hi | mid | lo] (3 kx) against W (2 kx) over 6 kx steps
    const bool hilo = c.kx > 0 && !wk_is_quant(W.wk);
    const int kt = hilo ? c.kx : W.K;      // the true K
    if (hilo && (W.K != 2 * c.kx || W.wk != WK_BF16)) return fail(MI_ERR_INVALID, "gemm_prefill: a [hi | lo] matrix has 2 kx bf16 columns");
    const bool pair = wk_is_quant(W.wk) || hilo;     // W itself is a [hi | lo] pair: 3 x-terms against 2 W-terms meet over 6 kt steps
    const int terms = c.ldx == 2 * kt ? 2 : 3;
    if (c.ldx != terms * kt || (pair && terms != 3))
      return fail(MI_ERR_INVALID, "gemm_prefill: float32 activations must be split first (launch_split3_rows; two terms: dense bf16 weights only)");
    p.ka = terms * kt; p.K = (pair ? 6 : terms) * kt; p.out32 = 1;
    if (hilo) p.kw = 2 * kt;
  }
  p.epi = c.epi; p.out = c.out; p.ldo = c.ldo; p.resid = c.resid; p.pair_offset = c.pair_offset;
  const bool sw = c.epi == EPI_SWIGLU;
  const int ncols = sw ? c.pair_offset : W.N;
  static const bool small_only = getenv("MI_GEMM_TILE128") != nullptr;        // A/B: always the 128 x 128 kernel
  static const bool b_in_lds = getenv("MI_GEMM_B_DIRECT") == nullptr;         // A/B: set = W fragments straight from global
  // The 256 x 256 tile only where its grid fills the chip: at 8 x 1024 rows every linear has >= 512 blocks, but one prompt
  // of 256..1024 tokens gives N/256 x (1..4) blocks -- 16..64 for N = 4096 -- and measured 25 / 25 / 30 ms for 256 / 512 /
  // 1024 tokens against 14 / 17 / 23 ms on the 128 x 128 tile (Mistral-7B bf16; tools/debug/prefill_sweep.py).
  const long blocks256 = (long)(((int)rows + BM2 - 1) / BM2) * ((ncols + (sw ? 128 : BN2) - 1) / (sw ? 128 : BN2));
  const char* dma_env = getenv("MI_GEMM_DMA");         // A/B and the bit-equality test: 0 = the register-staged tile (read per call)
  const bool dma = dma_env == nullptr || atoi(dma_env) != 0;
  // too few 256 x 256 tiles for the chip (one prompt of 256..4095 rows against N = 4096 / 6144): the LDS-DMA tile with K
  // split over up to 8 workgroups, each >= 8 K tiles (plain / residual epilogues; SwiGLU launches have enough tiles or
  // stay on the 128 x 128 kernel).  MI_GEMM_DMA_SPLITK=0 switches it off (A/B).
  int ks256 = 1;
  const char* sk_env = getenv("MI_GEMM_DMA_SPLITK");     // (read per call: the tests switch it)
  const bool splitk256 = sk_env == nullptr || atoi(sk_env) != 0;
  if (rows >= 256 && !small_only && dma && splitk256 && !sw && splitk_ws != nullptr && W.N % 4 == 0) {
    // one 128-KiB workgroup per CU: the launch runs in rounds of n_cu workgroups.  Pick the split whose last round is
    // fullest, charging 2 % per extra slice for its partial tile and its share of the reduce (q|k|v at 1024 rows: 96
    // tiles -> 5 slices = 480 workgroups, 1.9 rounds; o_proj: 64 tiles -> 4 slices = one full round)
    const int ncu = gemv_cu_count();
    double best = 0.0;
    for (int ks = 1; ks <= 8; ++ks) {
      if (ks > 1 && ((p.K / BK) / ks < 8 || (size_t)ks * rows * W.N * sizeof(float) > splitk_cap)) break;
      const long wgs = blocks256 * ks, rounds = (wgs + ncu - 1) / ncu;
      const double score = (double)wgs / (double)(rounds * ncu) * (1.0 - 0.02 * (ks - 1));
      if (score > best + 1e-9) { best = score; ks256 = ks; }
    }
  }
  if (rows >= 256 && !small_only && (blocks256 >= 192 || ks256 > 1)) {  // both operands through LDS (256 x 256 tile)
    const int bn = sw ? 128 : BN2;
    dim3 grid2(((int)rows + BM2 - 1) / BM2, (ncols + bn - 1) / bn), block2(512);
    if (ks256 > 1) { grid2.z = ks256; p.ksplit = ks256; p.ws = (float*)splitk_ws; }
#ifdef MI_GEMM_M32_BUILD
    static const bool m32 = getenv("MI_GEMM_M32") != nullptr;               // A/B: the 32x32x16 form of the tile
#endif
    const bool use_dma = dma && p.K >= 2 * BK;
#ifdef MI_GEMM_M32_BUILD
#define GO256_M32(T, S) else if (m32) { auto k = gemm_tile256_m32_kernel<T, S>; \
        MI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES)); \
        hipLaunchKernelGGL(k, grid2, block2, LDS2_BYTES, st, p); }
#else
#define GO256_M32(T, S)
#endif
#define GO256(T, S) do { \
      if (use_dma) { auto k = gemm_dma256_kernel<T, S>; \
        MI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DMA_BUF)); \
        hipLaunchKernelGGL(k, grid2, block2, 2 * DMA_BUF, st, p); } \
      GO256_M32(T, S) \
      else if (b_in_lds) { auto k = gemm_tile256_kernel<T, S, false>; \
        MI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES)); \
        hipLaunchKernelGGL(k, grid2, block2, LDS2_BYTES, st, p); } \
      else { auto k = gemm_tile256_kernel<T, S, true>; \
        MI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * A2_BYTES)); \
        hipLaunchKernelGGL(k, grid2, block2, 2 * A2_BYTES, st, p); } } while (0)
    if (c.act == MI_BF16 || x32) { if (sw) GO256(bf16, true); else GO256(bf16, false); }
    else { if (sw) GO256(f16, true); else GO256(f16, false); }
#undef GO256
    MI_HIP(hipGetLastError());
    if (p.ksplit > 1) {
      const size_t total4 = rows * (size_t)W.N / 4;
      const dim3 rg((unsigned)((total4 + 255) / 256));
      if (c.act == MI_F16) hipLaunchKernelGGL(splitk_epilogue_kernel<f16>, rg, dim3(256), 0, st, p);
      else hipLaunchKernelGGL(splitk_epilogue_kernel<bf16>, rg, dim3(256), 0, st, p);
      MI_HIP(hipGetLastError());
    }
    return MI_OK;
  }
  dim3 grid(((int)rows + BM - 1) / BM, (ncols + (sw ? 64 : BN) - 1) / (sw ? 64 : BN)), block(256);
  // too few (M, N) blocks for 256 CUs: split K (plain / residual epilogues; W.N % 4 == 0 for the reduce kernel)
  static const bool no_splitk = getenv("MI_GEMM_NO_SPLITK") != nullptr;
  const long blocks = (long)grid.x * grid.y;
  // ~512 workgroups of 256 threads in all (two per CU): measured against a target of 256 -- 10.5 vs 11.6 ms at 384 rows,
  // 20.0 vs 21.4 at 768, 21.4 vs 22.4 at 1024 (Mistral-7B bf16, whole prefill call)
  static const int target = getenv("MI_GEMM_SPLITK_TARGET") ? atoi(getenv("MI_GEMM_SPLITK_TARGET")) : 512;   // A/B
  if (!sw && !no_splitk && splitk_ws != nullptr && blocks < target && W.N % 4 == 0) {
    int ks = (int)std::min<long>(8, std::max<long>(1, target / blocks));
    ks = std::min(ks, (p.K / BK) / 8);                       // at least 8 K steps per slice
    while (ks > 1 && (size_t)ks * rows * W.N * sizeof(float) > splitk_cap) --ks;
    if (ks > 1) { p.ksplit = ks; p.ws = (float*)splitk_ws; grid.z = ks; }
  }
  if (c.act == MI_BF16 || x32) {
    if (sw) hipLaunchKernelGGL((gemm_tile_kernel<bf16, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_tile_kernel<bf16, false>), grid, block, 0, st, p);
  } else {
    if (sw) hipLaunchKernelGGL((gemm_tile_kernel<f16, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_tile_kernel<f16, false>), grid, block, 0, st, p);
  }
  MI_HIP(hipGetLastError());
  if (p.ksplit > 1) {
    const size_t total4 = rows * (size_t)W.N / 4;
    const dim3 rg((unsigned)((total4 + 255) / 256));
    if (c.act == MI_F16) hipLaunchKernelGGL(splitk_epilogue_kernel<f16>, rg, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(splitk_epilogue_kernel<bf16>, rg, dim3(256), 0, st, p);
    MI_HIP(hipGetLastError());
  }
  return MI_OK;
}

}  // namespace mi
