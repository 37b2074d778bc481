// gemm_q4.hip -- nn.QuantizedLinear (MLX-affine int4, group 64: mlx_parallm/utils.py:679-690; call sites llama.py:64-67,93,143,
// 160-165,250-252; qwen3.py:37-40,63,115) for the decode step of 17..128 sequences: y[M][N] = x[M][K] . W_hat[N][K]^T with
// W_hat = scale * q + bias per group of 64, residual / SwiGLU / float32-logits / LoRA epilogues.  Round 4's rewrite of the
// quantised form of gemm_skinny.hip for these row counts (BASELINE config 5: Qwen3-14B int4 + LoRA, 64 sequences).
//
// What the stamps and counters of rounds 2-3 said about skinny_kernel<.., 4, MT = 4>: ~10 vector instructions
// per MFMA -- 45 % of them re-staging x (nibble-order permutation, per-group sums of x, LDS writes) in EVERY workgroup and
// chunk, the rest unpack + two FMAs per (group, row tile, accumulator element) -- at two waves per SIMD; K split over
// workgroups (float32 partial tiles = 0.9 x the weights' bytes at 64 rows); 408 workgroups of which one fits per CU = two
// rounds; a workgroup streams 8-12 GB/s whatever the grid.  This kernel keeps the arithmetic
// (y += s * sum((16 + q) x) + (b - 16 s) * sum(x) per group, the (16 + q) products on v_mfma_f32_16x16x32) and changes the shape:
//
//   * x is prepared ONCE per launch (q4_prep_kernel; it also applies the RMSNorm in front of q|k|v and gate|up, i.e. it
//     REPLACES the rmsnorm_row_block launch): MFMA A fragments in the nibble order of unpack_q4, fragment-major, plus the
//     per-(group, row) sums of x.  A workgroup's staging of a K chunk is then a lane-linear copy (global_load_dwordx4 ->
//     ds_write_b128, no vector arithmetic), every fragment read a conflict-free lane-linear ds_read_b128.
//   * K is split over the WAVES of a workgroup, not over workgroups: wave (tw, kw) owns tile (pair) tw of the workgroup
//     and every KW-th 128-k block; the KW partial accumulators meet in LDS once, at the end -- no float32 partial tiles in
//     memory, one pass over K.  (A split over workgroups remains for the narrow linears, o_proj / down_proj.)
//   * a wave carries TN = 2 tiles (gate + up for SwiGLU, two neighbours otherwise): every A fragment read feeds two MFMAs.
//   * rows above 32 may run as slabs of 32 (twice the workgroups; the second slab finds the weights in L2).
//   * <= 128 VGPRs: two workgroups per CU.
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gemv_phase.h"

namespace mi {

namespace {

using namespace gemv;

constexpr int Q4_NWMAX = 12;      // waves per workgroup, at most: TW x KW compute waves + NS staging waves (three per SIMD: <= 168 VGPRs)

struct Q4Params {
  const void* xq; const float* sx;      // prepared activations: fragments [4 K/128][Mt][64 lanes][16 B], sums [K/64][16 Mt]
  int M, Mt;
  const void* w; int N, K;
  int epi; void* out; int ldo; void* resid; int pair_offset;
  int ntu;            // tile units: tile pairs (SwiGLU: gate tile t + up tile t; else tiles 2t, 2t + 1)
  int ntiles;         // 16-row tiles of W (SwiGLU: per half)
  int TW, KW, NS;     // compute waves = TW tile units x KW K lanes; NS staging waves behind them
  int nblk;           // K / 128
  int ksplit;         // K slices over workgroups
  int nslab, nunits;  // row slabs of 16 MT rows; units = tile groups x ksplit
  float* ws; unsigned* ctr;
  unsigned long long* trace;   // MI_Q4_DBG & 8: [8 workgroups][12 waves][64] shader-clock stamps (debug)
  int dbg;            // timing-only ablations (MI_Q4_DBG; results wrong on purpose): 1 = no MFMA / FMA work, 2 = no staging of x, 4 = no weight re-issue
  const float* lora_t; int lora_t_ld;
  const float* lora_b0; const float* lora_b1;
  int lora_row0_0, lora_n_0, lora_rank_0; float lora_scale_0;
  int lora_row0_1, lora_n_1, lora_rank_1; float lora_scale_1;
};

// ---- x -> fragment-major nibble order + group sums (+ RMSNorm).  One 256-thread workgroup per row (rows padded to 16 Mt:
// the padding rows are written as zeros).  Piece = 8 consecutive k of one row; its 16 bytes land where lane (m % 16, gg) of
// fragment F = ((k / 128) * 2 + sg) * 2 + t reads them: dword d = (k % 64) / 8 of its quantisation group is consumed by lane
// group gg = {0: 0, 4: 1, 2: 2, 6: 3}[d - t] in MFMA step t = d & 1 (gemv_phase.h: frag_slot / unpack_q4).
template <typename AT>
__global__ __launch_bounds__(256) void q4_prep_kernel(const AT* x, int ldx, int M, int Mt, int K, const AT* norm_w, float eps,
                                                      AT* xq, float* sx) {
  __shared__ float part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int Mpad = 16 * Mt;
  const bool live = row < M;
  const AT* xr = x + (size_t)(live ? row : 0) * ldx;
  float rs = 1.0f;
  if (norm_w != nullptr) {
    float ss = 0.f;
    if (live) {
      for (int k = tid * 8; k < K; k += 2048) {
        const u32x4 v = *(const u32x4*)(xr + k);
        const AT* e = (const AT*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)e[j]; ss = fmaf(f, f, ss); }
      }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    rs = 1.0f / sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)K + eps);
  }
  for (int k = tid * 8; k < K; k += 2048) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (live) {
      v = *(const u32x4*)(xr + k);
      if (norm_w != nullptr) {
        const u32x4 wv = *(const u32x4*)(norm_w + k);
        AT* e = (AT*)&v;
        const AT* we = (const AT*)&wv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const AT xn = (AT)((float)e[j] * rs);                // cast_T(x32 * rsqrt(..))
          e[j] = (AT)((float)xn * (float)we[j]);               // w * (.) in T
        }
      }
    }
    const AT* e = (const AT*)&v;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += (float)e[j];
    AT t2[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { t2[2 * q] = e[q]; t2[2 * q + 1] = e[q + 4]; }
    sum = lane8_sum(sum);                                      // the 8 pieces of a 64-group sit in 8 consecutive threads
    const int dd = (k >> 3) & 15, kb = k >> 7;
    const int sg = dd >> 3, d = dd & 7, t = d & 1, ee = d - t;
    const int gg = (ee == 0) ? 0 : (ee == 4) ? 1 : (ee == 2) ? 2 : 3;
    const int F = (kb * 2 + sg) * 2 + t;
    *(u32x4*)((char*)xq + ((((size_t)F * Mt + (row >> 4)) * 64) + gg * 16 + (row & 15)) * 16) = *(const u32x4*)t2;
    if ((tid & 7) == 0) sx[(size_t)(k >> 6) * Mpad + row] = sum;
  }
}

#define Q4_STAMP(i) do { if (p.trace != nullptr && blockIdx.x < 8 && lane == 0 && (i) < 64) \
    p.trace[((size_t)blockIdx.x * 12 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

template <typename AT, int MT, bool SWIGLU>
__global__ __launch_bounds__(Q4_NWMAX * 64, 3) void q4_kernel(const Q4Params p) {
  constexpr int TN = 2, MB = 16 * MT;
  // Registers: <= 168 (three waves per SIMD: one workgroup of up to 12 waves per CU); three weight blocks in flight per
  // wave and tile.
  constexpr int UK = 3;
  constexpr bool BOTH = true;               // both tiles of the unit share every A-fragment read (false: one tile at a time, 8 registers less)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int last_sh;

  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (provably uniform: piece / tile arithmetic stays scalar)
  const int TW = p.TW, KW = p.KW, NC = TW * KW;
  // (the staging waves are the FIRST waves of the workgroup: a 12-wave workgroup's last waves start ~3000 cycles after its
  // first, and the first chunk of x is what everybody waits for)
  const bool stager = wave < p.NS;                           // (uniform) staging waves: x only, never a weight load (see below)
  const int cw = stager ? 0 : wave - p.NS;
  const int kw = cw / TW, tw = cw - kw * TW;
  // blocks [16 j, 16 j + 8) = slab 0 of units 8 j .. 8 j + 7, the next 8 = slab 1, ... (slabs of a unit share an XCD under
  // round-robin placement: the later reader finds the weights in that XCD's L2 -- speed only)
  int bid = blockIdx.x, slab = 0;
  if (p.nslab > 1) {
    const int per = 8 * p.nslab, blk = bid / per, rem = bid - blk * per;
    slab = rem >> 3;
    bid = blk * 8 + (rem & 7);
  }
  if (bid >= p.nunits) return;                               // (grid padding: the whole workgroup leaves before any barrier)
  const int mt0 = slab * MT, row0 = mt0 * 16;
  const int grp = bid / p.ksplit, s = bid - grp * p.ksplit;
  const int tu_raw = grp * TW + tw;
  const bool valid = tu_raw < p.ntu;
  const int tu = valid ? tu_raw : p.ntu - 1;                 // a spare wave re-streams the last unit and stores nothing
  int tile[TN];
  if constexpr (SWIGLU) { tile[0] = tu; tile[1] = tu + (p.pair_offset >> 4); }
  else { tile[0] = min(2 * tu, p.ntiles - 1); tile[1] = min(2 * tu + 1, p.ntiles - 1); }
  const int b0 = (s * p.nblk) / p.ksplit, b1 = ((s + 1) * p.nblk) / p.ksplit;     // this workgroup's blocks
  const int nch = (b1 - b0 + KW - 1) / KW;                                        // chunks of KW blocks

  // Weight stream: block b of tile t at w + (t nblk + b) 1152; the wave walks b0 + kw, b0 + kw + KW, ...  Plain pointer
  // increments, no clamps: a block past the slice's end is loaded and never multiplied, and the matrix buffer is padded for
  // the loads past its last tile (repack.hip: tiled_bytes).
  const char* wp[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) wp[a] = (const char*)p.w + ((size_t)tile[a] * p.nblk + b0 + kw) * 1152;    // (wave-uniform: scalar registers)
  const int lane16 = lane * 16, c16x4 = c16 * 4;             // codes of the lane; scales of row c16 at + 1024 (biases 64 bytes further on)
  const size_t wstep = (size_t)KW * 1152;

  u32x4 wr[TN][UK];
  uint32_t sr[TN][UK], br[TN][UK];
  f32x4 acc[TN][MT];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The compute waves' weight loads are issued in inline asm and waited for with EXACT counts.  Left to hipcc, the wait in
  // front of a slot's first use came out as vmcnt(4) -- its wait-count pass merges states over the loop's back edge and the
  // `block is inside the slice` branch -- i.e. "all but 4 of my loads have landed", including the two slots just issued for
  // later chunks: the ring of UK blocks collapsed to less than one and every chunk waited an HBM round trip.  These waves
  // issue no other vector-memory instruction inside the loop, so "the slot issued UK chunks ago has landed" is exactly
  // vmcnt((UK - 1) x 6).  (cdna_hip_programming.md 5.7: "=v" loads, then a wait statement that names every destination.)
  auto issue = [&](int slot) {
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %3, %5 nt\n\t"
                   "global_load_dword %1, %4, %5 offset:1024\n\t"
                   "global_load_dword %2, %4, %5 offset:1088"
                   : "=v"(wr[a][slot]), "=v"(sr[a][slot]), "=v"(br[a][slot])
                   : "v"(lane16), "v"(c16x4), "s"(wp[a])
                   : "memory");
      wp[a] += wstep;
    }
  };
  // the loads of `slot` have landed when at most `left` younger loads of this wave are outstanding
  auto landed = [&](int slot, auto left) {
    static_assert(TN == 2, "two tiles per wave");
    asm volatile("s_waitcnt vmcnt(%c6)"
                 : "+v"(wr[0][slot]), "+v"(sr[0][slot]), "+v"(br[0][slot]), "+v"(wr[1][slot]), "+v"(sr[1][slot]), "+v"(br[1][slot])
                 : "i"(decltype(left)::value)
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- staging, by the NS waves behind the compute waves.  WHY OTHER WAVES: a wave's vector-memory results come back in
  // issue order.  Staged by the compute waves (rounds 1-3, and this kernel's first version), the chunk's x loads -- L2 hits --
  // queue behind the weight loads the wave has just issued for blocks it needs two or three chunks later -- HBM round trips --
  // so EVERY chunk waits one HBM latency for its activations: timing-only builds showed staging, weight stream and arithmetic
  // each adding its whole time to the launch (nothing overlapped), 2.5 us per 256-k chunk.  A staging wave's queue holds
  // nothing but x.
  // Chunk c = KW blocks x 4 fragments x MT row tiles (1 KiB each) + 1 KiB of group sums [2 KW][MB]; piece P = sw + NS j goes
  // to staging wave sw.  One wave-uniform pointer per piece, advanced by the chunk stride (the preparation buffers are padded,
  // so the chunk behind the slice's last one is readable).
  const int NP = KW * 4 * MT + 1;                            // pieces per chunk
  constexpr int NPC = MT >= 3 ? 17 : 9;                      // pieces per staging thread, at most: NP <= NPC NS (the host's plans see to it)
  const int chunk_bytes = (NP + 1) * 1024;                   // (+ a KiB that takes the stores of the pieces past the end: no branch in the copy)
  const int Mpad = 16 * p.Mt;
  const int NS = p.NS, sw = wave;
  const size_t xstep = (size_t)KW * 4 * p.Mt * 1024, sxstep = (size_t)2 * KW * Mpad * sizeof(float);
  const int sxi = lane * 4, sx_g = min(sxi / MB, 2 * KW - 1), sx_m = sxi % MB;   // the group sums: floats [4 lane, 4 lane + 4) of [2 KW][MB]
  const int sxoff = (sx_g * Mpad + sx_m) * 4;
  const char* xsrc[NPC];                                     // (wave-uniform) source of piece j in chunk 0
  int xdst[NPC];                                             // its KiB in a chunk buffer
  size_t xadv[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int P = sw + NS * j, Pc = min(P, NP - 2);
    const int kbi = Pc / (4 * MT), f = (Pc / MT) & 3, mt = Pc % MT;
    const bool sums = P == NP - 1;
    xsrc[j] = sums ? (const char*)(p.sx + (size_t)(2 * b0) * Mpad + row0)
                   : (const char*)p.xq + ((size_t)(b0 + kbi) * 4 + f) * p.Mt * 1024 + (size_t)(mt0 + mt) * 1024;
    xadv[j] = sums ? sxstep : xstep;
    xdst[j] = (P < NP ? P : NP) * 1024;
  }
  // STRAIGHT-LINE: nine loads, nine stores, nothing under a branch
  auto stage_chunk = [&](unsigned char* buf) {
    u32x4 xv[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const bool sums = sw + NS * j == NP - 1;               // (uniform)
      xv[j] = *(const u32x4*)(xsrc[j] + (sums ? sxoff : lane * 16));
      xsrc[j] += xadv[j];
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) *(u32x4*)(buf + xdst[j] + lane * 16) = xv[j];
  };

  unsigned char* cur = smem;
  unsigned char* nxt = smem + chunk_bytes;
  const int frag0 = kw * 4 * MT * 1024 + lane * 16;          // this wave's block inside a chunk buffer
  const int sxa0 = (NP - 1) * 1024 + ((kw * 2 + (g & 1)) * MB + c16) * 4;    // bias MFMA, A operand: sum(x) of row c16, group g & 1
  const int half_sh = (g & 1) ? 0 : 16;                      // 16-bit element g & 1 of a scale / bias dword -> the top of a float
  const bool klane = g < 2;                                  // lanes that carry a k index of the bias MFMA (2 of its 4)

  // One 128-k block of the wave's tiles against all row tiles:
  //   y += s_g * sum_k (16 + q) x   per quantisation group g (two 16 x 16 x 32 MFMAs + 4 FMAs per row tile and tile), and
  //   y += sum_g (b_g - 16 s_g) * sum_k x   for both groups by ONE v_mfma_f32_16x16x4_f32 per row tile and tile: A = the
  //   prepared sums of x (row on the lane, group on the k index), B = b - 16 s (column on the lane): the 8 FMAs per (group,
  //   row tile, tile) of the bias term were half of this loop's vector instructions -- and the loop is bound by those.
  // (gemv_phase.h Phase::mfma_u: after the swap {x,y} = quant group A, {z,w} = group B in every lane)
  auto block_mfma = [&](int slot, const unsigned char* buf) {
    uint32_t dw[TN][4];
    float bbv[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const u32x4 v = wr[a][slot];
      auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
      dw[a][0] = r0[0]; dw[a][1] = r1[0]; dw[a][2] = r0[1]; dw[a][3] = r1[1];
      float sv, bv;
      if constexpr (std::is_same<AT, bf16>::value) {
        sv = __uint_as_float((sr[a][slot] << half_sh) & 0xffff0000u);
        bv = __uint_as_float((br[a][slot] << half_sh) & 0xffff0000u);
      } else {
        sv = (float)__builtin_bit_cast(f16, (unsigned short)(sr[a][slot] >> (16 - half_sh)));
        bv = (float)__builtin_bit_cast(f16, (unsigned short)(br[a][slot] >> (16 - half_sh)));
      }
      bbv[a] = klane ? fmaf(-Magic<AT>::offs, sv, bv) : 0.f;
    }
    float sxa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      sxa[mt] = *(const float*)(buf + sxa0 + mt * 64);
#pragma unroll
      for (int a = 0; a < TN; ++a) acc[a][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sxa[mt], bbv[a], acc[a][mt], 0, 0, 0);
    }
    // The 2 MT (64-group, row tile) steps as a software pipeline over the x fragments: the two LDS reads of step i + 1 are
    // issued in front of the MFMAs of step i.  (Two steps ahead spills at 4 row tiles: 168 registers + scratch.)  (Left to itself hipcc emitted "ds_read x 2, s_waitcnt, MFMAs" eight times per
    // block -- every LDS round trip exposed, with three waves per SIMD to cover it.)
    auto rd = [&](int idx, u32x4& a0, u32x4& a1) {
      const int sg = idx / MT, mt = idx % MT;
      a0 = *(const u32x4*)(buf + frag0 + ((sg * 2 + 0) * MT + mt) * 1024);
      a1 = *(const u32x4*)(buf + frag0 + ((sg * 2 + 1) * MT + mt) * 1024);
    };
    u32x4 c0, c1, n0, n1;
    rd(0, c0, c1);
    u32x4 wq[TN][2];
    float sc[TN];
#pragma unroll
    for (int idx = 0; idx < 2 * MT; ++idx) {
      const int sg = idx / MT, mt = idx % MT;
      if (idx + 1 < 2 * MT) rd(idx + 1, n0, n1);
      if (mt == 0) {
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          wq[a][0] = unpack_q4<AT>(dw[a][sg * 2 + 0]);
          wq[a][1] = unpack_q4<AT>(dw[a][sg * 2 + 1]);
          // (integer arithmetic on the whole dword: a type-punned view of an asm load's destination lets the compiler split the
          // register where it is DEFINED and touch it in front of the wait)
          if constexpr (std::is_same<AT, bf16>::value) sc[a] = __uint_as_float(sg ? (sr[a][slot] & 0xffff0000u) : (sr[a][slot] << 16));
          else sc[a] = (float)__builtin_bit_cast(f16, (unsigned short)(sr[a][slot] >> (16 * sg)));
        }
      }
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = mfma16<AT>(c0, wq[a][0], d);
        d = mfma16<AT>(c1, wq[a][1], d);
        acc[a][mt].x = fmaf(sc[a], d.x, acc[a][mt].x);
        acc[a][mt].y = fmaf(sc[a], d.y, acc[a][mt].y);
        acc[a][mt].z = fmaf(sc[a], d.z, acc[a][mt].z);
        acc[a][mt].w = fmaf(sc[a], d.w, acc[a][mt].w);
      }
      c0 = n0; c1 = n1;
      __builtin_amdgcn_sched_barrier(0);
    }
    // The operand registers of the float32 MFMAs stay reserved to the end of the block.  Found on the 16-row instantiation:
    // hipcc re-used them for the unpack right behind the MFMA (a shift into the B operand's register one instruction later)
    // and the tile came out as garbage -- whichever write-after-read rule of v_mfma_f32_16x16x4_f32 that breaks, it is not
    // one the compiler pads.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(sxa[mt]));
#pragma unroll
    for (int a = 0; a < TN; ++a) asm volatile("" :: "v"(bbv[a]));
  };

  // ================= the slice.  Compute waves: chunk c is multiplied, the block UK chunks ahead is issued; staging waves:
  // chunk c + 1 goes into the other buffer (free since the barrier that closed chunk c - 1).  One barrier per chunk for all.
  const int nloop = (nch + UK - 1) / UK * UK;
  Q4_STAMP(0);
  if (stager) {
    stage_chunk(cur);
    __syncthreads();
    Q4_STAMP(1);
    for (int c = 0; c < nloop; ++c) {
      if (!(p.dbg & 2)) stage_chunk(nxt);
      Q4_STAMP(2 + 4 * c);
      __syncthreads();
      Q4_STAMP(5 + 4 * c);
      unsigned char* t = cur; cur = nxt; nxt = t;
    }
  } else {
#pragma unroll
    for (int u = 0; u < UK; ++u) issue(u);
    __syncthreads();
    Q4_STAMP(1);
    for (int c = 0; c < nloop; c += UK) {
#pragma unroll
      for (int ci = 0; ci < UK; ++ci) {
        const int cc = c + ci;
        landed(ci, std::integral_constant<int, (UK - 1) * 6>{});
        Q4_STAMP(2 + 4 * cc);
        if (cc < nch && b0 + cc * KW + kw < b1 && !(p.dbg & 1)) block_mfma(ci, cur);     // (uniform per wave)
        __builtin_amdgcn_sched_barrier(0);
        Q4_STAMP(3 + 4 * cc);
        issue(ci);
        __builtin_amdgcn_sched_barrier(0);
        Q4_STAMP(4 + 4 * cc);
        __syncthreads();                                     // the next chunk is complete; this chunk's buffer is free
        Q4_STAMP(5 + 4 * cc);
        unsigned char* t = cur; cur = nxt; nxt = t;
      }
    }
    // the blocks issued past the slice's end are still in flight: their registers stay reserved until they have landed
#pragma unroll
    for (int u = 0; u < UK; ++u) landed(u, std::integral_constant<int, 0>{});
  }

  // ================= the KW partial accumulators of a tile unit meet in LDS, in K-lane order
  if (KW > 1) {
    f32x4* red = (f32x4*)smem;                               // (both chunk buffers are free: the loop ended on a barrier)
    if (kw > 0 && !stager) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) red[(((kw - 1) * TW + tw) * (TN * MT) + a * MT + mt) * 64 + lane] = acc[a][mt];
    }
    __syncthreads();
    if (kw == 0 && !stager) {
      for (int k2 = 1; k2 < KW; ++k2) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = red[(((k2 - 1) * TW + tw) * (TN * MT) + a * MT + mt) * 64 + lane];
            acc[a][mt].x += v.x; acc[a][mt].y += v.y; acc[a][mt].z += v.z; acc[a][mt].w += v.w;
          }
      }
    }
  }
  Q4_STAMP(62);
  const bool owner = kw == 0 && valid && !stager;            // the wave that holds the unit's sums

  // ================= K split over workgroups (narrow linears): as gemm_skinny.hip -- write-through partial tiles, every wave
  // drains its stores, one counter per (tile group, slab), the last arriver adds the slices in slice order
  if (p.ksplit > 1) {
    if (owner) {
      unsigned long long* wp = (unsigned long long*)(p.ws + (((size_t)((slab * p.ksplit + s) * p.ntu + tu) * (TN * MT)) * 64 + lane) * 4);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[a][mt];
          unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
          unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
          __hip_atomic_store(wp + (size_t)(a * MT + mt) * 128, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(wp + (size_t)(a * MT + mt) * 128 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned* cp = &p.ctr[grp * p.nslab + slab];
      const unsigned old = __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)(p.ksplit - 1);
      if (last) __hip_atomic_store(cp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_sh = last;
    }
    __syncthreads();
    if (!last_sh) return;
    if (owner) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s2 = 0; s2 < p.ksplit; ++s2) {
        const float* rp = p.ws + (((size_t)((slab * p.ksplit + s2) * p.ntu + tu) * (TN * MT)) * 64 + lane) * 4;
        u32x4 pv[TN][MT];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pv[a][mt] = load16_agent(rp + (size_t)(a * MT + mt) * 256);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[a][mt].x += __uint_as_float(pv[a][mt].x); acc[a][mt].y += __uint_as_float(pv[a][mt].y);
            acc[a][mt].z += __uint_as_float(pv[a][mt].z); acc[a][mt].w += __uint_as_float(pv[a][mt].w);
          }
      }
    }
  }
  if (!owner) return;

  // ================= epilogue: lane (c16, g) holds y[row0 + 16 mt + 4 g + r][16 tile + c16]
  AT* out = (AT*)p.out;
  if constexpr (SWIGLU) {
    const int n = tile[0] * 16 + c16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row0 + mt * 16 + g * 4 + r;
        if (m >= p.M) continue;
        const float gt = (float)(AT)acc[0][mt][r], up = (float)(AT)acc[1][mt][r];
        const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
        const float sl = (float)(AT)(gt * sig);
        out[(size_t)m * p.ldo + n] = (AT)(sl * up);
      }
    return;
  }
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    if (a == 1 && 2 * tu + 1 >= p.ntiles) break;             // (odd tile count: the unit's second tile does not exist)
    const int n = tile[a] * 16 + c16;
    // all of the lane's h values of this tile in one round trip (left to the loop, every h load sits behind the previous h store)
    float hres[MT * 4];
    if (p.epi == EPI_RESID) {
      const AT* hp = (const AT*)p.resid;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hres[mt * 4 + r] = (float)hp[(size_t)min(row0 + mt * 16 + g * 4 + r, p.M - 1) * p.ldo + n];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row0 + mt * 16 + g * 4 + r;
        if (m >= p.M) continue;
        float y = (float)(AT)acc[a][mt][r];
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
        else ((AT*)p.resid)[(size_t)m * p.ldo + n] = (AT)(hres[mt * 4 + r] + y);
      }
  }
}

struct Q4Plan { int mt, nslab, TW, KW, NS, ksplit, ntu, ntiles, ngroups, nblk; size_t prep_bytes, ws_bytes; };

int q4_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// test / A-B hook (mi_op_gemm_skinny with ksplit < 0): the next plans of this thread; a field of 0 = the cost model's choice
thread_local int q4_force_mt = 0, q4_force_tw = 0, q4_force_kw = 0, q4_force_ks = 0, q4_force_ns = 0;

constexpr int Q4_PAD_BLOCKS = 64;       // padding of the preparation buffers, in 128-k blocks (clamp-free staging runs past a slice's end)

}  // namespace

void gemm_q4_force(int code) {          // code = mt | TW << 3 | KW << 7 | ksplit << 11 | NS << 15; 0 clears
  q4_force_mt = code & 7;
  q4_force_tw = (code >> 3) & 15;
  q4_force_kw = (code >> 7) & 15;
  q4_force_ks = (code >> 11) & 15;
  q4_force_ns = (code >> 15) & 7;
}

int gemv_cu_count();

// Row tiles per workgroup (MT; rows above 16 MT run as slabs), tile units per workgroup (TW) x K lanes (KW) = compute
// waves, staging waves (NS), K slices over workgroups.  One workgroup of up to 12 waves per CU (three per SIMD, <= 168
// VGPRs).  MEASURED, not modelled (tools/debug/q4_sweep.py, Qwen3-14B shapes at 64 rows, us per launch incl. the preparation
// pass; gemm_skinny.hip's split-K kernel in brackets): gate|up 5 x 2 compute + 2 staging waves at 4 row tiles 57.4 (70.8) --
// 218 workgroups, ONE round on 256 CUs where 4 x 2 gives 272 = two rounds (85); lm_head 2 row tiles 254 (272); the narrow
// linears stay on the split-K kernel (q|k|v 24.5 vs 23.4, o 24.1 vs 21.2, down 53 vs 37.5: 80-112 tile groups leave this
// shape ~10 chunks per workgroup around a 2.5 us head and tail) -- gemm_q4_supported() routes only the wide ones here.
static int q4_min_tiles() { static const int v = q4_env("MI_Q4_MIN_TILES", 1024); return v; }

static Q4Plan q4_plan(const LinearW& W, const GemvCall& c, size_t rows) {
  Q4Plan pl{};
  const bool sw = c.epi == EPI_SWIGLU;
  pl.ntiles = (sw ? c.pair_offset : W.N) / 16;
  pl.ntu = sw ? pl.ntiles : (pl.ntiles + 1) / 2;
  pl.nblk = W.K / 128;
  const int mt_all = (int)((rows + 15) / 16);
  double best = 1e30;
  static const int e_mt = q4_env("MI_Q4_MT", 0), e_tw = q4_env("MI_Q4_TW", 0), e_kw = q4_env("MI_Q4_KW", 0),
                   e_ks = q4_env("MI_Q4_KSPLIT", 0), e_ns = q4_env("MI_Q4_NS", 0);                       // A/B overrides
  const int f_mt = q4_force_mt > 0 ? q4_force_mt : e_mt, f_tw = q4_force_tw > 0 ? q4_force_tw : e_tw,
            f_kw = q4_force_kw > 0 ? q4_force_kw : e_kw, f_ks = q4_force_ks > 0 ? q4_force_ks : e_ks,
            f_ns = q4_force_ns > 0 ? q4_force_ns : e_ns;
  const double cus = gemv_cu_count();
  const bool forced = f_mt > 0 || f_tw > 0 || f_kw > 0 || f_ks > 0 || f_ns > 0;
  if (!forced && pl.nblk >= 2) {                     // the measured choice (see above)
    pl.mt = std::min(mt_all, sw ? 4 : 2);
    pl.nslab = (mt_all + pl.mt - 1) / pl.mt;
    pl.TW = 5; pl.KW = 2; pl.NS = 2; pl.ksplit = 1;
    // a narrow matrix behind an RMSNorm (q|k|v): few tile pairs, so two per workgroup and K over four waves (sweep on
    // 7168 x 5120 at 64 rows: 20 us + the 4.5-us preparation pass that is ALSO the norm, against 23.4 + 4.7)
    if (!sw && pl.ntiles < q4_min_tiles()) { pl.TW = 2; pl.KW = 4; pl.NS = 4; }
    pl.ngroups = (pl.ntu + pl.TW - 1) / pl.TW;
    best = 0.0;
  }
  for (int mt = 1; mt <= (sw ? 4 : 2) && best > 0.0; ++mt) {
    if (f_mt > 0 && mt != f_mt) continue;
    if (f_mt == 0 && mt < std::min(mt_all, 2) && rows > 16) continue;      // 16-row slabs only for <= 16 rows (the unpack is paid per slab)
    if (mt > mt_all) continue;
    const int nslab = (mt_all + mt - 1) / mt;
    for (int TW = 1; TW <= 10; ++TW) {
      if (f_tw > 0 && TW != f_tw) continue;
      for (int KW = 1; KW * TW <= 10; ++KW) {
        if (f_kw > 0 && KW != f_kw) continue;
        const int NP = KW * 4 * mt + 1;
        if (NP > 33) continue;                                                // (chunk <= 33 KiB: 9 pieces per staging thread)
        const int npc = mt >= 3 ? 17 : 9;                                       // pieces a staging thread can carry (q4_kernel: NPC)
        const int NS = f_ns > 0 ? f_ns : (NP > 2 * npc ? 4 : (TW * KW <= 8 && NP > 18 ? 4 : 2));
        if (TW * KW + NS > Q4_NWMAX || NS * npc < NP) continue;
        const int nwaves = TW * KW + NS;
        const double slots = cus * (nwaves <= 6 && mt <= 2 ? 2.0 : 1.0);
        const int ngroups = (pl.ntu + TW - 1) / TW;
        for (int ks = 1; ks <= 8; ++ks) {
          if (f_ks > 0 && ks != f_ks) continue;
          if (pl.nblk / ks < KW) continue;
          const double wgs = (double)ngroups * nslab * ks;
          const double rounds = std::ceil(wgs / slots);
          const double blocks = std::ceil((double)pl.nblk / ks / KW);            // per wave
          // a block of the wave's two tiles: ~30 + 16 MT matrix-core cycles x 2 tiles, stretched by the waves that share the SIMD
          const double per_block = 2.0 * (60.0 + 40.0 * mt) * std::max(1.0, (double)TW * KW / 8.0);
          double t = rounds * (blocks * per_block + 6000.0) / 2.0e3;            // ns at ~2 GHz (+ head / tail of a workgroup)
          if (ks > 1) t += 1500.0 + 2.0 * ks * (double)pl.ntu * 2 * mt * 1024 * nslab / 4.0e3;       // partial tiles written + read (~4 TB/s), ticket
          if (t < best) { best = t; pl.mt = mt; pl.nslab = nslab; pl.TW = TW; pl.KW = KW; pl.NS = NS; pl.ksplit = ks; pl.ngroups = ngroups; }
        }
      }
    }
  }
  const int Mt = pl.mt * pl.nslab;
  // (both buffers padded: the staging loads run past a slice's end, and they are clamp-free)
  pl.prep_bytes = (size_t)4 * (pl.nblk + Q4_PAD_BLOCKS) * Mt * 1024 + (size_t)2 * (pl.nblk + Q4_PAD_BLOCKS) * 16 * Mt * sizeof(float);
  pl.prep_bytes = (pl.prep_bytes + 255) & ~(size_t)255;
  pl.ws_bytes = pl.prep_bytes + (pl.ksplit > 1 ? (size_t)pl.nslab * pl.ksplit * pl.ntu * 2 * pl.mt * 1024 : 0);
  return pl;
}

// int4 (group 64) weights, 16-bit activations, 17..128 rows (MI_Q4_MIN_ROWS moves the lower bound for A/B runs)
bool gemm_q4_supported(const LinearW& W, const GemvCall& c, size_t rows) {
  static const int on = q4_env("MI_Q4", 1), min_rows = q4_env("MI_Q4_MIN_ROWS", 17);
  if (!on || c.force_v1 || W.layout != 1 || c.rnd != RND_NONE) return false;
  const bool q4 = ((W.wk == WK_Q4_BF16 && c.act == MI_BF16) || (W.wk == WK_Q4_F16 && c.act == MI_F16)) && W.group == 64 &&
                  W.K % 128 == 0;
  if (!q4 || (int)rows < min_rows || rows > 128 || c.ldx % 8 != 0) return false;
  const int n = c.epi == EPI_SWIGLU ? c.pair_offset : W.N;
  if (n % 16 != 0) return false;
  if (c.epi == EPI_SWIGLU_GU8) return false;
  // wide matrices only (gate|up, lm_head: >= 1024 tiles), unless a plan is forced (tests, A/B): see q4_plan
  // ... and, behind MI_Q4_NARROW=1 (A/B), the narrow matrix that has an RMSNorm in front (q|k|v), where the preparation pass
  // replaces the norm launch: measured on the config-5 shard 7301 / 7309 tok/s against 7345 / 7342 without (same box,
  // alternating) -- not taken
  static const int narrow = q4_env("MI_Q4_NARROW", 0);
  const bool forced = q4_force_mt > 0 || q4_force_tw > 0 || q4_force_kw > 0 || q4_force_ks > 0 || q4_force_ns > 0;
  if (!forced && W.N / 16 < q4_min_tiles() && !(narrow && c.pro == PRO_NORM && W.N / 16 >= 256 && rows > 32)) return false;
  return true;
}

size_t gemm_q4_ws_bytes(const LinearW& W, const GemvCall& c, size_t rows) { return q4_plan(W, c, rows).ws_bytes; }
int gemm_q4_ksplit(const LinearW& W, const GemvCall& c, size_t rows) { return q4_plan(W, c, rows).ksplit; }
int gemm_q4_groups(const LinearW& W, const GemvCall& c, size_t rows) {
  const Q4Plan pl = q4_plan(W, c, rows);
  return pl.ngroups * pl.nslab;
}

namespace {

template <typename AT, int MT, bool SW>
int q4_launch_k(const Q4Params& p, int grid, int nthreads, size_t lds, hipStream_t st) {
  auto kern = q4_kernel<AT, MT, SW>;
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024));
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), lds, st, p);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename AT>
int q4_launch_at(const Q4Params& p, int mt, bool sw, int grid, int nthreads, size_t lds, hipStream_t st) {
#define GO(MTV) return sw ? q4_launch_k<AT, MTV, true>(p, grid, nthreads, lds, st) : q4_launch_k<AT, MTV, false>(p, grid, nthreads, lds, st)
  // (3 / 4 row tiles: SwiGLU only -- the plain / residual / LoRA epilogue of 32 accumulator registers per tile pair does not
  // fit the 168-register budget without spilling, and q4_plan never asks for it)
  switch (mt) {
    case 1: GO(1);
    case 2: GO(2);
    case 3: if (sw) return q4_launch_k<AT, 3, true>(p, grid, nthreads, lds, st); break;
    case 4: if (sw) return q4_launch_k<AT, 4, true>(p, grid, nthreads, lds, st); break;
  }
#undef GO
  return fail(MI_ERR_INVALID, "gemm_q4: 1..4 row tiles per workgroup");
}

}  // namespace

// c.pro = PRO_NORM is taken here (the preparation pass normalises); `ws` holds gemm_q4_ws_bytes(), `ctr` gemm_q4_groups()
// zeroed words (used only when K is split over workgroups)
int launch_gemm_q4(const LinearW& W, const GemvCall& c, size_t rows, hipStream_t st, void* ws, unsigned* ctr) {
  if (!gemm_q4_supported(W, c, rows)) return fail(MI_ERR_INVALID, "gemm_q4: unsupported call");
  if (ws == nullptr) return fail(MI_ERR_INVALID, "gemm_q4: workspace missing");
  const Q4Plan pl = q4_plan(W, c, rows);
  if (pl.mt == 0) return fail(MI_ERR_INVALID, "gemm_q4: no plan");
  if (pl.ksplit > 1 && ctr == nullptr) return fail(MI_ERR_INVALID, "gemm_q4: counters missing");
  const int Mt = pl.mt * pl.nslab;
  void* xq = ws;
  float* sx = (float*)((char*)ws + (size_t)4 * (pl.nblk + Q4_PAD_BLOCKS) * Mt * 1024);
  const void* nw = c.pro == PRO_NORM ? c.norm_w : nullptr;
  if (c.act == MI_BF16)
    hipLaunchKernelGGL(q4_prep_kernel<bf16>, dim3(16 * Mt), dim3(256), 0, st, (const bf16*)c.x, c.ldx, (int)rows, Mt, W.K, (const bf16*)nw, c.eps, (bf16*)xq, sx);
  else
    hipLaunchKernelGGL(q4_prep_kernel<f16>, dim3(16 * Mt), dim3(256), 0, st, (const f16*)c.x, c.ldx, (int)rows, Mt, W.K, (const f16*)nw, c.eps, (f16*)xq, sx);
  MI_HIP(hipGetLastError());
  Q4Params p{};
  p.xq = xq; p.sx = sx; p.M = (int)rows; p.Mt = Mt;
  p.w = W.w; p.N = W.N; p.K = W.K;
  p.epi = c.epi; p.out = c.out; p.ldo = c.ldo; p.resid = c.resid; p.pair_offset = c.epi == EPI_SWIGLU ? c.pair_offset : 0;
  p.ntu = pl.ntu; p.ntiles = pl.ntiles; p.TW = pl.TW; p.KW = pl.KW; p.NS = pl.NS; p.nblk = pl.nblk; p.ksplit = pl.ksplit;
  p.nslab = pl.nslab; p.nunits = pl.ngroups * pl.ksplit;
  p.ws = (float*)((char*)ws + pl.prep_bytes); p.ctr = ctr;
  static const int dbg = q4_env("MI_Q4_DBG", 0);
  p.dbg = dbg;
  static unsigned long long* trace_buf = nullptr;
  if (dbg & 8) {
    if (!trace_buf) { MI_HIP(hipMalloc(&trace_buf, 8 * 12 * 64 * 8)); }
    MI_HIP(hipMemsetAsync(trace_buf, 0, 8 * 12 * 64 * 8, st));
    p.trace = trace_buf;
  }
  p.lora_t = c.lora_t; p.lora_t_ld = c.lora_t_ld;
  p.lora_b0 = W.lora_b[0]; p.lora_b1 = W.lora_b[1];
  p.lora_row0_0 = W.lora_row0[0]; p.lora_n_0 = W.lora_n[0]; p.lora_rank_0 = W.lora_rank[0]; p.lora_scale_0 = W.lora_scale[0];
  p.lora_row0_1 = W.lora_row0[1]; p.lora_n_1 = W.lora_n[1]; p.lora_rank_1 = W.lora_rank[1]; p.lora_scale_1 = W.lora_scale[1];
  const size_t lds = 2 * (size_t)(pl.KW * 4 * pl.mt + 2) * 1024;
  const int nthreads = (pl.TW * pl.KW + pl.NS) * 64;
  const int grid = pl.nslab > 1 ? (p.nunits + 7) / 8 * 8 * pl.nslab : p.nunits;
  const int rc = c.act == MI_BF16 ? q4_launch_at<bf16>(p, pl.mt, c.epi == EPI_SWIGLU, grid, nthreads, lds, st)
                                  : q4_launch_at<f16>(p, pl.mt, c.epi == EPI_SWIGLU, grid, nthreads, lds, st);
  if ((dbg & 8) && rc == MI_OK) {               // debug: per-wave timeline of the first workgroups, cycles relative to the workgroup's first stamp
    static int dumps = 0;
    std::vector<unsigned long long> h(8 * 12 * 64);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost);
    if (dumps++ == 2) {
      fprintf(stderr, "q4 trace: N=%d K=%d M=%d mt=%d TW=%d KW=%d NS=%d ks=%d grid=%d\n", W.N, W.K, (int)rows, pl.mt, pl.TW, pl.KW, pl.NS, pl.ksplit, grid);
      for (int wg = 0; wg < 2; ++wg) {
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 12; ++w) if (h[(wg * 12 + w) * 64]) t0 = std::min(t0, h[(wg * 12 + w) * 64]);
        for (int w = 0; w < 12; ++w) {
          if (!h[(wg * 12 + w) * 64]) continue;
          fprintf(stderr, " wg%d w%-2d:", wg, w);
          for (int i = 0; i < 64; ++i) { const unsigned long long v = h[(wg * 12 + w) * 64 + i]; if (v) fprintf(stderr, " %d:%llu", i, v - t0); }
          fprintf(stderr, "\n");
        }
      }
    }
  }
  return rc;
}

}  // namespace mi
