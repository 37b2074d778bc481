// chain.hip -- the decode step's linear layers of one transformer block as ONE persistent launch.
//
// STATUS: built, parity-checked (tests/test_gpu_chain.py), MEASURED SLOWER than the launches it would replace, and therefore
// NOT used by the engine -- it is reachable through mi_op_chain (include/mi355_ops.h) only.  DESIGN.md section 8a has the numbers.
//
// What it is.  For a decode step of at most 8 sequences on 16-bit dense weights, the four launches
//     o_proj (+ residual)  ->  RMSNorm + gate|up + SwiGLU  ->  down_proj (+ residual)  ->  RMSNorm + q|k|v of the NEXT block
// (reference call sites: llama.py:143,188 / 189,165 / 165,190 / 186,64-67; qwen3.py the same with its own norms) as one
// kernel whose weight stream does not stop at the seams between them (round-2 verdict, item 4).
//
// Why it was worth trying.  Each of those launches is a weight stream with a dependent chain at either end: kernel boundary,
// activation load, row statistics, staging, first MFMA ... cross-wave reduction, epilogue.  The stamps of round 2 show the
// stream phase of gate|up at ~7.5 TB/s and 5-8 us of head + tail per launch with HBM idle.  Two dependent GEMVs behind one
// in-launch seam (gemv_pair_kernel) did not help because the registers that hold the weights in flight are the registers
// the chain needs.  Here the two are decoupled by ROLES:
//
//   wave 0 of every workgroup is a LOADER: it walks the static schedule of this CU's weight slots (8 KiB = HALF a 16-row
//   tile x 512 k -- half tiles so that 1792 gate|up units fall 7 to a CU instead of 4 / 3 tile pairs) through all four
//   linears and drops them into a 14-slot ring in LDS with global_load_lds_dwordx4 (per-lane source addresses pick the 8
//   rows out of the tile-major blocks), six slots in flight behind the one it issues, waiting for nothing but a free
//   ring slot: weights are read-only, so it runs ahead across every seam.  Alone (consumers freeing slots unread) it streams
//   the 436 MB of a Mistral-7B block in 70-73 us = 6.0-6.25 TB/s.
//   waves 1..7 are CONSUMERS: they wait for the previous linear's outputs (8 sharded arrival counters per op, one poller,
//   write-through stores + agent-scope loads, no fences), then eat ring slots with one v_mfma_f32_16x16x32 per 512-byte
//   block.  k-slot ks belongs to consumer ks mod 7, which reads the A fragments of those 512 k straight from L2 (16-byte
//   sc1 buffer loads; rows >= M come back as zeros from the descriptor's range check), applies the RMSNorm once for all the
//   CU's units (row scales from the 8 x (N / 8) partial sums of h^2 that the residual epilogue in front publishes next to h,
//   added in a fixed order by ONE wave per CU), and keeps one accumulator pair per unit.  At the end of an op all seven leave
//   their partial sums in LDS, meet at an LDS counter, and consumer u mod 7 runs unit u's epilogue (residual / SwiGLU /
//   store through a 128-byte LDS image to 8-byte write-through stores); the last wave to drain signals the CU's arrival.
//
// What the measurements say (MI355X, Mistral-7B block, 8 rows, same process): four single launches 87-90 us back to back,
// the chain 147 us in its first complete form (activations staged in LDS: 7 stagings of 4-7 us each, ring of 5 x 16 KiB),
// 121 us with the activations read per slot from L2, 109 us in this form -- against 70-73 us for its loader alone.  The
// timing-only ablations (ChainParams::debug) put the difference on the consumers' side of every seam: ~5 us of waiting for
// the slowest CU, ~3.5 us of epilogues, ~7 us of activation loads, ~7 us of normalisation, ~7 us of slot reading + MFMA --
// none of it overlapped with the stream, because the ring (112 KiB of LDS = 4.6 us of stream) is the same size as the 128 KiB
// the launches keep in flight in REGISTERS, and an all-to-all seam inside a launch (publish, count, poll, gather: ~8 us) is no
// shorter than a kernel boundary + activation load (~6-8 us).  The guide's own price list says as much ("engine-vs-
// launches" 0.87-0.89x on a layer whose linears are a quarter of these; "cut GEMM -> GEMM seams at these sizes"); the
// experiment confirms it for this model.  What would change the answer is a run-ahead buffer two to three times the ring
// (consumers pre-reading their slots into registers while they wait at a seam) -- not built.
//
// One workgroup per CU (the LDS footprint guarantees it), grid = number of CUs: every hand-off is between resident
// workgroups, every spin is bounded and reports through *error.  Arithmetic: the rounding points are those of
// gemv_mfma.hip (T(acc); T(h + y); SwiGLU in T; RMSNorm as w * T(x * rsqrt(mean + eps))); float32 sums are taken in another
// order, so results agree with the single launches to 16-bit rounding flips: against a float64-exact reference both are
// equally close (tests/test_gpu_chain.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "../../../mlx_parallm_amd/csrc/gemv_phase.h"
#include "chain.h"

namespace mi {

namespace {

using namespace gemv;

constexpr int CH_NT = 512;          // wave 0 = loader, waves 1..7 = consumers
constexpr int CH_NC = 7;
constexpr int CH_SLOT = 8192;       // ring slot: HALF a 16-row tile (8 weight rows) x 512 k = 16 k-blocks x 512 bytes
constexpr int CH_SLOT_K = 512;
constexpr int CH_DMA = 8;           // DMA instructions per slot (one covers two k-blocks: 64 lanes x 16 bytes)
constexpr int CH_LAG = 6;           // slots in flight behind the one being issued (vmcnt counts CH_DMA per slot, 63 at most)
constexpr int CH_NRING = 14;
constexpr int CH_MAX_OPS = 4;
constexpr int CH_MAXU = 9;          // half-tile units of one op per CU (Qwen3-14B gate|up: 2176 / 256 = 8.5)
constexpr int CH_SHARDS = 8;
constexpr int CH_SHARD_STRIDE = 32; // words: one 128-byte line per shard
constexpr int CH_ROWS = 8;

struct ChainOp {
  const char* w;        // tile-major 16-bit weights (repack.hip)
  int N;                // output columns (SwiGLU: pair_offset; the matrix holds 2 N rows, gate then up)
  int K;
  int epi, pro;
  const void* norm_w; float eps;
  const void* x; int ldx;
  void* out; int ldo;
  void* resid;
  int wait_prev;        // x is written by the previous op of THIS launch
  float* sq_out;        // EPI_RESID: [8 rows][sq_ld] sums of h^2 over each half tile's 8 columns (for the RMSNorm behind it)
  const float* sq_in;   // PRO_NORM: the producer's partial sums; sq_ld = 8 * (floats per lane group), padded with zeros
  int sq_ld;
};

struct ChainParams {
  ChainOp op[CH_MAX_OPS];
  int nops, M;
  unsigned* ctr;        // [op][CH_SHARDS][CH_SHARD_STRIDE] monotonic arrival counters (one set per op of the chain)
  unsigned base;        // sum over an op's shards before this launch (every launch adds gridDim.x to each op's set)
  unsigned spin_limit;
  int* error;
  int debug;            // timing-only ablations (MI_CHAIN_DEBUG; results wrong on purpose): 1 = consumers free slots without reading
                        // them, 2 = no waiting for the previous op, 4 = no RMSNorm, 8 = no epilogue, 16 = activations loaded and
                        // normalised but slots freed unread, 32 = slots read and multiplied but no activation loads
};

// ---- LDS layout: [ring: 14 x 8 KiB][partial sums: 9 units x 7 waves x 2 x 256 B][flags]
constexpr unsigned L_RING = 0;
constexpr unsigned L_PART = CH_NRING * CH_SLOT;
constexpr unsigned L_FLAGS = L_PART + CH_MAXU * CH_NC * 2 * 256;
constexpr unsigned L_TOTAL = L_FLAGS + 64 * 4;
static_assert(L_TOTAL <= 160 * 1024, "LDS");
// flag words (ring slots <= 16)
enum : int { F_FULL = 0, F_FREE = 16, F_OPREADY = 32, F_ARRIVE = 33, F_RSREADY = 34, F_ABORT = 35, F_RS = 40 };   // F_RS: 8 floats

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {     // lanes 0..7 carry values, the rest 0
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  return __builtin_amdgcn_readfirstlane(v);
}
// Every LDS access goes through EXPLICIT address-space-3 pointers: a flag reached through a generic pointer compiles to
// flat_load / flat_store, which count on vmcnt -- hipcc then waits vmcnt(0) behind it and drains the loader's DMA queue.
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) volatile unsigned lds_flag;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
__device__ __forceinline__ unsigned lds_ld(lds_flag* p) { return *p; }
__device__ __forceinline__ void lds_st(lds_flag* p, unsigned v) { *p = v; }

// bounded spin on an LDS word; on timeout (or after another wave's timeout) reports and carries on
template <typename Pred>
__device__ __forceinline__ void spin_lds(lds_flag* w, Pred ok, const ChainParams& p, lds_flag* flags, int code) {
  unsigned spins = 0;
  while (!ok(lds_ld(w))) {
    if (lds_ld(flags + F_ABORT) != 0u) return;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > p.spin_limit) {
      lds_st(flags + F_ABORT, 1u);
      __hip_atomic_store(p.error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}

// one KiB: global (per-lane address, non-temporal: these bytes are read once by one CU) -> LDS at a wave-uniform address
__device__ __forceinline__ void dma_block(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// half-tile units of an op: N / 8 (SwiGLU: gate half tile + the up half tile of the same columns)
__device__ __forceinline__ int op_units(const ChainOp& o) { return o.N >> 3; }
__device__ __forceinline__ int op_na(const ChainOp& o) { return o.epi == EPI_SWIGLU ? 2 : 1; }

// ---------------------------------------------------------------------------------------------------------------- loader
// Schedule of a CU (loader and consumers enumerate it alike): for op, for k-slot ks (512 k), for unit u = cu, cu + G, ...,
// for a (gate, up): one ring slot.  k-slot-major, so that the consumer that owns ks (ks mod 7) loads and normalises the
// activations of those 512 k ONCE for all the CU's units.
__device__ __forceinline__ void chain_loader(const ChainParams& p, lds_byte* smem) {
  lds_flag* flags = (lds_flag*)(smem + L_FLAGS);
  const unsigned ring_base = (unsigned)(size_t)smem + L_RING;      // LDS byte address
  const int lane = threadIdx.x & 63;
  const int G = gridDim.x, cu = blockIdx.x;
  // lane -> (k-block of the pair, lane group g, row r of the half tile): source offset inside a 1-KiB tile-major block
  const int lsrc = (lane >> 5) * 1024 + ((lane >> 3) & 3) * 256 + (lane & 7) * 16;
  unsigned q = 0, published = 0;
  for (int oi = 0; oi < p.nops; ++oi) {
    const ChainOp& o = p.op[oi];
    const int nks = o.K / CH_SLOT_K, na = op_na(o), U = op_units(o);
    const size_t tile_bytes = (size_t)(o.K / 32) * 1024;
    for (int ks = 0; ks < nks; ++ks) {
      for (int u = cu; u < U; u += G) {
        for (int a = 0; a < na; ++a) {
          const unsigned ri = q % (unsigned)CH_NRING;
          if (q >= (unsigned)CH_NRING) {
            const unsigned want = q - (unsigned)CH_NRING + 1u;
            spin_lds(flags + F_FREE + ri, [&](unsigned v) { return v == want; }, p, flags, 2);
          }
          const int tile = (u >> 1) + a * (o.N >> 4);
          const char* src = o.w + (size_t)tile * tile_bytes + (size_t)ks * (CH_SLOT_K / 32) * 1024 + (u & 1) * 128 + lsrc;
          const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + ri * CH_SLOT);
#pragma unroll
          for (int j = 0; j < CH_DMA; ++j) dma_block(src + j * 2048, dst + j * 1024);
          ++q;
          if (q > (unsigned)CH_LAG) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(CH_DMA * CH_LAG) : "memory");
            lds_st(flags + F_FULL + (published % (unsigned)CH_NRING), published + 1u);
            ++published;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  while (published < q) { lds_st(flags + F_FULL + (published % (unsigned)CH_NRING), published + 1u); ++published; }
}

// -------------------------------------------------------------------------------------------------------------- consumers
// agent-scope 16-byte load through a buffer descriptor (aux 16 = sc1: served by L2, never by this CU's L1); bytes past the
// descriptor's range read as zero, which is how rows >= M of the A fragment come out as zeros
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 16));
}

// One half-tile unit's epilogue by ONE wave.  y0 / y1: lane (c16 = lane & 15 < 8, g = lane >> 4 < 2) holds rows 4 g + r of
// column c16 -- 16 lanes with data.  Through a 128-byte LDS image [row][8] to 8-byte row segments: lane (row = lane >> 1,
// q = lane & 1), lane < 16, owns columns 4 q .. 4 q + 3 of its row.
template <typename AT>
__device__ __forceinline__ void chain_epilogue(const ChainParams& p, const ChainOp& o, int unit, f32x4 y0, f32x4 y1,
                                               lds_byte* scratch, int lane) {
  const int c16 = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) AT lds_at;
  lds_at* img = (lds_at*)scratch;
  if (g < 2 && c16 < 8) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = (float)(AT)y0[r];
      if (o.epi == EPI_SWIGLU) {
        const float gt = y, up = (float)(AT)y1[r];
        const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
        const float sl = (float)(AT)(gt * sig);
        y = (float)(AT)(sl * up);
      }
      img[(g * 4 + r) * 8 + c16] = (AT)y;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float ss = 0.f;
  if (lane < 16) {
    const int row = lane >> 1, q = lane & 1;
    if (row < p.M) {
      const unsigned long long v = *(const __attribute__((address_space(3))) unsigned long long*)(img + row * 8 + q * 4);
      const size_t off = (size_t)row * o.ldo + (size_t)unit * 8 + q * 4;
      if (o.epi == EPI_RESID) {
        AT* h = (AT*)o.resid;
        // (this CU wrote these columns itself in an earlier op of the launch, or they are older than the launch)
        const unsigned long long hold = __hip_atomic_load((const unsigned long long*)(h + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const AT* ya = (const AT*)&v;
        const AT* ha = (const AT*)&hold;
        AT o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { o4[i] = (AT)((float)ha[i] + (float)ya[i]); ss = fmaf((float)o4[i], (float)o4[i], ss); }
        __hip_atomic_store((unsigned long long*)(h + off), *(const unsigned long long*)o4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store((unsigned long long*)((AT*)o.out + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (o.epi == EPI_RESID && o.sq_out != nullptr) {          // sum of h^2 over this unit's 8 columns, per row
    ss += __shfl_xor(ss, 1);
    if (lane < 16 && (lane & 1) == 0)
      __hip_atomic_store((unsigned*)(o.sq_out + (size_t)(lane >> 1) * o.sq_ld + unit), __float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename AT>
__device__ __forceinline__ void chain_consumer(const ChainParams& p, lds_byte* smem, int cw) {
  lds_flag* flags = (lds_flag*)(smem + L_FLAGS);
  const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
  const int G = gridDim.x, cu = blockIdx.x;
  unsigned q = 0, agen = 0;                          // slot sequence, arrival generation of the consumers
  for (int oi = 0; oi < p.nops; ++oi) {
    const ChainOp& o = p.op[oi];
    const int nks = o.K / CH_SLOT_K, na = op_na(o), U = op_units(o);
    const int nu = U > cu ? (U - cu + G - 1) / G : 0;          // units of this CU
    // ---- hand-off: the previous op's outputs are complete on every CU
    if (o.wait_prev && !(p.debug & 2)) {
      if (cw == 0) {
        const unsigned target = p.base + (unsigned)G;          // every CU has signalled op oi - 1
        const unsigned* set = p.ctr + (size_t)(oi - 1) * CH_SHARDS * CH_SHARD_STRIDE;
        unsigned spins = 0;
        for (;;) {
          unsigned v = 0;
          if (lane < CH_SHARDS) v = __hip_atomic_load(set + lane * CH_SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = wave_sum_u32(v);
          if ((int)(v - target) >= 0 || lds_ld(flags + F_ABORT) != 0u) break;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > p.spin_limit) {
            lds_st(flags + F_ABORT, 1u);
            __hip_atomic_store(p.error, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        lds_st(flags + F_OPREADY, (unsigned)oi + 1u);
      } else {
        const unsigned want = (unsigned)oi + 1u;
        spin_lds(flags + F_OPREADY, [&](unsigned v) { return v >= want; }, p, flags, 5);
      }
      asm volatile("" ::: "memory");
    }
    if (nu == 0) {
      // nothing to compute for this op on this CU: it still counts as arrived
      if (cw == 0 && lane == 0)
        __hip_atomic_fetch_add(p.ctr + ((size_t)oi * CH_SHARDS + (cu & (CH_SHARDS - 1))) * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    // ---- RMSNorm row scale of the lane's A-fragment row (lane & 7): the producer's per-unit sums of h^2, [row][sq_ld], lane
    // group j = lane >> 3 adds its sq_ld / 8 consecutive floats in order (<= 20 loads of 16 bytes, all in flight at
    // once; the pads are zero), then the 8 groups in a fixed tree.  Every wave does this for itself: no LDS, no barrier.
    float rs = 1.0f;
    bool rs_valid = false;
    const bool norm = o.pro == PRO_NORM && !(p.debug & 4);
    if (norm && cw == 0) {                         // ONE wave per CU reads the table (all seven did at first: 36 MB of L2 traffic per op)
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)o.sq_in, 0, CH_ROWS * o.sq_ld * 4, 0x00020000);
      const int per = o.sq_ld >> 3;                // floats per lane group (a multiple of 4)
      const unsigned so = (unsigned)(((lane & 7) * o.sq_ld + (lane >> 3) * per) * 4);
      u32x4 sv[20];
#pragma unroll
      for (int j = 0; j < 20; ++j) sv[j] = (j * 4 < per) ? buf_load16(srs, so + j * 16) : u32x4{0u, 0u, 0u, 0u};
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 20; ++j) {
        t += __uint_as_float(sv[j].x); t += __uint_as_float(sv[j].y); t += __uint_as_float(sv[j].z); t += __uint_as_float(sv[j].w);
      }
      t += __shfl_xor(t, 8);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      rs = 1.0f / sqrtf(t / (float)o.K + o.eps);
      rs_valid = true;
      if (lane < CH_ROWS) lds_st(flags + F_RS + lane, __float_as_uint(rs));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      lds_st(flags + F_RSREADY, (unsigned)oi + 1u);
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)o.x, 0, p.M * o.ldx * (int)sizeof(AT), 0x00020000);
    const unsigned xlane = (unsigned)((lane & 7) * o.ldx + g * 8) * (unsigned)sizeof(AT);
    f32x4 acc[CH_MAXU][2];
#pragma unroll
    for (int u = 0; u < CH_MAXU; ++u) { acc[u][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int ks = 0; ks < nks; ++ks) {
      const bool mine = (ks % CH_NC) == cw;
      if (!mine) { q += (unsigned)(nu * na); continue; }
      // the A fragments of these 512 k, straight from L2, normalised once for all the CU's units
      u32x4 xf[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) xf[j] = u32x4{0u, 0u, 0u, 0u};
      if (!(p.debug & (1 | 32))) {
        const unsigned xo = xlane + (unsigned)(ks * CH_SLOT_K) * (unsigned)sizeof(AT);
#pragma unroll
        for (int j = 0; j < 16; ++j) xf[j] = buf_load16(xrs, xo + j * 64);
        if (norm) {
          if (!rs_valid) {                           // (the loads above are in flight while this wave waits for the row scales)
            const unsigned want = (unsigned)oi + 1u;
            spin_lds(flags + F_RSREADY, [&](unsigned v) { return v >= want; }, p, flags, 9);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            rs = __uint_as_float(lds_ld(flags + F_RS + (lane & 7)));
            rs_valid = true;
          }
          const AT* wn = (const AT*)o.norm_w + ks * CH_SLOT_K + g * 8;
          const f32x2 rs2 = {rs, rs};
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const u32x4 nw = *(const u32x4*)(wn + j * 32);
            uint32_t* d = (uint32_t*)&xf[j];
            const uint32_t* w4 = (const uint32_t*)&nw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t xn = pack2<AT>(unpack2<AT>(d[i]) * rs2);         // cast_T(x32 * rsqrt(..))
              d[i] = pack2<AT>(unpack2<AT>(xn) * unpack2<AT>(w4[i]));         // w * (.)  in T
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH_MAXU; ++u) {
        if (u < nu) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            if (a < na) {
              const unsigned ri = q % (unsigned)CH_NRING;
              const unsigned want = q + 1u;
              ++q;
              spin_lds(flags + F_FULL + ri, [&](unsigned v) { return v == want; }, p, flags, 6);
              asm volatile("" ::: "memory");
              if (p.debug & (1 | 16)) { lds_st(flags + F_FREE + ri, want); continue; }
              // B fragment of k-block kb: lane (c16, g) -> [kb][g][c16 & 7] (columns 8..15 of the MFMA repeat 0..7 and are ignored)
              const lds_byte* slot = smem + L_RING + ri * CH_SLOT + g * 128 + (c16 & 7) * 16;
              u32x4 wf[16];
#pragma unroll
              for (int j = 0; j < 16; ++j) wf[j] = *(const lds_u32x4*)(slot + j * 512);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (orders the reads above before the flag below)
              lds_st(flags + F_FREE + ri, want);
              f32x4 c = acc[u][a];
#pragma unroll
              for (int j = 0; j < 16; ++j) c = mfma16<AT>(xf[j], wf[j], c);
              acc[u][a] = c;
            }
          }
        }
      }
    }
    // ---- end of the op on this CU: every consumer leaves its partial sums, all wait for all, then unit u's epilogue is run
    // by consumer u mod 7 (the 7 partials added in wave order)
    lds_byte* part = smem + L_PART;
    if (g < 2 && c16 < 8) {
#pragma unroll
      for (int u = 0; u < CH_MAXU; ++u) {
        if (u < nu) {
          *(lds_f32x4*)(part + ((u * CH_NC + cw) * 2 + 0) * 256 + (g * 8 + c16) * 16) = acc[u][0];
          *(lds_f32x4*)(part + ((u * CH_NC + cw) * 2 + 1) * 256 + (g * 8 + c16) * 16) = acc[u][1];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add((lds_u32*)(flags + F_ARRIVE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    {
      const unsigned want = ++agen * CH_NC;
      spin_lds(flags + F_ARRIVE, [&](unsigned v) { return (int)(v - want) >= 0; }, p, flags, 7);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int u = cw; u < nu; u += CH_NC) {
      f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
      if (g < 2 && c16 < 8) {
#pragma unroll
        for (int w = 0; w < CH_NC; ++w) {
          y0 += *(const lds_f32x4*)(part + ((u * CH_NC + w) * 2 + 0) * 256 + (g * 8 + c16) * 16);
          y1 += *(const lds_f32x4*)(part + ((u * CH_NC + w) * 2 + 1) * 256 + (g * 8 + c16) * 16);
        }
      }
      // the epilogue's 128-byte image goes where this wave's own partial of unit u was (read above, free now)
      if (!(p.debug & 8)) chain_epilogue<AT>(p, o, cu + u * G, y0, y1, part + ((u * CH_NC + cw) * 2) * 256, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have left
    // second arrival: every consumer is done READING the partial sums (the next op overwrites them) and has drained its
    // stores; the last one to arrive signals this CU's arrival for the op
    unsigned arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add((lds_u32*)(flags + F_ARRIVE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    ++agen;
    if (arrived == agen * CH_NC - 1u) {
      if (lane == 0) __hip_atomic_fetch_add(p.ctr + ((size_t)oi * CH_SHARDS + (cu & (CH_SHARDS - 1))) * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (oi + 1 < p.nops) {          // nobody overwrites the partial sums before everybody has read them
      const unsigned want = agen * CH_NC;
      spin_lds(flags + F_ARRIVE, [&](unsigned v) { return (int)(v - want) >= 0; }, p, flags, 8);
    }
  }
}

template <typename AT>
__global__ __launch_bounds__(CH_NT, 2) void chain_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds_byte* lds = (lds_byte*)smem;
  if (threadIdx.x < 64) ((lds_flag*)(lds + L_FLAGS))[threadIdx.x] = 0u;
  __syncthreads();
  if (wave == 0) chain_loader(p, lds);
  else chain_consumer<AT>(p, lds, wave - 1);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- host
bool chain_linear_ok(const LinearW& W, const GemvCall& c) {
  const bool dense = (W.wk == WK_BF16 && c.act == MI_BF16) || (W.wk == WK_F16 && c.act == MI_F16);
  if (!dense || W.layout != 1 || c.rnd != RND_NONE || c.force_v1) return false;
  if (W.lora_b[0] != nullptr || W.lora_b[1] != nullptr) return false;
  if (W.K % CH_SLOT_K != 0 || c.ldx % 8 != 0 || c.ldo % 4 != 0) return false;
  const int n = c.epi == EPI_SWIGLU ? c.pair_offset : W.N;
  if (n % 16 != 0) return false;
  if (c.epi != EPI_STORE && c.epi != EPI_RESID && c.epi != EPI_SWIGLU) return false;
  if ((n / 8 + chain_grid() - 1) / chain_grid() > CH_MAXU) return false;                   // half-tile units per CU
  if (c.pro == PRO_NORM && (c.sq_in == nullptr || c.sq_parts != chain_sq_ld(W.K))) return false;   // row statistics come from the producer's per-unit sums
  if (c.pro == PRO_NORM && chain_sq_ld(W.K) > 8 * 80) return false;
  return true;
}

int chain_grid() { return std::min(gemv_cu_count(), 256); }
// row stride (floats) of the [8][ld] table of per-unit sums of squares for a hidden size of K: K / 8 units, padded so that 8
// lane groups take equal runs of a multiple of 4 floats (pads stay zero: the table is zeroed once, units never write them)
int chain_sq_ld(int K) { const int per = ((K / 8 + 7) / 8 + 3) / 4 * 4; return 8 * per; }

int launch_chain(const LinearW* const* W, const GemvCall* calls, const int* wait_prev, int nops, int M, int act,
                 unsigned* ctr, unsigned base, unsigned spin_limit, int* error, hipStream_t st) {
  if (nops < 1 || nops > CH_MAX_OPS || M < 1 || M > CH_ROWS) return fail(MI_ERR_INVALID, "chain: 1..4 linears, 1..8 rows");
  ChainParams p{};
  for (int i = 0; i < nops; ++i) {
    if (!chain_linear_ok(*W[i], calls[i])) return fail(MI_ERR_UNSUPPORTED, "chain: linear not supported");
    ChainOp& o = p.op[i];
    o.w = (const char*)W[i]->w; o.K = W[i]->K;
    o.N = calls[i].epi == EPI_SWIGLU ? calls[i].pair_offset : W[i]->N;
    o.epi = calls[i].epi; o.pro = calls[i].pro; o.norm_w = calls[i].norm_w; o.eps = calls[i].eps;
    o.x = calls[i].x; o.ldx = calls[i].ldx; o.out = calls[i].out; o.ldo = calls[i].ldo; o.resid = calls[i].resid;
    o.wait_prev = wait_prev[i];
    o.sq_out = calls[i].epi == EPI_RESID ? calls[i].sq_out : nullptr;
    o.sq_in = calls[i].pro == PRO_NORM ? calls[i].sq_in : nullptr;
    o.sq_ld = o.sq_in ? chain_sq_ld(o.K) : (o.sq_out ? chain_sq_ld(o.N) : 0);
  }
  p.nops = nops; p.M = M;
  p.ctr = ctr; p.base = base; p.spin_limit = spin_limit; p.error = error;
  static const int dbg = getenv("MI_CHAIN_DEBUG") ? atoi(getenv("MI_CHAIN_DEBUG")) : 0;
  p.debug = dbg;
  const int grid = chain_grid();
  auto go = [&](auto kern) -> int {
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CH_NT), L_TOTAL, st, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
  };
  return act == MI_BF16 ? go(chain_kernel<bf16>) : go(chain_kernel<f16>);
}

}  // namespace mi
