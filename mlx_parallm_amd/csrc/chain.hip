// chain.hip -- the decode step's linear layers of one transformer block as ONE persistent launch.
//
// Replaces, for a decode step of at most 8 sequences on 16-bit dense weights, the four launches
//     o_proj (+ residual)  ->  RMSNorm + gate|up + SwiGLU  ->  down_proj (+ residual)  ->  RMSNorm + q|k|v of the NEXT block
// (reference call sites: llama.py:143,188 / 189,165 / 165,190 / 186,64-67; qwen3.py the same with its own norms) by one
// kernel whose weight stream never stops at the seams between them.
//
// Why.  Each of those launches is a weight stream with a dependent chain at either end: kernel boundary, activation
// load, row statistics, staging, first MFMA ... cross-wave reduction, epilogue.  The stamps of round 2
// show the stream phase of gate|up at ~7.5 TB/s and 5-8 us of head + tail per launch with HBM idle: 105 us per
// Mistral-7B block where the bytes take 59.  Two dependent GEMVs behind one in-launch seam (gemv_pair_kernel) did not
// help because the registers that hold the weights in flight are the same registers the chain needs.  Here the two are
// decoupled: ROLES.
//
//   wave 0 of every workgroup is a LOADER: it walks the static schedule of this CU's weight slots (16 KiB = one 16-row
//   tile x 512 k, i.e. 16 tile-major KiB blocks) through all four linears and drops them into a ring in LDS with
//   global_load_lds_dwordx4 (the tile-major block IS lane-linear), two slots in flight behind the one it issues, never
//   waiting for anything but a free ring slot -- weights are read-only, so it runs ahead across every seam.
//   waves 1..7 are CONSUMERS: they wait for the previous linear's outputs (sharded arrival counters, one poller) and then
//   eat ring slots -- slot q belongs to consumer q mod 7 -- with one v_mfma_f32_16x16x32 per KiB block.  The
//   activations are NOT staged in LDS (the first version did: 4-7 us of gather + barriers per seam and per 4096-wide
//   chunk of down_proj's K, with the ring -- all the LDS that was left -- hiding 3 of them): every consumer reads the A
//   fragments of its own slot straight from L2 (write-through stores + agent-scope 16-byte buffer loads, issued before
//   it waits for the slot), applies the RMSNorm on the fly (row statistics from the 8 x 16-column partial sums that the
//   residual epilogue in front publishes next to h), and the whole LDS is ring: 9 slots = 5.8 us of stream.  The wave that completes a tile's seventh partial sum adds the seven in wave order and runs the epilogue
//   (residual / SwiGLU / store), publishing with write-through stores; the wave that completes the CU's last tile of
//   a linear signals the arrival counter.
//
// One workgroup per CU (the LDS footprint guarantees it), grid = number of CUs: every hand-off is between resident
// workgroups, every spin is bounded and reports through *error (the engine fails the step: engine.hip seam_record).
// Arithmetic: the rounding points are those of gemv_mfma.hip (T(acc); T(h + y); SwiGLU in T; RMSNorm as
// w * T(x * rsqrt(mean + eps))); float32 sums are taken in another order (k blocks dealt to 7 waves, row statistics
// from 64-piece segments), so results agree with the single launches to float32 summation noise, not bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "gemv_phase.h"

namespace mi {

namespace {

using namespace gemv;

constexpr int CH_NT = 512;          // wave 0 = loader, waves 1..7 = consumers
constexpr int CH_NC = 7;
constexpr int CH_SLOT = 16384;      // ring slot: 16 tile-major blocks of 1 KiB = 16 weight rows x 512 k
constexpr int CH_SLOT_K = 512;
constexpr int CH_LAG = 3;           // slots in flight behind the one being issued (vmcnt counts 16 per slot, 63 at most)
constexpr int CH_MAX_OPS = 4;
constexpr int CH_SHARDS = 8;
constexpr int CH_SHARD_STRIDE = 32; // words: one 128-byte line per shard
constexpr int CH_ROWS = 8;
constexpr int CH_XPAD = 32;         // bytes added to an x row in LDS: rows land 8 banks apart, the 4 k-groups 4 banks apart

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
  float* sq_out;        // EPI_RESID: [tiles][8] sums of h^2 over the tile's 16 columns per row (for the RMSNorm behind it)
  const float* sq_in; int sq_tiles;   // PRO_NORM: the producer's partial sums and their count (K / 16)
};

struct ChainParams {
  ChainOp op[CH_MAX_OPS];
  int nops, M;
  int nring;            // ring slots
  unsigned* ctr;        // [op][CH_SHARDS][CH_SHARD_STRIDE] monotonic arrival counters (one set per op of the chain)
  unsigned base;        // sum over an op's shards before this launch (every launch adds gridDim.x to each op's set)
  unsigned spin_limit;
  int* error;
  int debug;            // timing-only ablations (MI_CHAIN_DEBUG; results wrong on purpose): 1 = consumers free slots without reading
                        // them, 2 = no waiting for the previous op, 4 = no activation staging, 8 = no epilogue
};

// ---- LDS layout
struct ChainLds {
  unsigned ring, part, flags, total;
};
__host__ __device__ inline ChainLds chain_lds(int nring) {
  ChainLds l;
  l.ring = 0;
  l.part = nring * CH_SLOT;
  l.flags = l.part + 2 * CH_NC * 2 * 512;        // [buf][wave][a][32 lanes x 16 B]
  l.total = l.flags + 64 * 4;
  return l;
}
// flag words
enum : int { F_FULL = 0, F_FREE = 16, F_OPREADY = 32, F_CBAR = 33, F_TILECNT = 34, F_EPIDONE = 36, F_TILESDONE = 38, F_ABORT = 39 };   // (ring slots <= 16)

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
typedef __attribute__((address_space(3))) float lds_f32;
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

// one KiB block: global (per-lane address, non-temporal: these bytes are read once by one CU) -> LDS at a wave-uniform address
__device__ __forceinline__ void dma_block(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ int op_tiles(const ChainOp& o) { return o.N >> 4; }
__device__ __forceinline__ int op_na(const ChainOp& o) { return o.epi == EPI_SWIGLU ? 2 : 1; }

// ---------------------------------------------------------------------------------------------------------------- loader
__device__ __forceinline__ void chain_loader(const ChainParams& p, lds_byte* smem, const ChainLds L) {
  lds_flag* flags = (lds_flag*)(smem + L.flags);
  const unsigned ring_base = (unsigned)(size_t)smem + L.ring;      // LDS byte address
  const int lane = threadIdx.x & 63;
  const int G = gridDim.x, cu = blockIdx.x;
  unsigned q = 0;                                  // slots issued
  unsigned published = 0;                          // slots whose FULL word is set
  for (int oi = 0; oi < p.nops; ++oi) {
    const ChainOp& o = p.op[oi];
    const int nks = o.K / CH_SLOT_K, na = op_na(o);
    const size_t tile_bytes = (size_t)(o.K / 32) * 1024;
    for (int t = cu; t < op_tiles(o); t += G) {
      for (int ks = 0; ks < nks; ++ks) {
        for (int a = 0; a < na; ++a) {
          const unsigned ri = q % (unsigned)p.nring;
          if (q >= (unsigned)p.nring) {
            const unsigned want = q - (unsigned)p.nring + 1u;
            spin_lds(flags + F_FREE + ri, [&](unsigned v) { return v == want; }, p, flags, 2);
          }
          const char* src = o.w + (size_t)(t + a * op_tiles(o)) * tile_bytes + (size_t)ks * CH_SLOT + lane * 16;
          const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + ri * CH_SLOT);
#pragma unroll
          for (int j = 0; j < 16; ++j) dma_block(src + j * 1024, dst + j * 1024);
          ++q;
          if (q > (unsigned)CH_LAG) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(16 * CH_LAG) : "memory");
            lds_st(flags + F_FULL + (published % (unsigned)p.nring), published + 1u);
            ++published;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  while (published < q) { lds_st(flags + F_FULL + (published % (unsigned)p.nring), published + 1u); ++published; }
}

// -------------------------------------------------------------------------------------------------------------- consumers
// agent-scope 16-byte load through a buffer descriptor (aux 16 = sc1: served by L2, never by this CU's L1); bytes past the
// descriptor's range read as zero, which is how rows >= M of the A fragment come out as zeros
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 16));
}

template <typename AT>
__device__ __forceinline__ void chain_epilogue(const ChainParams& p, const ChainOp& o, int tile, f32x4 y0, f32x4 y1,
                                               unsigned long long hold, lds_byte* scratch, int lane) {
  // lanes 0..31 hold rows 4 g + r (g = lane >> 4 < 2) of column c16 = lane & 15; through a 256-byte LDS image [row][16]
  // to 8-byte row segments: lane (row = lane >> 2, q = lane & 3) owns columns 4 q .. 4 q + 3 of its row
  const int c16 = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) AT lds_at;
  lds_at* img = (lds_at*)scratch;
  if (lane < 32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = (float)(AT)y0[r];
      if (o.epi == EPI_SWIGLU) {
        const float gt = y, up = (float)(AT)y1[r];
        const float sig = (float)(AT)(1.0f / (1.0f + expf(-gt)));
        const float sl = (float)(AT)(gt * sig);
        y = (float)(AT)(sl * up);
      }
      img[(g * 4 + r) * 16 + c16] = (AT)y;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float ss = 0.f;
  if (lane < 32) {
    const int row = lane >> 2, q = lane & 3;
    if (row < p.M) {
      unsigned long long v = *(const __attribute__((address_space(3))) unsigned long long*)(img + row * 16 + q * 4);
      const size_t off = (size_t)row * o.ldo + (size_t)tile * 16 + q * 4;
      if (o.epi == EPI_RESID) {
        AT* h = (AT*)o.resid;
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
  if (o.epi == EPI_RESID && o.sq_out != nullptr) {          // sum of h^2 over this tile's 16 columns, per row: q = 0..3 in a fixed tree
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    if (lane < 32 && (lane & 3) == 0)
      __hip_atomic_store((unsigned*)(o.sq_out + (size_t)tile * CH_ROWS + (lane >> 2)), __float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have left
}

template <typename AT>
__device__ __forceinline__ void chain_consumer(const ChainParams& p, lds_byte* smem, const ChainLds L, int cw) {
  lds_flag* flags = (lds_flag*)(smem + L.flags);
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int G = gridDim.x, cu = blockIdx.x;
  unsigned q = 0, ts = 0;                          // slot sequence, tile sequence of this CU
  for (int oi = 0; oi < p.nops; ++oi) {
    const ChainOp& o = p.op[oi];
    const int nks = o.K / CH_SLOT_K, na = op_na(o);
    const int ntile_cu = op_tiles(o) > cu ? (op_tiles(o) - cu + G - 1) / G : 0;
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
    if (ntile_cu == 0) {
      // nothing to compute for this op on this CU: it still counts as arrived
      if (cw == 0 && lane == 0)
        __hip_atomic_fetch_add(p.ctr + ((size_t)oi * CH_SHARDS + (cu & (CH_SHARDS - 1))) * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    // ---- RMSNorm row scale of the lane's A-fragment row (lane & 7): the producer's per-tile sums of h^2, added in a fixed
    // order (lane group j = lane >> 3 takes tiles j, j + 8, ...; then the 8 groups in a fixed tree).  Every wave does this
    // for itself: no LDS round, no barrier.
    float rs = 1.0f;
    const bool norm = o.pro == PRO_NORM && !(p.debug & 4);
    if (norm) {
      const int row = lane & 7, grp = lane >> 3;
      float t = 0.f;
      for (int tt = grp; tt < o.sq_tiles; tt += 8)
        t += __uint_as_float(__hip_atomic_load((const unsigned*)(o.sq_in + (size_t)tt * CH_ROWS + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      t += __shfl_xor(t, 8);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      rs = 1.0f / sqrtf(t / (float)o.K + o.eps);
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)o.x, 0, p.M * o.ldx * (int)sizeof(AT), 0x00020000);
    const unsigned xlane = (unsigned)((lane & 7) * o.ldx + g * 8) * (unsigned)sizeof(AT);
    for (int t = cu; t < op_tiles(o); t += G) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // (named: a run-time index would put them in scratch)
      // residual epilogue: this tile's h (written by this CU in an earlier op of the launch, or before the launch) is fetched now
      unsigned long long hold = 0ull;
      if (o.epi == EPI_RESID && lane < 32 && (lane >> 2) < p.M)
        hold = __hip_atomic_load((const unsigned long long*)((const AT*)o.resid + (size_t)(lane >> 2) * o.ldo + (size_t)t * 16 + (lane & 3) * 4),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int ks = 0; ks < nks; ++ks) {
        for (int a = 0; a < na; ++a, ++q) {
          if ((int)(q % CH_NC) != cw) continue;
          const unsigned ri = q % (unsigned)p.nring;
          const unsigned want = q + 1u;
          // the A fragments of this slot's 512 k, straight from L2 -- issued BEFORE waiting for the slot
          u32x4 xf[16];
          if (!(p.debug & 1)) {
            const unsigned xo = xlane + (unsigned)(ks * CH_SLOT_K) * (unsigned)sizeof(AT);
#pragma unroll
            for (int j = 0; j < 16; ++j) xf[j] = buf_load16(xrs, xo + j * 64);
          }
          spin_lds(flags + F_FULL + ri, [&](unsigned v) { return v == want; }, p, flags, 6);
          asm volatile("" ::: "memory");
          if (p.debug & 1) { lds_st(flags + F_FREE + ri, want); continue; }
          const lds_byte* slot = smem + L.ring + ri * CH_SLOT;
          f32x4 c = a ? acc1 : acc0;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {           // two halves of 8 blocks: 16 weight + 16 norm fragments at once would not fit the registers
            u32x4 wf[8], nw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[j] = *(const lds_u32x4*)(slot + (hf * 8 + j) * 1024 + lane * 16);
            if (hf == 1) {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (orders the reads above before the flag below)
              lds_st(flags + F_FREE + ri, want);
            }
            if (norm) {
              const AT* wn = (const AT*)o.norm_w + ks * CH_SLOT_K + hf * 256 + g * 8;
#pragma unroll
              for (int j = 0; j < 8; ++j) nw[j] = *(const u32x4*)(wn + j * 32);
              const f32x2 rs2 = {rs, rs};
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                uint32_t* d = (uint32_t*)&xf[hf * 8 + j];
                const uint32_t* w4 = (const uint32_t*)&nw[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const uint32_t xn = pack2<AT>(unpack2<AT>(d[i]) * rs2);         // cast_T(x32 * rsqrt(..))
                  d[i] = pack2<AT>(unpack2<AT>(xn) * unpack2<AT>(w4[i]));         // w * (.)  in T
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) c = mfma16<AT>(xf[hf * 8 + j], wf[j], c);
          }
          if (a) acc1 = c; else acc0 = c;
        }
      }
      // ---- tile end: partial sums of the 7 consumers -> the last to arrive adds them in wave order and runs the epilogue
      const unsigned buf = ts & 1u;
      if (ts >= 2u) {
        const unsigned want = ts - 1u;
        spin_lds(flags + F_EPIDONE + buf, [&](unsigned v) { return (int)(v - want) >= 0; }, p, flags, 7);
      }
      lds_byte* part = smem + L.part + buf * (CH_NC * 2 * 512);
      if (lane < 32) {
        *(lds_f32x4*)(part + (cw * 2 + 0) * 512 + lane * 16) = acc0;
        *(lds_f32x4*)(part + (cw * 2 + 1) * 512 + lane * 16) = acc1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      unsigned arrived = 0;
      if (lane == 0) arrived = __hip_atomic_fetch_add((lds_u32*)(flags + F_TILECNT + buf), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      arrived = __builtin_amdgcn_readfirstlane(arrived);
      if (arrived == (unsigned)(CH_NC - 1)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
        if (lane < 32) {
#pragma unroll
          for (int w = 0; w < CH_NC; ++w) {
            const f32x4 a0 = *(const lds_f32x4*)(part + (w * 2 + 0) * 512 + lane * 16);
            const f32x4 a1 = *(const lds_f32x4*)(part + (w * 2 + 1) * 512 + lane * 16);
            y0 += a0; y1 += a1;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_st(flags + F_TILECNT + buf, 0u);
        // the image of the epilogue goes where this wave's own partial was (read above, free now)
        if (!(p.debug & 8)) chain_epilogue<AT>(p, o, t, y0, y1, hold, part + (cw * 2) * 512, lane);
        lds_st(flags + F_EPIDONE + buf, ts + 1u);
        unsigned done = 0;
        if (lane == 0) done = __hip_atomic_fetch_add((lds_u32*)(flags + F_TILESDONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        done = __builtin_amdgcn_readfirstlane(done);
        if (done == (unsigned)(ntile_cu - 1)) {      // every tile of this op on this CU is published (each by a wave that drained)
          lds_st(flags + F_TILESDONE, 0u);
          if (lane == 0) __hip_atomic_fetch_add(p.ctr + ((size_t)oi * CH_SHARDS + (cu & (CH_SHARDS - 1))) * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      ++ts;
    }
  }
}

template <typename AT>
__global__ __launch_bounds__(CH_NT, 2) void chain_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ChainLds L = chain_lds(p.nring);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds_byte* lds = (lds_byte*)smem;
  if (threadIdx.x < 64) ((lds_flag*)(lds + L.flags))[threadIdx.x] = 0u;
  __syncthreads();
  if (wave == 0) chain_loader(p, lds, L);
  else chain_consumer<AT>(p, lds, L, wave - 1);
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
  if (c.pro == PRO_NORM && (c.sq_in == nullptr || c.sq_parts != W.K / 16)) return false;   // row statistics come from the producer's per-tile sums
  return true;
}

int chain_grid() { return std::min(gemv_cu_count(), 256); }

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
    o.sq_in = calls[i].pro == PRO_NORM ? calls[i].sq_in : nullptr; o.sq_tiles = calls[i].sq_parts;
  }
  p.nops = nops; p.M = M;
  p.nring = 9;
  p.ctr = ctr; p.base = base; p.spin_limit = spin_limit; p.error = error;
  static const int dbg = getenv("MI_CHAIN_DEBUG") ? atoi(getenv("MI_CHAIN_DEBUG")) : 0;
  p.debug = dbg;
  const ChainLds L = chain_lds(p.nring);
  const int grid = chain_grid();
  auto go = [&](auto kern) -> int {
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CH_NT), L.total, st, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
  };
  return act == MI_BF16 ? go(chain_kernel<bf16>) : go(chain_kernel<f16>);
}

}  // namespace mi
