// repack.hip -- one-time conversion of a weight matrix from the checkpoint's row-major layout
// (W[N][K], MLX / safetensors order) to the tile-major layout the streaming kernels read.
//
// Why: a wave of the skinny-GEMM kernel consumes a 16-row x 32-k (dense 16-bit) or 16-row x 128-k
// (int4) block per load instruction.  Row-major, that block is 16 separate 64-byte pieces 2*K bytes
// apart; tile-major it is ONE contiguous KiB, and a workgroup's whole K sweep over its rows is one
// sequential stream -- measured +10..14 % on the decode GEMVs (4096 x 4096 .. 28672 x 4096, batch 8).
//
// Layouts (N % 16 == 0):
//   dense 16-bit: block (i = row/16, j = k/32) at ((i * K/32 + j) * 1024) bytes; inside, lane
//                 l = g*16 + r (r = row%16, g = (k%32)/8) owns bytes [16 l, 16 l + 16) = W[16i+r][32j+8g .. +8].
//   int4 (g64):   block (i, j = k/128) at ((i * K/128 + j) * 1152) bytes:
//                 [0,1024)    codes: lane l = g*16 + r owns the 4 packed dwords 4g..4g+3 of row r's 16
//                 [1024,1088) scales: row r -> 2 x 16-bit (quantisation groups 2j, 2j+1)
//                 [1088,1152) biases: same shape.
//   int8 (g64):   block (i, j = k/64) at ((i * K/64 + j) * 1088) bytes:
//                 [0,1024)    codes: lane l = g*16 + r owns the 16 codes W[16i+r][64j + 16g .. +16) (4 packed dwords)
//                 [1024,1056) scales: row r -> one 16-bit value;  [1056,1088) biases.
#include "kernels.h"

namespace mi {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void repack_dense16_kernel(const uint16_t* src, uint16_t* dst, int N, int K) {
  const size_t pieces = (size_t)N * (K / 8);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pieces; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 6;
    const int lane = (int)(idx & 63), r = lane & 15, g = lane >> 4;
    const size_t i = blk / (K / 32), j = blk % (K / 32);
    const u32x4 v = *(const u32x4*)(src + (i * 16 + r) * (size_t)K + j * 32 + g * 8);
    *(u32x4*)(dst + idx * 8) = v;
  }
}

__global__ void repack_q4_kernel(const uint32_t* codes, const uint16_t* scales, const uint16_t* biases, uint8_t* dst,
                                 int N, int K) {
  const int nb = K / 128;
  const size_t blocks = (size_t)(N / 16) * nb;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < blocks * 64; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 6;
    const int lane = (int)(idx & 63), r = lane & 15, g = lane >> 4;
    const size_t i = blk / nb, j = blk % nb;
    const size_t row = i * 16 + r;
    uint8_t* base = dst + blk * 1152;
    *(u32x4*)(base + lane * 16) = *(const u32x4*)(codes + row * (K / 8) + j * 16 + g * 4);
    if (g == 0) {
      *(uint32_t*)(base + 1024 + r * 4) = *(const uint32_t*)(scales + row * (K / 64) + j * 2);
      *(uint32_t*)(base + 1088 + r * 4) = *(const uint32_t*)(biases + row * (K / 64) + j * 2);
    }
  }
}

__global__ void repack_q8_kernel(const uint32_t* codes, const uint16_t* scales, const uint16_t* biases, uint8_t* dst,
                                 int N, int K) {
  const int nb = K / 64;
  const size_t blocks = (size_t)(N / 16) * nb;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < blocks * 64; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 6;
    const int lane = (int)(idx & 63), r = lane & 15, g = lane >> 4;
    const size_t i = blk / nb, j = blk % nb;
    const size_t row = i * 16 + r;
    uint8_t* base = dst + blk * 1088;
    *(u32x4*)(base + lane * 16) = *(const u32x4*)(codes + row * (K / 4) + j * 16 + g * 4);
    if (g == 0) {
      *(uint16_t*)(base + 1024 + r * 2) = scales[row * (K / 64) + j];
      *(uint16_t*)(base + 1056 + r * 2) = biases[row * (K / 64) + j];
    }
  }
}

// int4 tile-major -> dense 16-bit tile-major [hi | lo] with K' = 2K (the prefill GEMM's operand for quantised
// weights): w = scale * q + bias in float32 (as the oracle dequantises), hi = T(w), lo = T(w - hi), so that
// x . hi + x . lo reproduces x . w to 2^-17 relative -- below the accumulation-order noise of any fp32 GEMM.
// One thread per destination 16-byte piece of the hi half (coalesced writes; the codes are read once).
template <typename T>
__global__ void dequant_q4_hilo_kernel(const uint8_t* src, T* dst, int N, int K) {
  const size_t pieces = (size_t)N * (K / 8);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pieces; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 6;
    const int lane = (int)(idx & 63), r = lane & 15, g = lane >> 4;
    const size_t i = blk / (K / 32), j = blk % (K / 32);
    const size_t row = i * 16 + r;
    const int k = (int)j * 32 + g * 8;
    const uint8_t* qb = src + tiled_block_q4(row, k, K);
    const uint32_t codes = *(const uint32_t*)(qb + tiled_q4_code_off(row, k));
    const float s = (float)*(const T*)(qb + tiled_q4_scale_off(row, k));
    const float b = (float)*(const T*)(qb + tiled_q4_scale_off(row, k) + 64);
    T hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = mul_add_unfused(s, (float)((codes >> (4 * e)) & 15u), b);
      hi[e] = (T)w;
      lo[e] = (T)(w - (float)hi[e]);
    }
    const size_t kb2 = (size_t)(2 * K) / 32;                       // 32-k blocks per row tile in the destination
    T* dh = dst + ((i * kb2 + j) * 64 + lane) * 8;
    T* dl = dst + ((i * kb2 + (size_t)(K / 32) + j) * 64 + lane) * 8;
    *(u32x4*)dh = *(const u32x4*)hi;
    *(u32x4*)dl = *(const u32x4*)lo;
  }
}

// the same for tile-major int8
template <typename T>
__global__ void dequant_q8_hilo_kernel(const uint8_t* src, T* dst, int N, int K) {
  const size_t pieces = (size_t)N * (K / 8);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pieces; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 6;
    const int lane = (int)(idx & 63), r = lane & 15, g = lane >> 4;
    const size_t i = blk / (K / 32), j = blk % (K / 32);
    const size_t row = i * 16 + r;
    const int k = (int)j * 32 + g * 8;
    const uint8_t* qb = src + tiled_block_q8(row, k, K);
    const uint8_t* cp = qb + tiled_q8_code_off(row, k);
    const uint32_t c0 = *(const uint32_t*)cp, c1 = *(const uint32_t*)(cp + 4);
    const float s = (float)*(const T*)(qb + tiled_q8_scale_off(row));
    const float b = (float)*(const T*)(qb + tiled_q8_scale_off(row) + 32);
    T hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t q = ((e < 4 ? c0 : c1) >> (8 * (e & 3))) & 255u;
      const float w = mul_add_unfused(s, (float)q, b);
      hi[e] = (T)w;
      lo[e] = (T)(w - (float)hi[e]);
    }
    const size_t kb2 = (size_t)(2 * K) / 32;
    T* dh = dst + ((i * kb2 + j) * 64 + lane) * 8;
    T* dl = dst + ((i * kb2 + (size_t)(K / 32) + j) * 64 + lane) * 8;
    *(u32x4*)dh = *(const u32x4*)hi;
    *(u32x4*)dl = *(const u32x4*)lo;
  }
}

}  // namespace

size_t dequant_hilo_bytes(int N, int K) { return (size_t)N * K * 2 * 2; }

int launch_dequant_q4_hilo(const LinearW& src, void* dst, hipStream_t st) {
  const bool q4 = src.wk == WK_Q4_BF16 || src.wk == WK_Q4_F16, q8 = src.wk == WK_Q8_BF16 || src.wk == WK_Q8_F16;
  if (src.layout != 1 || (!q4 && !q8) || src.group != 64 || src.K % (q4 ? 128 : 64) != 0 || src.N % 16 != 0)
    return fail(MI_ERR_UNSUPPORTED, "dequant: tile-major int4 / int8 (group 64) matrices only");
  const dim3 grid(4096), block(256);
  if (src.wk == WK_Q4_BF16) hipLaunchKernelGGL(dequant_q4_hilo_kernel<bf16>, grid, block, 0, st, (const uint8_t*)src.w, (bf16*)dst, src.N, src.K);
  else if (src.wk == WK_Q4_F16) hipLaunchKernelGGL(dequant_q4_hilo_kernel<f16>, grid, block, 0, st, (const uint8_t*)src.w, (f16*)dst, src.N, src.K);
  else if (src.wk == WK_Q8_BF16) hipLaunchKernelGGL(dequant_q8_hilo_kernel<bf16>, grid, block, 0, st, (const uint8_t*)src.w, (bf16*)dst, src.N, src.K);
  else hipLaunchKernelGGL(dequant_q8_hilo_kernel<f16>, grid, block, 0, st, (const uint8_t*)src.w, (f16*)dst, src.N, src.K);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

bool tiled_supported(int wk, int N, int K, int group) {
  if (N % 16 != 0) return false;
  if (wk == WK_BF16 || wk == WK_F16) return K % 32 == 0;
  if (wk == WK_Q4_BF16 || wk == WK_Q4_F16) return group == 64 && K % 128 == 0;
  if (wk == WK_Q8_BF16 || wk == WK_Q8_F16) return group == 64 && K % 64 == 0;
  return false;
}

size_t tiled_bytes(int wk, int N, int K) {
  if (wk == WK_BF16 || wk == WK_F16) return (size_t)N * K * 2;
  if (wk == WK_Q8_BF16 || wk == WK_Q8_F16) return (size_t)(N / 16) * (K / 64) * 1088;
  // (+ 64 KiB: gemm_q4.hip's weight stream is clamp-free -- its prefetch runs up to ~48 blocks past a slice's end, i.e. past
  // the last tile's end for the last workgroups; what it loads there is never multiplied)
  return (size_t)(N / 16) * (K / 128) * 1152 + ((size_t)64 << 10);
}

int launch_repack_tiled(const LinearW& src, void* dst, hipStream_t st) {
  if (!tiled_supported(src.wk, src.N, src.K, src.group)) return fail(MI_ERR_UNSUPPORTED, "repack: matrix not eligible for the tile-major layout");
  if (src.layout != 0) return fail(MI_ERR_INVALID, "repack: source is not row-major");
  const dim3 grid(2048), block(256);
  if (src.wk == WK_BF16 || src.wk == WK_F16)
    hipLaunchKernelGGL(repack_dense16_kernel, grid, block, 0, st, (const uint16_t*)src.w, (uint16_t*)dst, src.N, src.K);
  else if (src.wk == WK_Q8_BF16 || src.wk == WK_Q8_F16)
    hipLaunchKernelGGL(repack_q8_kernel, grid, block, 0, st, (const uint32_t*)src.w, (const uint16_t*)src.scales,
                       (const uint16_t*)src.biases, (uint8_t*)dst, src.N, src.K);
  else
    hipLaunchKernelGGL(repack_q4_kernel, grid, block, 0, st, (const uint32_t*)src.w, (const uint16_t*)src.scales,
                       (const uint16_t*)src.biases, (uint8_t*)dst, src.N, src.K);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// Tile-major f16 matrix (N x K) -> tile-major bf16 matrix (N x 2K) whose columns [0, K) hold hi = bf16(w) and [K, 2K) hold
// lo = bf16(w - hi).  An f16 value has 11 significant bits and the bf16 range covers f16's, so hi + lo == w EXACTLY: the
// float32-activation kernels (x split exactly into three bf16 terms) then compute x . w as a float32 dot product of exact
// products for an f16 model too (PagedKVCache mode of the reference: base.py:111-112 promotes everything behind layer 0 to
// float32 whatever the model dtype).  The conversion keeps a block's lane order: block (tile t, k block j) of w becomes
// blocks (t, j) and (t, K / 32 + j) of the result.
__global__ void f16_to_hilo_kernel(const uint16_t* src, uint16_t* dst, int N, int K) {
  const size_t pieces = (size_t)N * (K / 8);
  const size_t kb = (size_t)(K / 32);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pieces; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 6, lane = idx & 63;
    const size_t t = blk / kb, j = blk % kb;
    const u32x4 v = *(const u32x4*)(src + idx * 8);
    const _Float16* h = (const _Float16*)&v;
    __bf16 hi[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float w = (float)h[i];
      hi[i] = (__bf16)w;
      lo[i] = (__bf16)(w - (float)hi[i]);
    }
    *(u32x4*)(dst + ((t * 2 * kb + j) * 64 + lane) * 8) = *(const u32x4*)hi;
    *(u32x4*)(dst + ((t * 2 * kb + kb + j) * 64 + lane) * 8) = *(const u32x4*)lo;
  }
}

// Row-interleaved copy of a tile-major gate|up matrix for the decode GEMV (gemv_mfma, EPI_SWIGLU_GU8).  With gate and up
// as separate 16-row tiles a SwiGLU work item is a PAIR of tiles, and 14336 / 16 = 896 pairs deal 3.5 to each of 256 CUs:
// half the CUs stream a fourth pair while the others idle (12.5 % of the launch).  Tile t of the copy holds gate rows
// 8t .. 8t+7 in its rows 0..7 and up rows 8t .. 8t+7 in rows 8..15: 1792 one-tile items, 7 per CU, same bytes per item as
// any plain tile, and silu(g) * u pairs columns j and j + 8 of one MFMA tile.  16-byte pieces: lane (g, c16) of block
// (tile, k block) holds 8 k of row c16.
__global__ __launch_bounds__(256) void gate_up_interleave_kernel(const u32x4* src, u32x4* dst, int I, int kblocks, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int l = (int)(i & 63), c16 = l & 15, g = l >> 4;
  const size_t blk = i >> 6;
  const int kb = (int)(blk % kblocks);
  const int t = (int)(blk / kblocks);
  const int row = c16 < 8 ? 8 * t + c16 : I + 8 * t + (c16 - 8);
  dst[i] = src[((size_t)(row >> 4) * kblocks + kb) * 64 + g * 16 + (row & 15)];
}

int launch_gate_up_interleave(const LinearW& W, int pair_offset, void* dst, hipStream_t st) {
  if (W.layout != 1 || (W.wk != WK_BF16 && W.wk != WK_F16) || W.N != 2 * pair_offset || pair_offset % 16 != 0 || W.K % 32 != 0)
    return fail(MI_ERR_INVALID, "gate_up_interleave: a tile-major dense 16-bit gate|up matrix with pair_offset % 16 == 0");
  const size_t total = (size_t)W.N * W.K * 2 / 16;
  hipLaunchKernelGGL(gate_up_interleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u32x4*)W.w, (u32x4*)dst,
                     pair_offset, W.K / 32, total);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_f16_to_hilo(const LinearW& tiled_f16, void* dst, hipStream_t st) {
  if (tiled_f16.wk != WK_F16 || tiled_f16.layout != 1) return fail(MI_ERR_INVALID, "f16_to_hilo: a tile-major f16 matrix is expected");
  hipLaunchKernelGGL(f16_to_hilo_kernel, dim3(2048), dim3(256), 0, st, (const uint16_t*)tiled_f16.w, (uint16_t*)dst, tiled_f16.N, tiled_f16.K);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
