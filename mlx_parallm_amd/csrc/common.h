// common.h -- shared host/device helpers for libmi355_decode (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/mi355_decode.h"

namespace mi {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define MI_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::mi::fail(MI_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e));  \
  } while (0)

#define MI_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != MI_OK) return _rc; \
  } while (0)

// ---- element types ------------------------------------------------------------------------
using bf16 = __bf16;
using f16 = _Float16;

inline size_t dtype_size(int dt) {
  switch (dt) {
    case MI_F32: return 4;
    case MI_BF16: return 2;
    case MI_F16: return 2;
    case MI_U32: return 4;
  }
  return 0;
}

// Rounding of an fp32 value to the logical activation dtype, selected at run time.  Used
// where the storage is wider than the logical dtype (float32 KV "PagedKVCache" mode keeps
// every activation buffer in float32 but layer 0 still rounds like the 16-bit model would).
enum : int { RND_NONE = 0, RND_BF16 = 1, RND_F16 = 2 };

__device__ __forceinline__ float round_rt(float v, int rnd) {
  if (rnd == RND_BF16) return (float)(bf16)v;
  if (rnd == RND_F16) return (float)(f16)v;
  return v;
}

template <typename T>
__device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v) { return (T)v; }

// value -> storage type T with the run-time logical rounding applied first
// (a 16-bit storage type IS the logical dtype: the run-time mode only matters for float storage)
template <typename T>
__device__ __forceinline__ T store_act(float v, int rnd) {
  if constexpr (sizeof(T) == 4) return (T)round_rt(v, rnd);
  else return (T)v;
}

// scale * q + bias with two roundings (the oracle's `q * scale + bias`); hipcc would otherwise
// contract it into one FMA (-ffp-contract=fast is the HIP default, and __fmul_rn is a plain `*`).
__device__ __forceinline__ float mul_add_unfused(float a, float b, float c) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p + c;
}

// 16-byte vector of raw bits
struct alignas(16) u128 { uint32_t x, y, z, w; };

__device__ __forceinline__ float bf16lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// Cross-lane sums without LDS traffic: __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round
// trip of ~100+ cycles per step, and the steps of a butterfly are dependent); DPP row rotations are
// plain VALU modifiers.  row16_sum: all-reduce over the 16 lanes of a DPP row.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x128, 0xf, 0xf, false);   // row_ror:8
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x124, 0xf, 0xf, false);   // row_ror:4
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x122, 0xf, 0xf, false);   // row_ror:2
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x121, 0xf, 0xf, false);   // row_ror:1
  return v;
}
// all-reduce over aligned groups of 8 lanes: quad xor-1, quad xor-2, then mirror inside the half row
__device__ __forceinline__ float lane8_sum(float v) {
  v += __builtin_amdgcn_update_dpp(0.f, v, 0xB1, 0xf, 0xf, false);    // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x4E, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
  return v;
}
// z = sum_j t[j] * B[j][col] in rank order, the loads of 8 ranks issued together (a plain loop over a run-time rank
// waits for one L2 round trip per rank: 16 in a row per output tile)
__device__ __forceinline__ float lora_dot(const float* tt, const float* lb_col, int ln, int rk) {
  float z = 0.f;
  for (int j0 = 0; j0 < rk; j0 += 8) {
    float tv[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jj = min(j0 + j, rk - 1);
      tv[j] = tt[jj];
      bv[j] = lb_col[(size_t)jj * ln];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j0 + j < rk) z = fmaf(tv[j], bv[j], z);
  }
  return z;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// weight kinds for the generic kernels
enum : int {
  WK_F32 = 0, WK_BF16 = 1, WK_F16 = 2,
  WK_Q4_F32 = 3, WK_Q4_BF16 = 4, WK_Q4_F16 = 5,
  WK_Q8_F32 = 6, WK_Q8_BF16 = 7, WK_Q8_F16 = 8,
};
inline bool wk_is_quant(int wk) { return wk >= WK_Q4_F32; }
inline int wk_bits(int wk) { return wk >= WK_Q8_F32 ? 8 : (wk >= WK_Q4_F32 ? 4 : 0); }
inline int wk_scale_dtype(int wk) { return wk_is_quant(wk) ? (wk - 3) % 3 : wk; }

}  // namespace mi
