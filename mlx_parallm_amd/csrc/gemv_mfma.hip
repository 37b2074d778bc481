// gemv_mfma.hip -- the hot decode kernel: weight-streaming skinny GEMM for M <= 16 rows on the
// CDNA4 matrix cores (v_mfma_f32_16x16x32_{bf16,f16}).
//
// Same operator as gemv_v1.hip (nn.Linear / nn.QuantizedLinear call sites llama.py:64-67,93,
// 143,160-165,250-252; qwen3.py:37-40,63,115 with RMSNorm / residual / SwiGLU fused), for the
// configurations BASELINE.json is quoted on: 16-bit activations with dense 16-bit weights or
// MLX-affine int4 (group 64) weights.  Why MFMA at M = 8: the op is HBM-bound (8 MACs per
// weight), and on the VALU the int4 unpack + 8 FMAs per weight would need ~80 % of the issue
// slots at the HBM rate; one MFMA per KiB of weights keeps the vector pipe nearly idle.
//
// Orientation: D[m][n] = sum_k A[m][k] B[k][n] with A = activations (from LDS), B = W^T
// (16 B per lane straight from HBM, non-temporal).  Lane l supplies W[n0 + (l&15)][k0 +
// 8(l>>4) .. +8] and receives y[m = 4(l>>4)+r][n0 + (l&15)], r = 0..3 -- so a quantisation
// scale (per weight row, per 64-k group) is a per-lane scalar.
//
// Workgroup = 256 threads = 4 waves = ONE 16-row tile of W (or one gate tile + its up tile);
// the four waves split K round-robin in 32-wide (dense) or 128-wide (int4) blocks, so that one
// step of the workgroup reads 256 contiguous bytes of each weight row; partial tiles are
// summed through LDS.  Activations are staged (RMSNorm applied) as 16-bit MFMA A-fragments.
#include <type_traits>

#include "kernels.h"

namespace mi {

namespace {

constexpr int NTHR = 256;
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
  int ntiles;   // 16-row tiles (SWIGLU: gate tiles)
  int kc;       // K elements staged in LDS per chunk
  const float* lora_t; int lora_t_ld;
  const float* lora_b0; const float* lora_b1;
  int lora_row0_0, lora_n_0, lora_rank_0; float lora_scale_0;
  int lora_row0_1, lora_n_1, lora_rank_1; float lora_scale_1;
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

template <typename AT, bool Q4, int MB, bool SWIGLU>
__global__ __launch_bounds__(NTHR, 2) void gemv_mfma_kernel(MfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // layout: [frag: kc*MB*2 bytes][sx: (kc/64)*MB floats (int4 only)][red: 4*64*4*(1|2) floats][rs: 16 floats]
  u32x4* frag = (u32x4*)smem_raw;
  float* sx = (float*)(smem_raw + (size_t)p.kc * MB * 2);
  float* red = sx + (Q4 ? (p.kc / 64) * MB : 0);
  float* rs_sh = red + 4 * 64 * 4 * (SWIGLU ? 2 : 1);
  float* red2 = rs_sh + 16;  // [4][16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const AT* x = (const AT*)p.x;
  using S = AT;  // scale dtype == activation dtype on this path

  if (p.pro == PRO_NORM) {
    float ss[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) ss[m] = 0.f;
    for (int k = tid * 8; k < p.K; k += NTHR * 8) {
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        if (m < p.M) {
          const u32x4 v = *(const u32x4*)(x + (size_t)m * p.ldx + k);
          const AT* e = (const AT*)&v;
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = (float)e[j]; ss[m] = fmaf(f, f, ss[m]); }
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
      const float v = red2[tid] + red2[16 + tid] + red2[32 + tid] + red2[48 + tid];
      rs_sh[tid] = 1.0f / sqrtf(v / (float)p.K + p.eps);
    }
    __syncthreads();
  }

  const int nchunks = (p.K + p.kc - 1) / p.kc;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int n0 = tile * 16;
    const int row0 = n0 + c16;                                   // this lane's weight row
    const int row1 = SWIGLU ? row0 + p.pair_offset : row0;

    for (int c = 0; c < nchunks; ++c) {
      const int kbase = c * p.kc;
      const int klen = min(p.kc, p.K - kbase);
      if (nchunks > 1 || tile == (int)blockIdx.x) {
        __syncthreads();
        // ---- stage x[:, kbase : kbase+klen] as MFMA A-fragments
        const int n8 = klen / 8;
        for (int idx = tid; idx < MB * n8; idx += NTHR) {
          const int m = idx / n8, k8 = idx % n8, k = kbase + k8 * 8;
          u32x4 v = {0u, 0u, 0u, 0u};
          float sum = 0.f;
          if (m < p.M) {
            v = *(const u32x4*)(x + (size_t)m * p.ldx + k);
            AT* e = (AT*)&v;
            if (p.pro == PRO_NORM) {
              const u32x4 wv = *(const u32x4*)((const AT*)p.norm_w + k);
              const AT* we = (const AT*)&wv;
              const float rs = rs_sh[m];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const AT xn = (AT)((float)e[j] * rs);              // cast_T(x32 * rsqrt(..))
                e[j] = (AT)((float)xn * (float)we[j]);            // w * (.)  in T
              }
            }
            if constexpr (Q4) {
#pragma unroll
              for (int j = 0; j < 8; ++j) sum += (float)e[j];
              // nibble order: element 2p <- k+p, element 2p+1 <- k+p+4
              AT t[8];
#pragma unroll
              for (int q = 0; q < 4; ++q) { t[2 * q] = e[q]; t[2 * q + 1] = e[q + 4]; }
#pragma unroll
              for (int j = 0; j < 8; ++j) e[j] = t[j];
            }
          }
          frag[frag_slot<Q4>(k8, m, MB)] = v;
          if constexpr (Q4) {
            // 8 consecutive lanes hold the 8 pieces of one 64-wide quantisation group
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            if ((k8 & 7) == 0) sx[(k8 >> 3) * MB + m] = sum;
          }
        }
        __syncthreads();
      }

      if constexpr (!Q4) {
        // ---- dense: this wave takes 32-wide k-blocks wave, wave+4, ...
        const int nkb = klen / 32;
        const AT* w0 = (const AT*)p.w + (size_t)row0 * p.K + kbase + g * 8;
        const AT* w1 = (const AT*)p.w + (size_t)row1 * p.K + kbase + g * 8;
        constexpr int U = 8;
        for (int kb0 = wave; kb0 < nkb; kb0 += 4 * U) {
          u32x4 b0[U], b1[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int kb = kb0 + 4 * u;
            if (kb < nkb) {
              b0[u] = __builtin_nontemporal_load((const u32x4*)(w0 + kb * 32));
              if constexpr (SWIGLU) b1[u] = __builtin_nontemporal_load((const u32x4*)(w1 + kb * 32));
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int kb = kb0 + 4 * u;
            if (kb < nkb) {
              u32x4 a = {0u, 0u, 0u, 0u};
              if (MB == 16 || c16 < MB) a = frag[(kb * 4 + g) * MB + c16];
              acc0 = mfma16<AT>(a, b0[u], acc0);
              if constexpr (SWIGLU) acc1 = mfma16<AT>(a, b1[u], acc1);
            }
          }
        }
      } else {
        // ---- int4 (group 64): this wave takes 128-wide k-blocks wave, wave+4, ...
        const int nkb = klen / 128;
        const int ng = p.K / 64;
        const uint32_t* w0 = (const uint32_t*)p.w + (size_t)row0 * (p.K / 8) + kbase / 8 + g * 4;
        const uint32_t* w1 = (const uint32_t*)p.w + (size_t)row1 * (p.K / 8) + kbase / 8 + g * 4;
        const S* sc0 = (const S*)p.scales + (size_t)row0 * ng + kbase / 64;
        const S* bi0 = (const S*)p.biases + (size_t)row0 * ng + kbase / 64;
        const S* sc1 = (const S*)p.scales + (size_t)row1 * ng + kbase / 64;
        const S* bi1 = (const S*)p.biases + (size_t)row1 * ng + kbase / 64;
        constexpr int U = 4;
        for (int kb0 = wave; kb0 < nkb; kb0 += 4 * U) {
          u32x4 q0[U], q1[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int kb = kb0 + 4 * u;
            if (kb < nkb) {
              q0[u] = __builtin_nontemporal_load((const u32x4*)(w0 + kb * 16));
              if constexpr (SWIGLU) q1[u] = __builtin_nontemporal_load((const u32x4*)(w1 + kb * 16));
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int kb = kb0 + 4 * u;
            if (kb < nkb) {
              // lanes g<2 hold quant group A (dwords 0-7 of the 128 block), g>=2 group B.  Swap
              // the upper half's {x,y} with the lower half's {z,w}: afterwards {x,y} = group A
              // and {z,w} = group B in EVERY lane (v_permlane32_swap).
              u32x4 v = q0[u];
              {
                auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
                v.x = r0[0]; v.z = r0[1]; v.y = r1[0]; v.w = r1[1];
              }
              u32x4 vv = v;
              if constexpr (SWIGLU) {
                vv = q1[u];
                auto r0 = __builtin_amdgcn_permlane32_swap(vv.x, vv.z, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(vv.y, vv.w, false, false);
                vv.x = r0[0]; vv.z = r0[1]; vv.y = r1[0]; vv.w = r1[1];
              }
              const uint32_t dw0[4] = {v.x, v.y, v.z, v.w};
              const uint32_t dw1[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
              for (int s = 0; s < 2; ++s) {
                const int gi = kb * 2 + s;
                f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  u32x4 a = {0u, 0u, 0u, 0u};
                  if (MB == 16 || c16 < MB) a = frag[((((kb * 2 + s) * 2 + t) * 4 + g) * MB + c16)];
                  d0 = mfma16<AT>(a, unpack_q4<AT>(dw0[s * 2 + t]), d0);
                  if constexpr (SWIGLU) d1 = mfma16<AT>(a, unpack_q4<AT>(dw1[s * 2 + t]), d1);
                }
                // y += scale * sum((OFFS+q) x) + (bias - OFFS*scale) * sum(x)   per output row m
                const float s0 = (float)sc0[gi], bb0 = (float)bi0[gi] - Magic<AT>::offs * s0;
                f32x4 sxv = {0.f, 0.f, 0.f, 0.f};
                if (MB == 16 || g * 4 < MB) sxv = *(const f32x4*)&sx[gi * MB + g * 4];
                acc0.x = fmaf(s0, d0.x, fmaf(bb0, sxv.x, acc0.x));
                acc0.y = fmaf(s0, d0.y, fmaf(bb0, sxv.y, acc0.y));
                acc0.z = fmaf(s0, d0.z, fmaf(bb0, sxv.z, acc0.z));
                acc0.w = fmaf(s0, d0.w, fmaf(bb0, sxv.w, acc0.w));
                if constexpr (SWIGLU) {
                  const float s1 = (float)sc1[gi], bb1 = (float)bi1[gi] - Magic<AT>::offs * s1;
                  acc1.x = fmaf(s1, d1.x, fmaf(bb1, sxv.x, acc1.x));
                  acc1.y = fmaf(s1, d1.y, fmaf(bb1, sxv.y, acc1.y));
                  acc1.z = fmaf(s1, d1.z, fmaf(bb1, sxv.z, acc1.z));
                  acc1.w = fmaf(s1, d1.w, fmaf(bb1, sxv.w, acc1.w));
                }
              }
            }
          }
        }
      }
    }

    // ---- sum the four waves' partial tiles through LDS; thread (lane, r) finishes y[4g+r][n0+c16]
    constexpr int NA = SWIGLU ? 2 : 1;
    *(f32x4*)&red[((wave * NA + 0) * 64 + lane) * 4] = acc0;
    if constexpr (SWIGLU) *(f32x4*)&red[((wave * NA + 1) * 64 + lane) * 4] = acc1;
    __syncthreads();
    {
      const int el = tid & 63, r = tid >> 6;
      const int m = 4 * (el >> 4) + r, n = n0 + (el & 15);
      float y0 = 0.f, y1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        y0 += red[((w * NA + 0) * 64 + el) * 4 + r];
        if constexpr (SWIGLU) y1 += red[((w * NA + 1) * 64 + el) * 4 + r];
      }
      if (m < p.M && n < p.N) {
        AT* out = (AT*)p.out;
        if constexpr (SWIGLU) {
          const float gt = (float)(AT)y0, up = (float)(AT)y1;
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
                const float* t = p.lora_t + (size_t)m * p.lora_t_ld + sl * (p.lora_t_ld / 2);
                float z = 0.f;
                for (int j = 0; j < rk; ++j) z = fmaf(t[j], lb[(size_t)j * ln + (n - r0)], z);
                z = (sl ? p.lora_scale_1 : p.lora_scale_0) * z;
                y = (float)(AT)(y + (float)(AT)z);
              }
            }
          }
          if (p.epi == EPI_STORE) out[(size_t)m * p.ldo + n] = (AT)y;
          else if (p.epi == EPI_STORE_F32) ((float*)p.out)[(size_t)m * p.ldo + n] = y;
          else {
            AT* h = (AT*)p.resid;
            h[(size_t)m * p.ldo + n] = (AT)((float)h[(size_t)m * p.ldo + n] + y);
          }
        }
      }
    }
    __syncthreads();  // `red` is reused by the next tile
  }
}

template <typename AT, bool Q4, int MB, bool SWIGLU>
int launch_one(const MfmaParams& p, int nwg, hipStream_t st) {
  auto kern = gemv_mfma_kernel<AT, Q4, MB, SWIGLU>;
  const size_t lds = (size_t)p.kc * MB * 2 + (Q4 ? (size_t)(p.kc / 64) * MB * 4 : 0) +
                     (size_t)4 * 64 * 4 * (SWIGLU ? 2 : 1) * 4 + 16 * 4 + 64 * 4;
  MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHR), lds, st, p);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename AT>
int launch_at(bool q4, const MfmaParams& p, int nwg, hipStream_t st) {
  const bool sw = p.epi == EPI_SWIGLU;
  const bool m8 = p.M <= 8;
  if (!q4) {
    if (m8) return sw ? launch_one<AT, false, 8, true>(p, nwg, st) : launch_one<AT, false, 8, false>(p, nwg, st);
    return sw ? launch_one<AT, false, 16, true>(p, nwg, st) : launch_one<AT, false, 16, false>(p, nwg, st);
  }
  if (m8) return sw ? launch_one<AT, true, 8, true>(p, nwg, st) : launch_one<AT, true, 8, false>(p, nwg, st);
  return sw ? launch_one<AT, true, 16, true>(p, nwg, st) : launch_one<AT, true, 16, false>(p, nwg, st);
}

}  // namespace

// true when the MFMA path can run this call
bool gemv_mfma_supported(const LinearW& W, const GemvCall& c) {
  if (c.force_v1) return false;
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
  MfmaParams p{};
  p.x = c.x; p.ldx = c.ldx; p.M = c.M; p.pro = c.pro; p.norm_w = c.norm_w; p.eps = c.eps;
  p.w = W.w; p.scales = W.scales; p.biases = W.biases; p.N = (c.epi == EPI_SWIGLU) ? c.pair_offset : W.N; p.K = W.K;
  p.epi = c.epi; p.out = c.out; p.ldo = c.ldo; p.resid = c.resid; p.pair_offset = c.pair_offset;
  p.ntiles = p.N / 16;
  const int MB = c.M <= 8 ? 8 : 16;
  // K chunk: whole K when the fragments fit in 64 KiB (two workgroups per CU), else 4096/2048
  const int kc_max = (64 * 1024) / (MB * 2);
  p.kc = (W.K <= kc_max) ? W.K : kc_max;
  p.lora_t = c.lora_t; p.lora_t_ld = c.lora_t_ld;
  p.lora_b0 = W.lora_b[0]; p.lora_b1 = W.lora_b[1];
  p.lora_row0_0 = W.lora_row0[0]; p.lora_n_0 = W.lora_n[0]; p.lora_rank_0 = W.lora_rank[0]; p.lora_scale_0 = W.lora_scale[0];
  p.lora_row0_1 = W.lora_row0[1]; p.lora_n_1 = W.lora_n[1]; p.lora_rank_1 = W.lora_rank[1]; p.lora_scale_1 = W.lora_scale[1];
  const int nwg = p.ntiles;
  if (c.act == MI_BF16) return launch_at<bf16>(q4, p, nwg, st);
  return launch_at<f16>(q4, p, nwg, st);
}

}  // namespace mi
