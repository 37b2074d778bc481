// gemv_v1.hip -- generic weight-streaming skinny GEMM  y[M][N] = x[M][K] . W[N][K]^T  for M <= 8.
//
// Replaces nn.Linear / nn.QuantizedLinear (= x @ W.T / mx.quantized_matmul) at the call sites
// llama.py:64-67,93,143,160-165,250-252 and qwen3.py:37-40,63,115, with the neighbouring
// element-wise ops fused in: RMSNorm prologue (llama.py:175-177,205), residual add
// (llama.py:188,190), SwiGLU (llama.py:165), LoRALinear epilogue (mlx-lm, utils.py:742-744).
//
// This is the exact-fp32 VALU version that handles EVERY dtype combination (float32 /
// bfloat16 / float16 activations; dense or MLX-affine int4/int8 weights with scales in any
// of the three types).  The 16-bit hot configurations go through gemv_mfma.hip instead.
//
// Structure: one 256-thread workgroup = 4 waves; every wave owns PW row PAIRS of W and walks
// the whole K dimension; the 64 lanes split K in 16-byte pieces (coalesced 1 KiB per
// wave-instruction, weights go HBM -> VGPR directly, never through LDS); the activations are
// staged once per workgroup into LDS as fp32 in K-chunks of KC; fp32 accumulate; one
// butterfly reduction per row at the end.
#include "kernels.h"

namespace mi {

namespace {

constexpr int KC = 2048;  // K elements of x held in LDS at a time
constexpr int PW = 2;     // row pairs per wave
constexpr int NTHR = 256;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemvParams {
  const void* x; int ldx; int M;
  int rnd; int pro; const void* norm_w; float eps;
  const void* w; const void* scales; const void* biases;
  int N, K, group, layout;
  int epi; void* out; int ldo; void* resid; int pair_offset;
  int npairs;
  // LoRA
  const float* lora_t; int lora_t_ld;
  const float* lora_b0; const float* lora_b1;
  int lora_row0_0, lora_n_0, lora_rank_0; float lora_scale_0;
  int lora_row0_1, lora_n_1, lora_rank_1; float lora_scale_1;
  int lora_rnd;
};

// ---- 8 consecutive weights of one row, as fp32 -------------------------------------------
template <int WK>
__device__ __forceinline__ void load8(const GemvParams& p, int row, int kb, float (&o)[8]) {
  if constexpr (WK == WK_F32) {
    const float* base = (const float*)p.w + (size_t)row * p.K + kb;
    f32x4 a = __builtin_nontemporal_load((const f32x4*)base);
    f32x4 b = __builtin_nontemporal_load((const f32x4*)(base + 4));
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else if constexpr (WK == WK_BF16) {
    const uint16_t* base = p.layout ? (const uint16_t*)((const char*)p.w + tiled_piece_dense16(row, kb, p.K))
                                    : (const uint16_t*)p.w + (size_t)row * p.K + kb;
    u32x4 v = __builtin_nontemporal_load((const u32x4*)base);
    o[0] = bf16lo(v.x); o[1] = bf16hi(v.x); o[2] = bf16lo(v.y); o[3] = bf16hi(v.y);
    o[4] = bf16lo(v.z); o[5] = bf16hi(v.z); o[6] = bf16lo(v.w); o[7] = bf16hi(v.w);
  } else if constexpr (WK == WK_F16) {
    const f16* base = p.layout ? (const f16*)((const char*)p.w + tiled_piece_dense16(row, kb, p.K))
                               : (const f16*)p.w + (size_t)row * p.K + kb;
    u32x4 v = __builtin_nontemporal_load((const u32x4*)base);
    const f16* h = (const f16*)&v;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)h[j];
  } else {
    constexpr int BITS = (WK >= WK_Q8_F32) ? 8 : 4;
    constexpr int SDT = (WK - 3) % 3;  // 0 f32, 1 bf16, 2 f16
    const int ng = p.K / p.group;
    const size_t gi = (size_t)row * ng + kb / p.group;
    float s, b;
    const char* blk = nullptr;
    if (BITS == 4 && SDT != 0 && p.layout) {       // tile-major int4 (repack.hip): scales live in the block
      blk = (const char*)p.w + tiled_block_q4(row, kb, p.K);
      const int so = tiled_q4_scale_off(row, kb);
      if constexpr (SDT == 1) { s = (float)*(const bf16*)(blk + so); b = (float)*(const bf16*)(blk + so + 64); }
      else { s = (float)*(const f16*)(blk + so); b = (float)*(const f16*)(blk + so + 64); }
    } else if (BITS == 8 && SDT != 0 && p.layout) {  // tile-major int8
      blk = (const char*)p.w + tiled_block_q8(row, kb, p.K);
      const int so = tiled_q8_scale_off(row);
      if constexpr (SDT == 1) { s = (float)*(const bf16*)(blk + so); b = (float)*(const bf16*)(blk + so + 32); }
      else { s = (float)*(const f16*)(blk + so); b = (float)*(const f16*)(blk + so + 32); }
    } else if constexpr (SDT == 0) { s = ((const float*)p.scales)[gi]; b = ((const float*)p.biases)[gi]; }
    else if constexpr (SDT == 1) { s = (float)((const bf16*)p.scales)[gi]; b = (float)((const bf16*)p.biases)[gi]; }
    else { s = (float)((const f16*)p.scales)[gi]; b = (float)((const f16*)p.biases)[gi]; }
    if constexpr (BITS == 4) {
      const uint32_t* base = blk ? (const uint32_t*)(blk + tiled_q4_code_off(row, kb))
                                 : (const uint32_t*)p.w + (size_t)row * (p.K / 8) + kb / 8;
      uint32_t v = __builtin_nontemporal_load(base);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = mul_add_unfused((float)((v >> (4 * j)) & 15u), s, b);
    } else {
      const uint32_t* base = blk ? (const uint32_t*)(blk + tiled_q8_code_off(row, kb))
                                 : (const uint32_t*)p.w + (size_t)row * (p.K / 4) + kb / 4;
      u32x2 v = __builtin_nontemporal_load((const u32x2*)base);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = mul_add_unfused((float)((v.x >> (8 * j)) & 255u), s, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[4 + j] = mul_add_unfused((float)((v.y >> (8 * j)) & 255u), s, b);
    }
  }
}

// LDS address (in floats) of x[m][kk] (kk inside the chunk): the 8 k's a lane needs for one
// 512-wide step are stored as two 16-byte halves [half][lane] so that each ds_read_b128 of a
// wave is 1 KiB contiguous (conflict free).
__device__ __forceinline__ int xs_addr(int m, int kk) {
  const int ks = kk >> 9, ln = (kk & 511) >> 3, half = (kk & 7) >> 2, j = kk & 3;
  return ((((m * (KC / 512) + ks) * 2 + half) * 64 + ln) << 2) + j;
}

template <typename AT, int WK, int MT>
__global__ __launch_bounds__(NTHR) void gemv_v1_kernel(GemvParams p) {
  extern __shared__ float smem[];
  float* xs = smem;                 // [MT][KC] permuted
  float* rs_sh = smem + MT * KC;    // [MT] rsqrt(mean(x^2)+eps)
  float* red = rs_sh + MT;          // [4][MT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const AT* x = (const AT*)p.x;

  if (p.pro == PRO_NORM) {
    float ss[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ss[m] = 0.f;
    for (int k = tid; k < p.K; k += NTHR) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m < p.M) { float v = to_f32(x[(size_t)m * p.ldx + k]); ss[m] += v * v; }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v = wave_sum(ss[m]);
      if (lane == 0) red[wave * MT + m] = v;
    }
    __syncthreads();
    if (tid < MT) {
      float v = red[tid] + red[MT + tid] + red[2 * MT + tid] + red[3 * MT + tid];
      rs_sh[tid] = 1.0f / sqrtf(v / (float)p.K + p.eps);
    }
    __syncthreads();
  }

  float acc[PW][2][MT];
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[a][r][m] = 0.f;

  int rows[PW][2];
  bool pvalid[PW];
#pragma unroll
  for (int a = 0; a < PW; ++a) {
    const int pair = (blockIdx.x * 4 + wave) * PW + a;
    pvalid[a] = pair < p.npairs;
    const int pp = pvalid[a] ? pair : 0;
    if (p.epi == EPI_SWIGLU) { rows[a][0] = pp; rows[a][1] = pp + p.pair_offset; }
    else { rows[a][0] = 2 * pp; rows[a][1] = min(2 * pp + 1, p.N - 1); }
  }

  const int nchunks = (p.K + KC - 1) / KC;
  for (int c = 0; c < nchunks; ++c) {
    if (c > 0) __syncthreads();
    // ---- stage x[:, c*KC : (c+1)*KC] (with the RMSNorm applied) into LDS as fp32
    for (int idx = tid * 4; idx < MT * KC; idx += NTHR * 4) {
      const int m = idx / KC, kk = idx % KC, k = c * KC + kk;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (m < p.M && k < p.K) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = to_f32(x[(size_t)m * p.ldx + k + j]);
          if (p.pro == PRO_NORM) {
            // w * cast_T(x32 * rsqrt(mean + eps))   
            AT xn = store_act<AT>(t * rs_sh[m], p.rnd);
            t = to_f32(store_act<AT>(to_f32(xn) * to_f32(((const AT*)p.norm_w)[k + j]), p.rnd));
          }
          v[j] = t;
        }
      }
      *(f32x4*)&xs[xs_addr(m, kk)] = f32x4{v[0], v[1], v[2], v[3]};
    }
    __syncthreads();

    const int ksteps = min(KC, p.K - c * KC);
    for (int ks = 0; ks * 512 < ksteps; ++ks) {
      const int kk = ks * 512 + lane * 8;
      const int kb = c * KC + kk;
      if (kb < p.K) {
        float w8[PW][2][8];
#pragma unroll
        for (int a = 0; a < PW; ++a)
#pragma unroll
          for (int r = 0; r < 2; ++r) load8<WK>(p, rows[a][r], kb, w8[a][r]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int base = (((m * (KC / 512) + ks) * 2) * 64 + lane) << 2;
          const f32x4 x0 = *(const f32x4*)&xs[base];
          const f32x4 x1 = *(const f32x4*)&xs[base + 256];
#pragma unroll
          for (int a = 0; a < PW; ++a)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              float s = acc[a][r][m];
              s = fmaf(w8[a][r][0], x0.x, s); s = fmaf(w8[a][r][1], x0.y, s);
              s = fmaf(w8[a][r][2], x0.z, s); s = fmaf(w8[a][r][3], x0.w, s);
              s = fmaf(w8[a][r][4], x1.x, s); s = fmaf(w8[a][r][5], x1.y, s);
              s = fmaf(w8[a][r][6], x1.z, s); s = fmaf(w8[a][r][7], x1.w, s);
              acc[a][r][m] = s;
            }
        }
      }
    }
  }

  // ---- reduce over the 64 lanes; lane m finishes batch row m
#pragma unroll
  for (int a = 0; a < PW; ++a) {
    float mine[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float v = wave_sum(acc[a][r][m]);
        if (lane == m) mine[r] = v;
      }
    if (!pvalid[a] || lane >= p.M || lane >= MT) continue;
    const int m = lane;
    AT* out = (AT*)p.out;
    if (p.epi == EPI_SWIGLU) {
      // nn.silu(gate) * up, every op rounded to the activation dtype (llama.py:165)
      const float g = to_f32(store_act<AT>(mine[0], p.rnd));
      const float u = to_f32(store_act<AT>(mine[1], p.rnd));
      const float sig = to_f32(store_act<AT>(1.0f / (1.0f + expf(-g)), p.rnd));
      const float s = to_f32(store_act<AT>(g * sig, p.rnd));
      out[(size_t)m * p.ldo + rows[a][0]] = store_act<AT>(s * u, p.rnd);
      continue;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int n = rows[a][r];
      if (r == 1 && 2 * ((blockIdx.x * 4 + wave) * PW + a) + 1 >= p.N) continue;
      float y = to_f32(store_act<AT>(mine[r], p.rnd));
      // LoRALinear: y + (scale * ((x A) B)).astype(x.dtype)
      if (p.lora_t != nullptr) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int row0 = sl ? p.lora_row0_1 : p.lora_row0_0;
          const int ln = sl ? p.lora_n_1 : p.lora_n_0;
          const int rk = sl ? p.lora_rank_1 : p.lora_rank_0;
          const float* lb = sl ? p.lora_b1 : p.lora_b0;
          if (lb != nullptr && n >= row0 && n < row0 + ln) {
            const float* t = p.lora_t + (size_t)m * p.lora_t_ld + sl * (p.lora_t_ld / 2);
            float z = 0.f;
            for (int j = 0; j < rk; ++j) z = fmaf(t[j], lb[(size_t)j * ln + (n - row0)], z);
            z = round_rt(z, p.lora_rnd);
            z = round_rt((sl ? p.lora_scale_1 : p.lora_scale_0) * z, p.lora_rnd);
            z = to_f32(store_act<AT>(z, p.rnd));
            y = to_f32(store_act<AT>(y + z, p.rnd));
          }
        }
      }
      if (p.epi == EPI_STORE) {
        out[(size_t)m * p.ldo + n] = from_f32<AT>(y);
      } else if (p.epi == EPI_STORE_F32) {
        ((float*)p.out)[(size_t)m * p.ldo + n] = y;
      } else {  // EPI_RESID
        AT* h = (AT*)p.resid;
        const float hv = to_f32(h[(size_t)m * p.ldo + n]);
        h[(size_t)m * p.ldo + n] = store_act<AT>(hv + y, p.rnd);
      }
    }
  }
}

// t[m][slot][j] = round(sum_k xin[m][k] * A[k][j]); xin = (normed) x, same prologue as above.
template <typename AT>
__global__ __launch_bounds__(NTHR) void lora_down_kernel(GemvParams p, const float* a0, const float* a1,
                                                          float* t, int t_ld) {
  __shared__ float rs_sh;
  __shared__ float red[4];
  const int m = blockIdx.x, slot = blockIdx.y;
  const float* A = slot ? a1 : a0;
  const int rk = slot ? p.lora_rank_1 : p.lora_rank_0;
  if (A == nullptr) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const AT* x = (const AT*)p.x + (size_t)m * p.ldx;
  if (p.pro == PRO_NORM) {
    float ss = 0.f;
    for (int k = tid; k < p.K; k += NTHR) { float v = to_f32(x[k]); ss += v * v; }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (tid == 0) rs_sh = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)p.K + p.eps);
    __syncthreads();
  }
  // 16 columns of A at a time: one pass over x per 16 ranks instead of one per rank (the per-thread order of the
  // k sum, the wave reduction and the order of the four wave partials are those of a single-column pass, so t is
  // unchanged bit for bit)
  __shared__ float red16[4][16];
  for (int j0 = 0; j0 < rk; j0 += 16) {
    float s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = 0.f;
    // four k per trip with all their loads issued before the first fma (a thread's 20 k at K = 5120 were 20
    // dependent L2 round trips: 37 us per launch); the fma order per column is still k ascending
    const bool vec = (rk % 4 == 0) && (j0 + 16 <= rk);
    for (int k0 = tid; k0 < p.K; k0 += 4 * NTHR) {
      float v[4], nw[4];
      float4 a4[4][4];
      float a1[4][16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * NTHR, kc = k < p.K ? k : tid;
        v[u] = to_f32(x[kc]);
        nw[u] = p.pro == PRO_NORM ? to_f32(((const AT*)p.norm_w)[kc]) : 1.0f;
        const float* ar = A + (size_t)kc * rk + j0;
        if (vec) {
#pragma unroll
          for (int q = 0; q < 4; ++q) a4[u][q] = ((const float4*)ar)[q];
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) a1[u][j] = ar[j0 + j < rk ? j : 0];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * NTHR;
        if (k >= p.K) break;
        float vv = v[u];
        if (p.pro == PRO_NORM) {
          AT xn = store_act<AT>(vv * rs_sh, p.rnd);
          vv = to_f32(store_act<AT>(to_f32(xn) * nw[u], p.rnd));
        }
        if (vec) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s[4 * q + 0] = fmaf(vv, a4[u][q].x, s[4 * q + 0]);
            s[4 * q + 1] = fmaf(vv, a4[u][q].y, s[4 * q + 1]);
            s[4 * q + 2] = fmaf(vv, a4[u][q].z, s[4 * q + 2]);
            s[4 * q + 3] = fmaf(vv, a4[u][q].w, s[4 * q + 3]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (j0 + j < rk) s[j] = fmaf(vv, a1[u][j], s[j]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float w = wave_sum(s[j]);
      if (lane == 0) red16[wave][j] = w;
    }
    __syncthreads();
    if (tid < 16 && j0 + tid < rk)
      t[(size_t)m * t_ld + slot * (t_ld / 2) + j0 + tid] =
          round_rt(red16[0][tid] + red16[1][tid] + red16[2][tid] + red16[3][tid], p.lora_rnd);
  }
}

// y[m][n] = T(y[m][n] + T(scale * sum_j t[m][slot][j] * B[j][n - row0]))  for the adapted column ranges: the LoRA term
// of the GEMV epilogues, applied after a tile GEMM has stored y = T(x W^T) (prefill)
template <typename AT>
__global__ __launch_bounds__(256) void lora_up_add_kernel(GemvParams p, const float* t, int t_ld, AT* y, int ldy) {
  const int m = blockIdx.x;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const float* lb = sl ? p.lora_b1 : p.lora_b0;
    if (lb == nullptr) continue;
    const int r0 = sl ? p.lora_row0_1 : p.lora_row0_0;
    const int ln = sl ? p.lora_n_1 : p.lora_n_0;
    const int rk = sl ? p.lora_rank_1 : p.lora_rank_0;
    const float sc = sl ? p.lora_scale_1 : p.lora_scale_0;
    const float* tt = t + (size_t)m * t_ld + sl * (t_ld / 2);
    for (int n = threadIdx.x; n < ln; n += 256) {
      float z = lora_dot(tt, lb + n, ln, rk);
      z = sc * z;
      AT* o = y + (size_t)m * ldy + r0 + n;
      if constexpr (sizeof(AT) == 4) *o = round_rt((float)*o + round_rt(z, p.rnd), p.rnd);   // float32 storage, logical rounding (layer 0 of the PagedKVCache mode)
      else *o = (AT)((float)*o + (float)(AT)z);
    }
  }
}

template <typename AT, int WK>
int launch_mt(const GemvParams& p, hipStream_t st) {
  const int nwg = (p.npairs + 4 * PW - 1) / (4 * PW);
  auto go = [&](auto kern, int MT) -> int {
    const size_t lds = (size_t)(MT * KC + MT + 4 * MT) * sizeof(float);
    static thread_local const void* configured[4] = {nullptr, nullptr, nullptr, nullptr};
    (void)configured;
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHR), lds, st, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
  };
  if (p.M <= 1) return go(gemv_v1_kernel<AT, WK, 1>, 1);
  if (p.M <= 4) return go(gemv_v1_kernel<AT, WK, 4>, 4);
  return go(gemv_v1_kernel<AT, WK, 8>, 8);
}

template <typename AT>
int launch_wk(int wk, const GemvParams& p, hipStream_t st) {
  constexpr bool F32A = sizeof(AT) == 4;
  switch (wk) {
    case WK_F32: if constexpr (F32A) return launch_mt<AT, WK_F32>(p, st); break;
    case WK_BF16: if constexpr (F32A || std::is_same<AT, bf16>::value) return launch_mt<AT, WK_BF16>(p, st); break;
    case WK_F16: if constexpr (F32A || std::is_same<AT, f16>::value) return launch_mt<AT, WK_F16>(p, st); break;
    case WK_Q4_F32: if constexpr (F32A) return launch_mt<AT, WK_Q4_F32>(p, st); break;
    case WK_Q8_F32: if constexpr (F32A) return launch_mt<AT, WK_Q8_F32>(p, st); break;
    case WK_Q4_BF16: if constexpr (F32A || std::is_same<AT, bf16>::value) return launch_mt<AT, WK_Q4_BF16>(p, st); break;
    case WK_Q8_BF16: if constexpr (F32A || std::is_same<AT, bf16>::value) return launch_mt<AT, WK_Q8_BF16>(p, st); break;
    case WK_Q4_F16: if constexpr (F32A || std::is_same<AT, f16>::value) return launch_mt<AT, WK_Q4_F16>(p, st); break;
    case WK_Q8_F16: if constexpr (F32A || std::is_same<AT, f16>::value) return launch_mt<AT, WK_Q8_F16>(p, st); break;
  }
  return fail(MI_ERR_UNSUPPORTED, "gemv: weight kind / activation dtype combination not supported");
}

GemvParams make_params(const LinearW& W, const GemvCall& c) {
  GemvParams p{};
  p.x = c.x; p.ldx = c.ldx; p.M = c.M; p.rnd = c.rnd; p.pro = c.pro; p.norm_w = c.norm_w; p.eps = c.eps;
  p.w = W.w; p.scales = W.scales; p.biases = W.biases; p.N = W.N; p.K = W.K; p.group = W.group; p.layout = W.layout;
  p.epi = c.epi; p.out = c.out; p.ldo = c.ldo; p.resid = c.resid; p.pair_offset = c.pair_offset;
  p.npairs = (c.epi == EPI_SWIGLU) ? c.pair_offset : (W.N + 1) / 2;
  p.lora_t = c.lora_t; p.lora_t_ld = c.lora_t_ld;
  p.lora_b0 = W.lora_b[0]; p.lora_b1 = W.lora_b[1];
  p.lora_row0_0 = W.lora_row0[0]; p.lora_n_0 = W.lora_n[0]; p.lora_rank_0 = W.lora_rank[0]; p.lora_scale_0 = W.lora_scale[0];
  p.lora_row0_1 = W.lora_row0[1]; p.lora_n_1 = W.lora_n[1]; p.lora_rank_1 = W.lora_rank[1]; p.lora_scale_1 = W.lora_scale[1];
  p.lora_rnd = RND_NONE;
  return p;
}

}  // namespace

int launch_gemv_v1(const LinearW& W, const GemvCall& c, hipStream_t st) {
  if (c.M < 1 || c.M > 8) return fail(MI_ERR_INVALID, "gemv_v1: M must be in [1,8]");
  if (W.K % 8 != 0) return fail(MI_ERR_UNSUPPORTED, "gemv: K must be a multiple of 8");
  if (wk_is_quant(W.wk) && (W.group % 8 != 0 || W.K % W.group != 0))
    return fail(MI_ERR_UNSUPPORTED, "gemv: bad quantisation group size");
  if (c.epi == EPI_SWIGLU && c.pair_offset * 2 != W.N) return fail(MI_ERR_INVALID, "gemv: swiglu needs N = 2*pair_offset");
  GemvParams p = make_params(W, c);
  switch (c.act) {
    case MI_F32: return launch_wk<float>(W.wk, p, st);
    case MI_BF16: return launch_wk<bf16>(W.wk, p, st);
    case MI_F16: return launch_wk<f16>(W.wk, p, st);
  }
  return fail(MI_ERR_INVALID, "gemv: bad activation dtype");
}

int launch_lora_down(const LinearW& W, const GemvCall& c, float* t, int t_ld, hipStream_t st) {
  GemvParams p = make_params(W, c);
  dim3 grid(c.M, 2);
  switch (c.act) {
    case MI_F32: hipLaunchKernelGGL(lora_down_kernel<float>, grid, dim3(NTHR), 0, st, p, W.lora_a[0], W.lora_a[1], t, t_ld); break;
    case MI_BF16: hipLaunchKernelGGL(lora_down_kernel<bf16>, grid, dim3(NTHR), 0, st, p, W.lora_a[0], W.lora_a[1], t, t_ld); break;
    case MI_F16: hipLaunchKernelGGL(lora_down_kernel<f16>, grid, dim3(NTHR), 0, st, p, W.lora_a[0], W.lora_a[1], t, t_ld); break;
    default: return fail(MI_ERR_INVALID, "lora_down: bad activation dtype");
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// c.out ([rows][ldo], EPI_STORE) += the LoRA term, rows = c.M, from t = launch_lora_down's output
int launch_lora_up_add(const LinearW& W, const GemvCall& c, const float* t, int t_ld, hipStream_t st) {
  if (c.epi != EPI_STORE) return fail(MI_ERR_UNSUPPORTED, "lora_up_add: plain store epilogue only");
  GemvParams p = make_params(W, c);
  const dim3 grid(c.M), block(256);
  switch (c.act) {
    case MI_BF16: hipLaunchKernelGGL(lora_up_add_kernel<bf16>, grid, block, 0, st, p, t, t_ld, (bf16*)c.out, c.ldo); break;
    case MI_F16: hipLaunchKernelGGL(lora_up_add_kernel<f16>, grid, block, 0, st, p, t, t_ld, (f16*)c.out, c.ldo); break;
    case MI_F32: hipLaunchKernelGGL(lora_up_add_kernel<float>, grid, block, 0, st, p, t, t_ld, (float*)c.out, c.ldo); break;
    default: return fail(MI_ERR_UNSUPPORTED, "lora_up_add: bad activation dtype");
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
