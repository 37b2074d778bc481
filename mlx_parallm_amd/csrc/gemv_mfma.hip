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

#include "gemv_phase.h"

namespace mi {

#ifdef MI_SK_TRACE
unsigned long long* dbg_trace_slot(int N, int K, int grid, int epi, int M, int kind, int pro, int act);   // gemm_skinny.hip
#endif

namespace {

using namespace gemv;

template <typename AT, bool Q4, int MB, bool SWIGLU, int NW, int J, bool DB>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_kernel(MfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Phase<AT, Q4, MB, SWIGLU, NW, J, DB> ph(p, smem_raw);
  if (ph.ntiles <= 0) return;
  ph.template run<false, false, false>();
}

// The same body under its own symbol for the launches on a row-interleaved gate|up copy (EPI_SWIGLU_GU8): the dominant
// kernel of a decode step keeps a name of its own in kernel traces and counter passes (bench.py's roofline block reads it).
template <typename AT, int MB, int NW, int J, bool DB>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_gu8_kernel(MfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Phase<AT, false, MB, false, NW, J, DB> ph(p, smem_raw);
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
  seam_arrive(seam);
  Phase<AT, Q4, MB, SWB, NW, JB> B(pb, smem_raw);
  B.prefetch_weights();
  seam_wait(seam);
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
  if constexpr (!Q4 && !SWIGLU) {
    if (p.epi == EPI_SWIGLU_GU8) kern = gemv_mfma_gu8_kernel<AT, MB, NW, J, DB>;
  }
  constexpr int NA = SWIGLU ? 2 : 1;
  const size_t lds = phase_lds_bytes<NW, NA, Q4>(p.kc, MB, DB ? 2 : 1);
  MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nwg = std::min(p.N / 16, cu_count());        // one workgroup per CU; items are dealt in-kernel
#ifdef MI_SK_TRACE
  MfmaParams pt = p;
  pt.trace = dbg_trace_slot(p.N, p.K, nwg, p.epi, p.M, Q4 ? -4 : -1, p.pro, MI_BF16);
  if (g_ev_start != nullptr) hipExtLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, st, g_ev_start, g_ev_stop, 0, pt);
  else hipLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, st, pt);
  MI_HIP(hipGetLastError());
  return MI_OK;
#endif
  if (g_ev_start != nullptr)   // measurement: dispatch-level begin/end timestamps of THIS kernel
    hipExtLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, st, g_ev_start, g_ev_stop, 0, p);
  else
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, st, p);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <typename AT, bool Q4, int MB, bool SWIGLU, int NW>
int launch_one(const MfmaParams& p, hipStream_t st) {
  // make_params(): the chunk is the whole K up to 6144 (<= 8 rows) / 4096 (9..16 rows), else 4096 -- so only the
  // single-buffered 8-row form ever holds more than 8 * NW * 64 columns (J = 2).  The other J = 2 forms are not
  // instantiated: they would spill (tools/kernel_resources.py) and nothing selects them.
  if constexpr (MB == 8) {
    if (phase_nbuf(p.K, p.kc, MB, Q4) == 2) {    // K spans several activation chunks: the double-buffered instantiation
      if (p.kc > 8 * NW * 64) return fail(MI_ERR_INVALID, "gemv_mfma: a double-buffered chunk holds at most 4096 columns");
      return launch_j<AT, Q4, MB, SWIGLU, NW, 1, true>(p, st);
    }
    if (p.kc > 8 * NW * 64) return launch_j<AT, Q4, MB, SWIGLU, NW, 2, false>(p, st);
  } else {
    if (p.kc > 8 * NW * 64) return fail(MI_ERR_INVALID, "gemv_mfma: 9..16 rows hold at most 4096 columns per chunk");
  }
  return launch_j<AT, Q4, MB, SWIGLU, NW, 1, false>(p, st);
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

gemv::MfmaParams gemv_make_params(const LinearW& W, const GemvCall& c) { return make_params(W, c); }
int gemv_cu_count() { return cu_count(); }

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
  SeamParams seam{s.counter, s.base + (unsigned)cu_count(), s.spin_limit, s.error};
  g_ev_start = (hipEvent_t)a.ev_start; g_ev_stop = (hipEvent_t)a.ev_stop;
  const int rc = (a.act == MI_BF16) ? launch_pair_at<bf16>(q4, pa, pb, seam, st) : launch_pair_at<f16>(q4, pa, pb, seam, st);
  g_ev_start = g_ev_stop = nullptr;
  return rc;
}

int gemv_pair_grid() { return cu_count(); }

}  // namespace mi
