// chain_api.hip -- `mi_op_chain`, the C entry point of the layer-persistent decode-block EXPERIMENT (chain.hip).
// NOT part of libmi355_decode.so: the engine never launches this kernel (it measured 24 % slower than the four launches).
// tools/debug/build_chain_lib.sh links it, together with the product objects, into mlx_parallm_amd/csrc/alt/libmi355_chain.so
// for tools/debug/chain_probe.py and tools/debug/chain/test_chain.py.
#include "../../../include/mi355_ops.h"
#include "../../../mlx_parallm_amd/csrc/kernels.h"
#include "chain.h"

using namespace mi;

namespace {
int ready() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MI_ERR_RUNTIME, "no HIP device available");
  return MI_OK;
}
LinearW to_linear(const mi_op_linear* w) {
  LinearW W;
  W.wk = w->wk; W.w = w->w; W.scales = w->scales; W.biases = w->biases; W.N = w->N; W.K = w->K;
  W.group = w->group > 0 ? w->group : 64;
  W.layout = w->layout;
  return W;
}
GemvCall to_call(const mi_op_gemv_args* a) {
  GemvCall c;
  c.x = a->x; c.ldx = a->ldx; c.M = a->M; c.act = a->act; c.rnd = a->rnd; c.pro = a->pro; c.norm_w = a->norm_w;
  c.eps = a->eps; c.epi = a->epi; c.out = a->out; c.ldo = a->ldo; c.resid = a->resid; c.pair_offset = a->pair_offset;
  c.force_v1 = a->force_generic;
  return c;
}
}  // namespace

// up to 4 dependent linears (a[i].x may be what a[i-1] wrote: wait_prev[i] != 0) of a decode step of <= 8 rows as ONE
// persistent launch; iters >= 1 additionally times `iters` back-to-back launches (-> *avg_ms); *error_out = the kernel's
// give-up code (0 = every bounded wait was satisfied)
extern "C" int mi_op_chain(const mi_op_linear* const* w, const mi_op_gemv_args* a, const int32_t* wait_prev, int nops, int iters,
                           float* avg_ms, int32_t* error_out) {
  if (!w || !a || !wait_prev || nops < 1 || nops > 4) return fail(MI_ERR_INVALID, "bad argument");
  MI_TRY(ready());
  LinearW W[4]; GemvCall c[4]; const LinearW* Wp[4]; int wp[4];
  float* sq[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned* ctr = nullptr; int* err = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  for (int i = 0; i < nops; ++i) { W[i] = to_linear(w[i]); c[i] = to_call(&a[i]); Wp[i] = &W[i]; wp[i] = wait_prev[i]; }
  // (every check before any allocation; one cleanup block for every exit)
  for (int i = 0; i < nops; ++i)
    if (!chain_linear_ok(W[i], c[i])) return fail(MI_ERR_UNSUPPORTED, "mi_op_chain: linear not supported by the chain kernel");
  int rc = MI_OK, herr = 0;
  auto hip_ok = [&](hipError_t x, const char* what) {
    if (x != hipSuccess && rc == MI_OK) rc = fail(MI_ERR_RUNTIME, std::string(what) + ": " + hipGetErrorString(x));
    return x == hipSuccess;
  };
  // a normalised linear takes its row statistics from the residual epilogue in front of it (sums of h^2 per 16-column tile)
  for (int i = 1; i < nops && rc == MI_OK; ++i) {
    if (c[i].pro == PRO_NORM && wp[i] && c[i - 1].epi == EPI_RESID && c[i - 1].resid == c[i].x && W[i - 1].N == W[i].K) {
      const size_t bytes = (size_t)chain_sq_ld(W[i].K) * 8 * sizeof(float);
      if (!hip_ok(hipMalloc(&sq[i], bytes), "hipMalloc") || !hip_ok(hipMemset(sq[i], 0, bytes), "hipMemset")) break;
      c[i - 1].sq_out = sq[i]; c[i].sq_in = sq[i]; c[i].sq_parts = chain_sq_ld(W[i].K);
    }
  }
  if (rc == MI_OK) {
    hip_ok(hipMalloc(&ctr, CHAIN_CTR_WORDS * sizeof(unsigned)), "hipMalloc");
    hip_ok(hipMalloc(&err, sizeof(int)), "hipMalloc");
  }
  if (rc == MI_OK) {
    hip_ok(hipMemset(ctr, 0, CHAIN_CTR_WORDS * sizeof(unsigned)), "hipMemset");
    hip_ok(hipMemset(err, 0, sizeof(int)), "hipMemset");
  }
  unsigned base = 0;
  const unsigned spin = 1u << 22;
  if (rc == MI_OK) { rc = launch_chain(Wp, c, wp, nops, c[0].M, c[0].act, ctr, base, spin, err, nullptr); base += (unsigned)chain_grid(); }
  if (rc == MI_OK && iters >= 1 && avg_ms) {
    hipStreamSynchronize(nullptr);
    if (hip_ok(hipEventCreate(&e0), "hipEventCreate") && hip_ok(hipEventCreate(&e1), "hipEventCreate")) {
      hipEventRecord(e0, nullptr);
      for (int i = 0; i < iters && rc == MI_OK; ++i) { rc = launch_chain(Wp, c, wp, nops, c[0].M, c[0].act, ctr, base, spin, err, nullptr); base += (unsigned)chain_grid(); }
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      *avg_ms = ms / iters;
    }
  }
  if (rc == MI_OK) { hip_ok(hipGetLastError(), "launch"); hip_ok(hipStreamSynchronize(nullptr), "hipStreamSynchronize"); }
  else hipStreamSynchronize(nullptr);
  if (err) hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
  if (error_out) *error_out = herr;
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  hipFree(ctr); hipFree(err);
  for (int i = 0; i < 4; ++i) hipFree(sq[i]);
  return rc;
}
