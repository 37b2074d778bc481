// mfma_peak.hip -- yardstick, not part of the product: the rate at which this part issues bf16 MFMAs from registers
// (no memory traffic), to put the prefill GEMM's TFLOP/s in proportion.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k16(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, int waves, double flop_per_wave_iter, int iters) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 2;
  float* out = nullptr;
  hipMalloc(&out, (size_t)blocks * waves * 64 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * waves * iters * flop_per_wave_iter;
  printf("%-28s %2d waves/WG x %d WGs: %8.3f ms  %7.0f TFLOP/s\n", name, waves, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  const int iters = 20000;
  run("mfma_f32_16x16x32_bf16", k16<4>, 4, 16.0 * 16384.0, iters);
  run("mfma_f32_16x16x32_bf16", k16<8>, 8, 16.0 * 16384.0, iters);
  run("mfma_f32_32x32x16_bf16", k32<4>, 4, 4.0 * 32768.0, iters);
  run("mfma_f32_32x32x16_bf16", k32<8>, 8, 4.0 * 32768.0, iters);
  return 0;
}
