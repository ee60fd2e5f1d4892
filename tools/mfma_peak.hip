// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate (what the chip really holds under load).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && ./gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, int iters) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int grid = 256 * wg_per_cu;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  k<NACC><<<grid, 256>>>(out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(s);
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)grid * 4 * iters * 16 * NACC * 32 * 32 * 2 * 2;
    printf("NACC=%d wg/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, iters, ms, flops / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<4>(1, 2000);    // ~ms-scale
  run<4>(2, 2000);
  run<4>(2, 20000);   // ~10x longer: sustained clocks
  run<1>(2, 8000);
  return 0;
}
