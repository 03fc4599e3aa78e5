// Sustained fp32-MFMA rate of this device: a register-only loop of v_mfma_f32_32x32x2_f32 (no memory traffic), so
// that kernel efficiencies can be read against what the chip sustains under load as well as against the 2.4 GHz
// nameplate figure (157.3 TFLOP/s).  Also reports the shader clock seen inside the kernel (s_memtime ticks per
// 100 MHz s_memrealtime tick).    hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, unsigned long long* clk, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  const int cus = p.multiProcessorCount;
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = cus * wgs_per_cu, iters = 200000;
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&clk, (size_t)grid * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, clk, 1000, 1.f, 2.f);  // warm-up
    hipDeviceSynchronize();
    float best = 1e30f, last = 0.f;
    for (int rep = 0; rep < 5; ++rep) {  // ~0.5 s each: long enough for the clock to settle
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.f, 2.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&last, e0, e1);
      if (last < best) best = last;
    }
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), clk, (size_t)grid * 16, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    ghz /= grid;
    const double flops = (double)grid * 4 /*waves*/ * iters * 32.0 * 4096.0;
    printf("{\"cus\": %d, \"waves_per_simd\": %d, \"tflops_best\": %.1f, \"tflops_last\": %.1f, \"shader_ghz_in_kernel\": %.3f, "
           "\"ms\": %.1f}\n", cus, wgs_per_cu, flops / (best * 1e-3) / 1e12, flops / (last * 1e-3) / 1e12, ghz, last);
    hipFree(out);
    hipFree(clk);
  }
  return 0;
}
