// What does s_memtime tick at, and what is the shader clock under full fp32-MFMA load?
// Each wave issues N back-to-back v_mfma_f32_32x32x2_f32 on 4 accumulators (64 shader cycles each when the pipe is its
// own) and records s_memtime and s_memrealtime (constant 100 MHz) around them.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_calib.hip -o /tmp/clock_calib && /tmp/clock_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void calib(unsigned long long* out, int n, float seed) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = seed * r; a1[r] = seed + r; a2[r] = seed - r; a3[r] = seed * 0.5f * r; }
  const float x = seed * (threadIdx.x % 7 + 1) * 0.37f, y = seed * (threadIdx.x % 5 + 1) * 0.21f;
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 12345.f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = out + 1 + 2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    o[0] = t1 - t0;
    o[1] = w1 - w0;
  }
}

int main() {
  const int n = 20000;  // x4 MFMAs per wave
  for (int blocks : {1, 256, 512}) {
    unsigned long long* d;
    hipMalloc(&d, (1 + 2 * 4 * 512) * sizeof(unsigned long long));
    hipMemset(d, 0, (1 + 2 * 4 * 512) * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(calib, dim3(blocks), dim3(256), 0, 0, d, n, 0.001f);  // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib, dim3(blocks), dim3(256), 0, 0, d, n, 0.37f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1 + 2 * 4 * blocks);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double tick = 0, wall = 0;
    for (int i = 0; i < 4 * blocks; ++i) { tick += h[1 + 2 * i]; wall += h[2 + 2 * i]; }
    tick /= 4 * blocks; wall /= 4 * blocks;
    const double mf = 4.0 * n;
    printf("%3d workgroups (%s): s_memtime ticks per MFMA %.2f; wave time %.1f us (100 MHz counter) = %.2f ns per MFMA -> "
           "%.3f GHz if an MFMA takes 64 (x waves sharing the SIMD) shader cycles; s_memtime rate %.3f GHz; kernel %.1f us; "
           "%.1f TFLOP/s\n",
           blocks, blocks == 1 ? "one wave per SIMD of one CU" : blocks == 256 ? "one wave per SIMD, every CU" : "two waves per SIMD",
           tick / mf, wall / 100.0, wall * 10.0 / mf, 64.0 * (blocks == 512 ? 2 : 1) / (wall * 10.0 / mf), tick / (wall * 10.0),
           ms * 1e3, blocks * 4 * mf * 4096.0 / (ms * 1e-3) / 1e12);
    hipFree(d);
  }
  return 0;
}
