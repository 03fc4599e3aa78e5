// Which fp32 MFMA shape sustains the higher rate on this part under full load?  Register-only loops (no memory traffic)
// of v_mfma_f32_32x32x2_f32 (4096 FLOP, 16 passes) and v_mfma_f32_16x16x4_f32 (2048 FLOP, 8 passes) with random-ish
// operands on every SIMD, ~1 s each so that the clock settles; prints TFLOP/s and the shader clock seen in the kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, unsigned long long* clk, int iters, float a0, float b0) {
  const float a = __sinf(a0 * (threadIdx.x + 1) * 0.37f), b = __cosf(b0 * (threadIdx.x + 3) * 0.21f) * 1e-3f;
  float s = 0.f;
  unsigned long long t0, r0, t1, r1;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    t0 = __builtin_readcyclecounter(); r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a * 1e-3f, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
      }
    }
    t1 = __builtin_readcyclecounter(); r1 = wall_clock64();
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t)
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    t0 = __builtin_readcyclecounter(); r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // same FLOPs per iteration: 64 x 2048 = 32 x 4096
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a * 1e-3f, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a * 1e-3f, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[7], 0, 0, 0);
      }
    }
    t1 = __builtin_readcyclecounter(); r1 = wall_clock64();
    for (int t = 0; t < 8; ++t)
      for (int r = 0; r < 4; ++r) s += acc[t][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(int cus, int wgs_per_cu) {
  const int grid = cus * wgs_per_cu, iters = 150000;
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipMalloc(&clk, (size_t)grid * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(256), 0, 0, out, clk, 1000, 1.f, 2.f);
  hipDeviceSynchronize();
  float last = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&last, e0, e1);
  }
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), clk, (size_t)grid * 16, hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
  ghz /= grid;
  const double flops = (double)grid * 4 * iters * 32.0 * 4096.0;
  printf("{\"shape\": \"%s\", \"waves_per_simd\": %d, \"tflops\": %.1f, \"shader_ghz_in_kernel\": %.3f, \"ms\": %.1f}\n",
         SHAPE == 32 ? "32x32x2" : "16x16x4", wgs_per_cu, flops / (last * 1e-3) / 1e12, ghz, last);
  hipFree(out);
  hipFree(clk);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  for (int rep = 0; rep < 2; ++rep)
    for (int w = 1; w <= 2; ++w) {
      run<32>(p.multiProcessorCount, w);
      run<16>(p.multiProcessorCount, w);
    }
  return 0;
}
