// Feasibility probe (round 5): can the fp32 VALU add FLOPs NEXT TO a saturated fp32 matrix pipe?  MI355X_MICROARCH.md
// states both at 64 FLOP/clk/SIMD and that the two pipes are separate.  A register-only loop issues, per
// v_mfma_f32_32x32x2_f32 (64 cycles of matrix pipe), V independent v_pk_fma_f32 (4 cycles of VALU each: 256 FLOP) from the
// same wave; V = 0 is the matrix-only baseline, M = 0 the VALU-only one.  Reported: total TFLOP/s, the split, and the
// shader clock inside the kernel -- i.e. whether the sum survives the power limit.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_hybrid.hip -o tools/micro/bin/mfma_valu_hybrid
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// KIND: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_add_f32, 3 v_mov_b32, 4 v_exp_f32, 5 v_pk_add_f32, 6 v_max_f32, 7 v_cndmask (select),
//       8 v_and_b32 (integer)
template <int KIND>
__device__ __forceinline__ void valu_op(v2f& c, v2f x, v2f y) {
  if constexpr (KIND == 0) c = __builtin_elementwise_fma(x, y, c);
  else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c.x) : "v"(x.x), "v"(y.x));
  else if constexpr (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(c.x) : "v"(x.x));
  else if constexpr (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(c.x) : "v"(x.x));
  else if constexpr (KIND == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(c.x) : "v"(x.x));
  else if constexpr (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(c) : "v"(x));
  else if constexpr (KIND == 6) asm volatile("v_max_f32 %0, %1, %0" : "+v"(c.x) : "v"(x.x));
  else if constexpr (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(c.x) : "v"(x.x));
  else if constexpr (KIND == 8) asm volatile("v_and_b32 %0, %1, %0" : "+v"(c.x) : "v"(x.x));
  else if constexpr (KIND == 9) asm volatile("ds_read_b32 %0, %1" : "=v"(c.x) : "v"(__builtin_amdgcn_mbcnt_lo(~0u, 0u) * 4u) : "memory");
  else if constexpr (KIND == 10) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t;
    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(__builtin_amdgcn_mbcnt_lo(~0u, 0u) * 16u) : "memory");
    c.x = t.x;
  } else asm volatile("s_nop 0");
}

template <int V, bool MFMA, int KIND = 0>
__global__ __launch_bounds__(256, 2) void hybrid_loop(float* out, unsigned long long* clk, int iters, float a0, float b0) {
  __shared__ float lds_dummy[4096];
  if (threadIdx.x < 64) lds_dummy[threadIdx.x] = a0;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  v2f c[16];
  for (int i = 0; i < 16; ++i) c[i] = v2f{0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  v2f x = v2f{a * 1e-3f, b * 1e-3f}, y = v2f{b * 1e-3f, -a * 1e-3f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MFMA) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) valu_op<KIND>(c[(u * V + v) & 15], x, y);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 16; ++i) s += c[i].x + c[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V, bool MFMA, int KIND = 0>
static void run(int cus) {
  const int grid = cus * 2, iters = 40000;
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipMalloc(&clk, (size_t)grid * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((hybrid_loop<V, MFMA, KIND>), dim3(grid), dim3(256), 0, 0, out, clk, 1000, 1.f, 2.f);
  hipDeviceSynchronize();
  float best = 1e30f, last = 0.f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((hybrid_loop<V, MFMA, KIND>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.f, 2.f);
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
  const double waves = (double)grid * 4, per_iter_mfma = MFMA ? 4 * 4096.0 : 0.0, per_iter_valu = 4.0 * V * 256.0;
  const double tf_m = waves * iters * per_iter_mfma / (last * 1e-3) / 1e12, tf_v = waves * iters * per_iter_valu / (last * 1e-3) / 1e12;
  static const char* names[] = {"v_pk_fma_f32", "v_fma_f32", "v_add_f32", "v_mov_b32", "v_exp_f32", "v_pk_add_f32", "v_max_f32",
                                "v_cndmask_b32", "v_and_b32", "ds_read_b32", "ds_read_b128", "s_nop"};
  // cycles per loop iteration and SIMD (two waves per SIMD): what the 4 MFMAs alone would take is 2 x 4 x 64 = 512
  const double cyc = last * 1e-3 * ghz * 1e9 / iters;
  printf("mfma %d + %2d x %-13s per mfma: matrix %6.1f TFLOP/s (+ vector %6.1f if FMA), %.0f cycles per 4-MFMA iteration and SIMD "
         "(2 waves; matrix alone 512), clock %.3f GHz\n", (int)MFMA, V, names[KIND], tf_m, KIND == 0 ? tf_v : 0.0, cyc, ghz);
  hipFree(out);
  hipFree(clk);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  const int cus = p.multiProcessorCount;
  run<0, true>(cus);
  run<2, true>(cus);
  run<4, true>(cus);
  run<8, true>(cus);
  run<12, true>(cus);
  run<16, true>(cus);
  run<16, false>(cus);
  // which VALU instructions cost the fp32 matrix pipe its cycles?  8 per MFMA slot, other instruction kinds
  run<8, true, 1>(cus);
  run<8, true, 2>(cus);
  run<8, true, 5>(cus);
  run<8, true, 6>(cus);
  run<8, true, 3>(cus);
  run<8, true, 7>(cus);
  run<8, true, 8>(cus);
  run<8, true, 4>(cus);
  run<8, true, 9>(cus);   // LDS instructions (results waited for at the end of the loop body by the compiler's s_waitcnt)
  run<8, true, 10>(cus);
  run<8, true, 11>(cus);  // s_nop: pure issue slots
  return 0;
}
