// Where does an LDS-fed fp32-MFMA loop lose issue slots?  Variants of the conv / GEMM inner loop shape
// (per step: 16 ds_read_b128 feeding 64 v_mfma_f32_32x32x2_f32 on 4 accumulators, 4 waves per workgroup):
//   mode 0: MFMAs only (registers)          mode 1: + LDS fragment reads
//   mode 2: + one workgroup barrier / step  mode 3: + global->register->LDS staging of a weight tile / step
// Prints TFLOP/s for 1, 2 and 3 workgroups per CU.   hipcc --offload-arch=gfx950 -O3 mfma_feed.hip -o mfma_feed
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 36

template <int MODE>
__global__ __launch_bounds__(256, 2) void feed_loop(float* out, const float* __restrict__ src, int steps, int random_data) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_s = smem;              // [324][LD] "input halo tile"
  float* w_s = smem + 324 * LD;   // [2][64][LD] "weights"
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 324 * LD + 2 * 64 * LD; i += 256) {
    // random_data: full-entropy mantissas and mixed signs, like real activations / weights (switching power)
    unsigned hsh = (unsigned)(i + 977 * blockIdx.x) * 2654435761u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    smem[i] = random_data ? ((float)(hsh & 0xffffff) / 8388608.f - 1.f) : 1e-3f * (float)(i % 97);
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  int a_off[2], b_off[2];
  for (int m = 0; m < 2; ++m) a_off[m] = ((4 * wave + 2 * m + (l31 >> 4)) * 18 + (l31 & 15)) * LD + 4 * h;
  for (int n = 0; n < 2; ++n) b_off[n] = (n * 32 + l31) * LD + 4 * h;
  const int st_co = tid >> 3, st_c4 = (tid & 7) * 4;
  const float* wsrc = src + (size_t)(blockIdx.x % 64) * 64 * 32;
  float4 wreg0 = make_float4(0, 0, 0, 0), wreg1 = wreg0;
  float4 regA[2] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(.5f, .25f, .125f, 1.f)};
  for (int step = 0; step < steps; ++step) {
    const int tap = step % 9;
    if (MODE == 5) {
      // write-after-barrier order: the registers loaded during the previous step go to LDS first, then the load of
      // the step after next is issued; the step ends on the barrier with nothing else pending
      float* dst = w_s + ((step + 1) & 1) * 64 * LD + st_co * LD + st_c4;
      *reinterpret_cast<float4*>(dst) = wreg0;
      *reinterpret_cast<float4*>(dst + 32 * LD) = wreg1;
    }
    if (MODE >= 3) {
      wreg0 = *reinterpret_cast<const float4*>(wsrc + (size_t)(step & 7) * 64 * 32 * 64 + st_co * 32 + st_c4);
      wreg1 = *reinterpret_cast<const float4*>(wsrc + (size_t)(step & 7) * 64 * 32 * 64 + (st_co + 32) * 32 + st_c4);
    }
    const float* ap = a_s + ((tap / 3) * 18 + tap % 3) * LD;
    const float* bp = w_s + (step & 1) * 64 * LD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 af[2], bf[2];
      if (MODE >= 1) {
        for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const float4*>(ap + a_off[m] + 8 * g);
        for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const float4*>(bp + b_off[n] + 8 * g);
      } else {
        af[0] = regA[0]; af[1] = regA[1]; bf[0] = regA[1]; bf[1] = regA[0];
      }
#define M4(c_)                                                                         \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0].c_, bf[0].c_, acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0].c_, bf[1].c_, acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1].c_, bf[0].c_, acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1].c_, bf[1].c_, acc[1][1], 0, 0, 0);
      M4(x) M4(y) M4(z) M4(w)
    }
    if (MODE == 3) {
      float* dst = w_s + ((step + 1) & 1) * 64 * LD + st_co * LD + st_c4;
      *reinterpret_cast<float4*>(dst) = wreg0;
      *reinterpret_cast<float4*>(dst + 32 * LD) = wreg1;
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

// mode 4: like mode 3 but the weight tile is TRIPLE-buffered, so the fragments of the next step's first k-group can
// be read before the step's barrier (they were made visible one barrier earlier) and fragment reads run one group
// ahead of the MFMAs: nothing waits on LDS right after a barrier.
__global__ __launch_bounds__(256, 2) void feed_loop_pipe(float* out, const float* __restrict__ src, int steps, int random_data) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_s = smem;
  float* w_s = smem + 324 * LD;   // [3][64][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 324 * LD + 3 * 64 * LD; i += 256) {
    unsigned hsh = (unsigned)(i + 977 * blockIdx.x) * 2654435761u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    smem[i] = random_data ? ((float)(hsh & 0xffffff) / 8388608.f - 1.f) : 1e-3f * (float)(i % 97);
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  int a_off[2], b_off[2];
  for (int m = 0; m < 2; ++m) a_off[m] = ((4 * wave + 2 * m + (l31 >> 4)) * 18 + (l31 & 15)) * LD + 4 * h;
  for (int n = 0; n < 2; ++n) b_off[n] = (n * 32 + l31) * LD + 4 * h;
  const int st_co = tid >> 3, st_c4 = (tid & 7) * 4;
  const float* wsrc = src + (size_t)(blockIdx.x % 64) * 64 * 32;
  float4 wreg0, wreg1;
  float4 af0, af1, bf0, bf1, an0, an1, bn0, bn1;
#define RD(dstA0, dstA1, dstB0, dstB1, ap_, bp_, g_)                         \
  dstA0 = *reinterpret_cast<const float4*>((ap_) + a_off[0] + 8 * (g_));     \
  dstA1 = *reinterpret_cast<const float4*>((ap_) + a_off[1] + 8 * (g_));     \
  dstB0 = *reinterpret_cast<const float4*>((bp_) + b_off[0] + 8 * (g_));     \
  dstB1 = *reinterpret_cast<const float4*>((bp_) + b_off[1] + 8 * (g_));
#define MM(c_)                                                                          \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.c_, bf0.c_, acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.c_, bf1.c_, acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.c_, bf0.c_, acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.c_, bf1.c_, acc[1][1], 0, 0, 0);
  RD(af0, af1, bf0, bf1, a_s, w_s, 0)
  for (int step = 0; step < steps; ++step) {
    const int tap = step % 9, ntap = (step + 1) % 9;
    wreg0 = *reinterpret_cast<const float4*>(wsrc + (size_t)(step & 7) * 64 * 32 * 64 + st_co * 32 + st_c4);
    wreg1 = *reinterpret_cast<const float4*>(wsrc + (size_t)(step & 7) * 64 * 32 * 64 + (st_co + 32) * 32 + st_c4);
    const float* ap = a_s + ((tap / 3) * 18 + tap % 3) * LD;
    const float* bp = w_s + (step % 3) * 64 * LD;
    const float* nap = a_s + ((ntap / 3) * 18 + ntap % 3) * LD;
    const float* nbp = w_s + ((step + 1) % 3) * 64 * LD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) { RD(an0, an1, bn0, bn1, ap, bp, g + 1) } else { RD(an0, an1, bn0, bn1, nap, nbp, 0) }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (hipcc sinks them to their first use)
      MM(x) MM(y) MM(z) MM(w)
      __builtin_amdgcn_sched_barrier(0);
      af0 = an0; af1 = an1; bf0 = bn0; bf1 = bn1;
    }
    float* dst = w_s + ((step + 2) % 3) * 64 * LD + st_co * LD + st_c4;
    *reinterpret_cast<float4*>(dst) = wreg0;
    *reinterpret_cast<float4*>(dst + 32 * LD) = wreg1;
    __syncthreads();
  }
  float sacc = 0.f;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 16; ++r) sacc += acc[m][n][r];
  out[(size_t)blockIdx.x * 256 + tid] = sacc + af0.x;
}

static void run_pipe(int cus, float* out, const float* src, int random_data) {
  const size_t lds_base = (324 * LD + 3 * 64 * LD) * sizeof(float);  // 74 KB -> 2 workgroups / CU
  hipFuncSetAttribute((const void*)feed_loop_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int per_cu = 1; per_cu <= 2; ++per_cu) {
    const size_t lds = per_cu == 1 ? 120 * 1024 : lds_base;
    const int grid = cus * per_cu, steps = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(feed_loop_pipe, dim3(grid), dim3(256), lds, 0, out, src, 100, random_data);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      hipEventRecord(e0);
      hipLaunchKernelGGL(feed_loop_pipe, dim3(grid), dim3(256), lds, 0, out, src, steps, random_data);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * steps * 64.0 * 4096.0;
    printf("{\"mode\": 4, \"random_data\": %d, \"wg_per_cu\": %d, \"tflops\": %.1f, \"frac_of_157.3\": %.3f}\n", random_data,
           per_cu, flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12 / 157.3);
  }
}

template <int MODE>
static void run(int cus, float* out, const float* src, int random_data) {
  const size_t lds_base = (324 * LD + 2 * 64 * LD) * sizeof(float);  // 65 KB -> 2 workgroups / CU
  hipFuncSetAttribute((const void*)feed_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int per_cu = 1; per_cu <= 2; ++per_cu) {
    // pad the dynamic LDS so that exactly per_cu workgroups fit on a CU
    const size_t lds = per_cu == 1 ? 120 * 1024 : lds_base;
    const int grid = cus * per_cu, steps = 40000;  // ~50 ms per launch: long enough for the clock to settle
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(feed_loop<MODE>, dim3(grid), dim3(256), lds, 0, out, src, 100, random_data);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      hipEventRecord(e0);
      hipLaunchKernelGGL(feed_loop<MODE>, dim3(grid), dim3(256), lds, 0, out, src, steps, random_data);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * steps * 64.0 * 4096.0;
    printf("{\"mode\": %d, \"random_data\": %d, \"wg_per_cu\": %d, \"tflops\": %.1f, \"frac_of_157.3\": %.3f}\n", MODE, random_data, per_cu,
           flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12 / 157.3);
  }
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  float *out, *src;
  hipMalloc(&out, (size_t)p.multiProcessorCount * 2 * 256 * 4);
  hipMalloc(&src, (size_t)64 * 64 * 32 * 64 * 8 * 4 / 8 + (1 << 20));
  hipMemset(src, 0, (size_t)64 * 64 * 32 * 64 * 8 * 4 / 8 + (1 << 20));
  run<0>(p.multiProcessorCount, out, src, 1);
  run<1>(p.multiProcessorCount, out, src, 1);
  run<2>(p.multiProcessorCount, out, src, 1);
  run<3>(p.multiProcessorCount, out, src, 1);
  run_pipe(p.multiProcessorCount, out, src, 1);
  run<5>(p.multiProcessorCount, out, src, 1);
  return 0;
}
