// Round-5 decision experiment (DESIGN.md 8.1 option (a)): bidirectional cross attention with ONE
// evaluation of sim = qk0 . qk1^T per tile (reference gluefactory/models/matchers/lightglue.py:207-217), timed against
// today's launch (attention_kernel<2,4>, both directions as two problems, sim evaluated twice).
//
// One workgroup = 4 waves = 128 rows of image 0 for one (pair, head); wave w owns 32 rows.  Sweep over the columns
// (key points of image 1) in tiles of 32:
//   S^T[c][m] = Q1[tile] . Q0[rows]^T                         32 MFMAs   (lane <-> row m, registers <-> column c)
//   row direction:    online softmax over c (lane-local), O0^T += V1^T . P   32 MFMAs        -- as attention.hip
//   column direction: the raw tile is transposed through LDS (lane <-> column c, registers <-> row m), softmax
//                     statistics over the wave's 32 rows are lane-local, O1p^T[d][c] = V0^T[d][m] . P[m][c]   32 MFMAs
//                     (A operand = the wave's own 32 rows of V0, register-resident for the whole sweep);
//                     the four waves' partials (64 values + max + sum per column) are merged through LDS and written as
//                     ONE partial record per (128-row block, column): 8 records per column at M = 1024.
// A second kernel merges the 8 records per column (the key-split merge of attention.hip).  96 MFMAs per 32 x 32 tile
// instead of 128; the price is the transpose, two workgroup barriers per column tile, 277 MB of partial records written
// and read per launch at 32 pairs x 1024^2, and the merge launch.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/cross_shared_sim.hip \
//         -Lglue-factory-colon_amd -lgfc_amd -Wl,-rpath,'$ORIGIN/../../../glue-factory-colon_amd' -o tools/micro/bin/cross_shared_sim
//   tools/micro/bin/cross_shared_sim [pairs = 32] [points = 1024]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <cmath>
#include <vector>

#include "../../glue-factory-colon_amd/csrc/common.h"

#define CK 32        // columns per tile
#define CD 64        // head dim
#define CKLD (CD + 4)
#define CPT 68       // floats per column record in the combine area: 64 values, max, sum, pad (b128-aligned)
#define CTP 36       // pitch of the transpose scratch [32 rows][36]

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// problems[p] = {row0 of image 0, M, row0 of image 1, N}; part: [p][head][row block][N][66]
__global__ __launch_bounds__(256, 2) void cross_shared_kernel(const float* __restrict__ QK, int ldq, const float* __restrict__ V,
                                                              int ldv, float* __restrict__ O, int ldo,
                                                              const int4* __restrict__ problems, float scale_log2e,
                                                              float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 K tiles, 2 V tiles, 4 combine regions: 68.6 KB
  float* Ks_all = smem;
  float* Vs_all = smem + 2 * CK * CKLD;
  float* comb = smem + 2 * CK * CKLD + 2 * CK * CD;

  const int4 pb = problems[blockIdx.z];
  const int r0 = pb.x, M = pb.y, r1 = pb.z, N = pb.w;
  const int head = blockIdx.y, rb = blockIdx.x, nrb = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int m_row = rb * 128 + wave * 32 + l31;  // this lane's row of image 0 (query of the row direction)

  // Q0 fragments (B operand of S^T): lane (m, h) keeps Q0[m][8g + 4h + s]
  float4 qf[8];
  {
    const float* qp = QK + (size_t)(r0 + m_row) * ldq + head * CD + 4 * h;
#pragma unroll
    for (int g = 0; g < 8; ++g) qf[g] = *reinterpret_cast<const float4*>(qp + 8 * g);
  }
  // V0 fragments (A operand of the column direction): v0f[2r + t] = V0[row acc_row(r, h) of this wave][32 t + l31]
  float v0f[32];
  {
    const float* vp = V + (size_t)(r0 + rb * 128 + wave * 32) * ldv + head * CD + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      v0f[2 * r] = vp[(size_t)row * ldv];
      v0f[2 * r + 1] = vp[(size_t)row * ldv + 32];
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  // staging of the column tile (Q1 rows as K, V1 rows as V): 32 x 64 floats each = 512 float4 -> 2 per thread
  const int st_key = tid >> 4, st_c4 = (tid & 15) * 4;  // keys st_key, st_key + 16
  const float* kbase = QK + (size_t)r1 * ldq + head * CD + st_c4;
  const float* vbase = V + (size_t)r1 * ldv + head * CD + st_c4;
  float4 kr0, kr1, vr0, vr1;
#define CS_LOAD(j_)                                                                          \
  do {                                                                                       \
    const size_t ra_ = (size_t)((j_) * CK + st_key), rb_ = ra_ + 16;                         \
    kr0 = *reinterpret_cast<const float4*>(kbase + ra_ * ldq);                               \
    kr1 = *reinterpret_cast<const float4*>(kbase + rb_ * ldq);                               \
    vr0 = *reinterpret_cast<const float4*>(vbase + ra_ * ldv);                               \
    vr1 = *reinterpret_cast<const float4*>(vbase + rb_ * ldv);                               \
  } while (0)
#define CS_STORE(buf_)                                                                       \
  do {                                                                                       \
    float* kd_ = Ks_all + (buf_) * CK * CKLD + st_c4;                                        \
    float* vd_ = Vs_all + (buf_) * CK * CD + st_c4;                                          \
    *reinterpret_cast<float4*>(kd_ + st_key * CKLD) = kr0;                                   \
    *reinterpret_cast<float4*>(kd_ + (st_key + 16) * CKLD) = kr1;                            \
    *reinterpret_cast<float4*>(vd_ + st_key * CD) = vr0;                                     \
    *reinterpret_cast<float4*>(vd_ + (st_key + 16) * CD) = vr1;                              \
  } while (0)

  const int ntiles = N / CK;
  CS_LOAD(0);
  CS_STORE(0);
  __syncthreads();
  float* mine = comb + wave * CK * CPT;  // this wave's combine region; its head doubles as the transpose scratch
  // merge role of this thread: column c_m, eight values at d = 8 dq
  const int c_m = tid >> 3, dq = tid & 7;
  float* prec = part + ((((size_t)blockIdx.z * gridDim.y + head) * nrb + rb) * N) * 66;

  for (int j = 0; j < ntiles; ++j) {
    const bool has_next = j + 1 < ntiles;
    if (has_next) CS_LOAD(j + 1);
    const int buf = j & 1;
    const float* Ks = Ks_all + buf * CK * CKLD;
    const float* Vs = Vs_all + buf * CK * CD;

    // ---- S^T = Q1[tile] . Q0^T ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const float* kp = Ks + l31 * CKLD + 4 * h;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float4 kf = *reinterpret_cast<const float4*>(kp + 8 * g);
        s = mfma32(kf.x, qf[g].x, s);
        s = mfma32(kf.y, qf[g].y, s);
        s = mfma32(kf.z, qf[g].z, s);
        s = mfma32(kf.w, qf[g].w, s);
      }
    }
    // ---- raw tile -> transpose scratch T[m][c] (wave-private): c = (r & 3) + 8 (r >> 2) + 4 h, four consecutive c per float4 ----
    {
      float* tp = mine + l31 * CTP + 4 * h;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(tp + 8 * g4) = make_float4(s[4 * g4], s[4 * g4 + 1], s[4 * g4 + 2], s[4 * g4 + 3]);
    }
    // ---- row direction (as attention.hip, one query tile per wave) ----
    {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f((s[r] - m_new) * scale_log2e);
        rs += s[r];
      }
      rs += __shfl_xor(rs, 32);
      l_run = l_run * alpha + rs;
      if (!__all(m_new == m_run)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
      m_run = m_new;
      const float* vp = Vs + (4 * h) * CD + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int krow = (r & 3) + 8 * (r >> 2);
        o[0] = mfma32(vp[krow * CD], s[r], o[0]);
        o[1] = mfma32(vp[krow * CD + 32], s[r], o[1]);
      }
    }
    // ---- column direction: X[m][c] with the column on the lane ----
    f32x16 x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const float* tp = mine + (4 * h) * CTP + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = tp[((r & 3) + 8 * (r >> 2)) * CTP];
    }
    float cmx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) cmx = fmaxf(cmx, x[r]);
    cmx = fmaxf(cmx, __shfl_xor(cmx, 32));
    float cl = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      x[r] = __builtin_amdgcn_exp2f((x[r] - cmx) * scale_log2e);
      cl += x[r];
    }
    cl += __shfl_xor(cl, 32);
    f32x16 o1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o1[0][r] = 0.f; o1[1][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o1[0] = mfma32(v0f[2 * r], x[r], o1[0]);
      o1[1] = mfma32(v0f[2 * r + 1], x[r], o1[1]);
    }
    // partial of this wave -> its combine region [c][d | max | sum] (the scratch reads above are complete: same wave, in order)
    __builtin_amdgcn_wave_barrier();
    {
      float* cp = mine + l31 * CPT + 4 * h;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(cp + 32 * t + 8 * g4) =
              make_float4(o1[t][4 * g4], o1[t][4 * g4 + 1], o1[t][4 * g4 + 2], o1[t][4 * g4 + 3]);
      if (h == 0) { mine[l31 * CPT + 64] = cmx; mine[l31 * CPT + 65] = cl; }
    }
    __syncthreads();
    // ---- merge the four waves' partials of this column tile, one record per column to HBM ----
    {
      float mw[4], lw[4];
      float mmax = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        mw[w] = comb[(w * CK + c_m) * CPT + 64];
        lw[w] = comb[(w * CK + c_m) * CPT + 65];
        mmax = fmaxf(mmax, mw[w]);
      }
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
      float lsum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float f = __builtin_amdgcn_exp2f((mw[w] - mmax) * scale_log2e);
        const float4 u0 = *reinterpret_cast<const float4*>(comb + (w * CK + c_m) * CPT + 8 * dq);
        const float4 u1 = *reinterpret_cast<const float4*>(comb + (w * CK + c_m) * CPT + 8 * dq + 4);
        a0.x = fmaf(f, u0.x, a0.x); a0.y = fmaf(f, u0.y, a0.y); a0.z = fmaf(f, u0.z, a0.z); a0.w = fmaf(f, u0.w, a0.w);
        a1.x = fmaf(f, u1.x, a1.x); a1.y = fmaf(f, u1.y, a1.y); a1.z = fmaf(f, u1.z, a1.z); a1.w = fmaf(f, u1.w, a1.w);
        lsum = fmaf(f, lw[w], lsum);
      }
      float* dst = prec + (size_t)(j * CK + c_m) * 66 + 8 * dq;  // 8-byte aligned (66 floats per record)
      reinterpret_cast<float2*>(dst)[0] = make_float2(a0.x, a0.y);
      reinterpret_cast<float2*>(dst)[1] = make_float2(a0.z, a0.w);
      reinterpret_cast<float2*>(dst)[2] = make_float2(a1.x, a1.y);
      reinterpret_cast<float2*>(dst)[3] = make_float2(a1.z, a1.w);
      if (dq == 0) reinterpret_cast<float2*>(prec + (size_t)(j * CK + c_m) * 66 + 64)[0] = make_float2(mmax, lsum);
    }
    if (has_next) CS_STORE(buf ^ 1);
    __syncthreads();
  }
  // ---- row direction: normalise and store ----
  {
    const float inv = 1.f / l_run;
    float* op = O + (size_t)(r0 + m_row) * ldo + head * CD + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + db * 32 + 8 * g) =
            make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
  }
  (void)M;
}

// merge the row-block partials of the column direction: one wave per (column, head); lane = channel
__global__ __launch_bounds__(256) void cross_merge_kernel(const float* __restrict__ part, float* __restrict__ O, int ldo,
                                                          const int4* __restrict__ problems, int nrb, float scale_log2e) {
  const int4 pb = problems[blockIdx.z];
  const int N = pb.w;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), head = blockIdx.y, lane = threadIdx.x & 63;
  if (n >= N) return;
  const float* pp = part + (((size_t)blockIdx.z * gridDim.y + head) * nrb * N + n) * 66;
  const size_t stride = (size_t)N * 66;
  float m = -INFINITY;
  for (int s = 0; s < nrb; ++s) m = fmaxf(m, pp[s * stride + 64]);
  float acc = 0.f, l = 0.f;
  for (int s = 0; s < nrb; ++s) {
    const float w = __builtin_amdgcn_exp2f((pp[s * stride + 64] - m) * scale_log2e);
    acc += w * pp[s * stride + lane];
    l += w * pp[s * stride + 65];
  }
  O[(size_t)(pb.z + n) * ldo + head * CD + lane] = acc / l;
}

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 32, K = argc > 2 ? atoi(argv[2]) : 1024, heads = 4;
  if (K % 128) { printf("points must be a multiple of 128\n"); return 1; }
  const int R = pairs * 2 * K, ld = 256;
  std::vector<float> hq((size_t)R * ld), hv((size_t)R * ld);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)); };
  auto gauss = [&]() { float u = rnd() + 1e-12f, v = rnd(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); };
  for (auto& x : hq) x = 1.6f * gauss();  // logits scale * q.k: standard deviation ~2.6 (peaky rows, as a trained matcher's)
  for (auto& x : hv) x = gauss();
  float *dq, *dv, *o_ref, *o_new, *part;
  int4* dprob;
  int32_t* dprob_lib;
  HIP_OK(hipMalloc(&dq, hq.size() * 4)); HIP_OK(hipMalloc(&dv, hv.size() * 4));
  HIP_OK(hipMalloc(&o_ref, hq.size() * 4)); HIP_OK(hipMalloc(&o_new, hq.size() * 4));
  HIP_OK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
  const int nrb = K / 128;
  const size_t part_bytes = (size_t)pairs * heads * nrb * K * 66 * 4;
  HIP_OK(hipMalloc(&part, part_bytes));
  std::vector<int4> prob(pairs);
  std::vector<int32_t> prob_lib(8 * pairs);
  for (int p = 0; p < pairs; ++p) {
    const int r0 = p * 2 * K, r1 = r0 + K;
    prob[p] = make_int4(r0, K, r1, K);
    const int32_t a[8] = {r0, K, r1, K, r1, K, r0, K};
    for (int i = 0; i < 8; ++i) prob_lib[8 * p + i] = a[i];
  }
  HIP_OK(hipMalloc(&dprob, pairs * sizeof(int4))); HIP_OK(hipMalloc(&dprob_lib, prob_lib.size() * 4));
  HIP_OK(hipMemcpy(dprob, prob.data(), pairs * sizeof(int4), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dprob_lib, prob_lib.data(), prob_lib.size() * 4, hipMemcpyHostToDevice));
  const float scale = 0.125f, sl2 = scale * 1.4426950408889634f;
  hipStream_t s0;
  HIP_OK(hipStreamCreate(&s0));

  auto run_lib = [&]() {
    int rc = gfc_attention(dq, ld, dq, ld, dv, ld, o_ref, ld, dprob_lib, 2 * pairs, K, heads, scale, nullptr, 0, s0);
    if (rc) { printf("gfc_attention failed %d\n", rc); exit(1); }
  };
  const size_t lds = (size_t)(2 * CK * CKLD + 2 * CK * CD + 4 * CK * CPT) * sizeof(float);
  HIP_OK(hipFuncSetAttribute((const void*)cross_shared_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto run_new = [&](bool merge) {
    hipLaunchKernelGGL(cross_shared_kernel, dim3(nrb, heads, pairs), dim3(256), lds, s0, dq, ld, dv, ld, o_new, ld, dprob, sl2, part);
    if (merge)
      hipLaunchKernelGGL(cross_merge_kernel, dim3(K / 4, heads, pairs), dim3(256), 0, s0, part, o_new, ld, dprob, nrb, sl2);
  };
  run_lib();
  run_new(true);
  HIP_OK(hipStreamSynchronize(s0));
  HIP_OK(hipGetLastError());
  std::vector<float> a((size_t)R * ld), b((size_t)R * ld);
  HIP_OK(hipMemcpy(a.data(), o_ref, a.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(b.data(), o_new, b.size() * 4, hipMemcpyDeviceToHost));
  double e0 = 0, e1 = 0, mag = 0;
  for (int p = 0; p < pairs; ++p)
    for (int r = 0; r < 2 * K; ++r)
      for (int c = 0; c < ld; ++c) {
        const size_t i = ((size_t)p * 2 * K + r) * ld + c;
        const double d = fabs((double)a[i] - (double)b[i]);
        if (r < K) e0 = fmax(e0, d); else e1 = fmax(e1, d);
        mag = fmax(mag, fabs((double)a[i]));
      }
  printf("max |difference| against gfc_attention (two problems per pair): row direction %.3g, column direction %.3g (|O| max %.3g)\n", e0, e1, mag);

  hipEvent_t ev0, ev1;
  HIP_OK(hipEventCreate(&ev0)); HIP_OK(hipEventCreate(&ev1));
  auto time_us = [&](auto fn, int reps) {
    for (int i = 0; i < 3; ++i) fn();
    HIP_OK(hipStreamSynchronize(s0));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      HIP_OK(hipEventRecord(ev0, s0));
      for (int i = 0; i < reps; ++i) fn();
      HIP_OK(hipEventRecord(ev1, s0));
      HIP_OK(hipEventSynchronize(ev1));
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, ev0, ev1));
      best = fminf(best, ms * 1e3f / reps);
    }
    return best;
  };
  const float t_lib = time_us(run_lib, 20);
  const float t_main = time_us([&]() { run_new(false); }, 20);
  const float t_both = time_us([&]() { run_new(true); }, 20);
  const double prod = 2.0 * K * K * 64 * 4;
  printf("%d pairs x %d^2, 4 heads:\n", pairs, K);
  printf("  today  (attention_kernel, 2 problems per pair, sim twice): %8.1f us  (%.1f TFLOP/s algorithmic, 3 products)\n", t_lib,
         pairs * 3 * prod / t_lib * 1e-6);
  printf("  shared (one sim, row direction online + column partials):  %8.1f us main kernel, %8.1f us with the merge "
         "(%.1f TFLOP/s algorithmic); partial records %.0f MB written + read per launch\n",
         t_main, t_both, pairs * 3 * prod / t_both * 1e-6, part_bytes / 1e6);
  printf("  decision rule (VERDICT r04 item 5): keep only if <= 480 us including the merge at 32 x 1024^2 -> %s\n",
         (pairs == 32 && K == 1024) ? (t_both <= 480.f ? "KEEP" : "DROP") : "(other size: informative only)");
  return 0;
}
