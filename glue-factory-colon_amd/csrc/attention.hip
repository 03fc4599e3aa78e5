// Flash-style multi-head attention on fp32 MFMA (head_dim 64), score matrix never materialised.
//
// Replaces F.scaled_dot_product_attention in SelfBlock (reference
// gluefactory/models/matchers/lightglue.py:119-122,160-162) and the bidirectional cross attention
// of CrossBlock (lightglue.py:207-217: softmax over rows and over columns of the same similarity
// matrix), issued as two problems of one launch.
//
// Layout: one workgroup = 4 waves = 128 query rows of one (problem, head); each wave owns 32
// queries.  Keys/values stream through LDS in tiles of 64.  Per 32-key half tile a wave computes
//     S^T[key][q] = K . Q^T            A = K rows (LDS, ds_read_b128), B = Q (registers)
// so that the accumulator holds, for lane (q = lane&31, half h), the 16 keys r -> (r&3)+8(r>>2)+4h.
// Softmax statistics are then lane-local (+ one exchange with lane^32), and the accumulator
// registers are *already* the B operand of
//     O^T[d][q] += V^T[d][key] . P^T[key][q]      A = V[key][d] (LDS, ds_read_b32), B = P register r
// -- no transpose, no LDS round trip for P.  O^T keeps q on the lane, so the running rescale is a
// per-lane scalar multiply.
#include <stdlib.h>

#include "common.h"

#define AK 64    // keys per LDS tile
#define AD 64    // head dim
#define AKLD (AD + 4)

// QT = 32-query tiles per wave (1 or 2): a workgroup covers 128*QT queries.  With QT = 2 every K / V
// fragment read from LDS feeds two independent score tiles, the barrier count per query halves, and
// the MFMAs of one tile can issue while the softmax of the other runs on the VALU.
template <int QT>
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ Q, int ldq,
                                                           const float* __restrict__ Kp, int ldk,
                                                           const float* __restrict__ V, int ldv,
                                                           float* __restrict__ O, int ldo,
                                                           const int4* __restrict__ problems, float scale_log2e) {
  // double-buffered K / V tiles: [2][AK*AKLD] keys, then [2][AK*AD] values (67.6 KB -> 2 workgroups / CU)
  __shared__ __attribute__((aligned(16))) float smem[2 * AK * AKLD + 2 * AK * AD];
  constexpr int AQ = 128 * QT;

  const int4 pb = problems[blockIdx.z];
  const int q_row0 = pb.x, nq = pb.y, kv_row0 = pb.z, nk = pb.w;
  const int qt0 = blockIdx.x * AQ;
  if (qt0 >= nq) return;  // uniform for the whole workgroup
  const int head = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- Q fragments: lane (q, h) keeps Q[q][8g + 4h + s], g = 0..7, s = 0..3 ----
  int q[QT];
  float4 qf[QT][8];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    q[t] = qt0 + (wave * QT + t) * 32 + l31;
    const int qc = min(q[t], nq - 1);
    const float* qp = Q + (size_t)(q_row0 + qc) * ldq + head * AD + 4 * h;
#pragma unroll
    for (int g = 0; g < 8; ++g) qf[t][g] = *reinterpret_cast<const float4*>(qp + 8 * g);
  }

  f32x16 o[QT][2];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[t][0][r] = 0.f; o[t][1][r] = 0.f; }
    m_run[t] = -INFINITY;
    l_run[t] = 0.f;
  }

  const int ntiles = (nk + AK - 1) / AK;
  // staging: thread -> (key = tid>>4 (+16 i), 4 floats at c4); the next tile is prefetched into
  // registers while the current one is multiplied, then written to the other LDS buffer
  const int st_key = tid >> 4, st_c4 = (tid & 15) * 4;
  const float* kbase = Kp + head * AD + st_c4;
  const float* vbase = V + head * AD + st_c4;
  float4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;
#define ATT_LOAD(kt_)                                                                     \
  do {                                                                                    \
    const int kb_ = (kt_) * AK + st_key;                                                  \
    const size_t r0_ = kv_row0 + min(kb_, nk - 1), r1_ = kv_row0 + min(kb_ + 16, nk - 1); \
    const size_t r2_ = kv_row0 + min(kb_ + 32, nk - 1), r3_ = kv_row0 + min(kb_ + 48, nk - 1); \
    kr0 = *reinterpret_cast<const float4*>(kbase + r0_ * ldk);                            \
    kr1 = *reinterpret_cast<const float4*>(kbase + r1_ * ldk);                            \
    kr2 = *reinterpret_cast<const float4*>(kbase + r2_ * ldk);                            \
    kr3 = *reinterpret_cast<const float4*>(kbase + r3_ * ldk);                            \
    vr0 = *reinterpret_cast<const float4*>(vbase + r0_ * ldv);                            \
    vr1 = *reinterpret_cast<const float4*>(vbase + r1_ * ldv);                            \
    vr2 = *reinterpret_cast<const float4*>(vbase + r2_ * ldv);                            \
    vr3 = *reinterpret_cast<const float4*>(vbase + r3_ * ldv);                            \
  } while (0)
#define ATT_STORE(buf_)                                                                   \
  do {                                                                                    \
    float* kd_ = smem + (buf_) * AK * AKLD + st_key * AKLD + st_c4;                       \
    float* vd_ = smem + 2 * AK * AKLD + (buf_) * AK * AD + st_key * AD + st_c4;           \
    *reinterpret_cast<float4*>(kd_) = kr0;                                                \
    *reinterpret_cast<float4*>(kd_ + 16 * AKLD) = kr1;                                    \
    *reinterpret_cast<float4*>(kd_ + 32 * AKLD) = kr2;                                    \
    *reinterpret_cast<float4*>(kd_ + 48 * AKLD) = kr3;                                    \
    *reinterpret_cast<float4*>(vd_) = vr0;                                                \
    *reinterpret_cast<float4*>(vd_ + 16 * AD) = vr1;                                      \
    *reinterpret_cast<float4*>(vd_ + 32 * AD) = vr2;                                      \
    *reinterpret_cast<float4*>(vd_ + 48 * AD) = vr3;                                      \
  } while (0)

  ATT_LOAD(0);
  ATT_STORE(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const bool has_next = kt + 1 < ntiles;
    if (has_next) ATT_LOAD(kt + 1);
    const float* Ks = smem + (kt & 1) * AK * AKLD;
    const float* Vs = smem + 2 * AK * AKLD + (kt & 1) * AK * AD;

#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int key0 = kt * AK + half * 32;
      if (key0 >= nk) break;  // uniform
      f32x16 s[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      const float* kp = Ks + (half * 32 + l31) * AKLD + 4 * h;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float4 kf = *reinterpret_cast<const float4*>(kp + 8 * g);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          s[t] = mfma32(kf.x, qf[t][g].x, s[t]);
          s[t] = mfma32(kf.y, qf[t][g].y, s[t]);
          s[t] = mfma32(kf.z, qf[t][g].z, s[t]);
          s[t] = mfma32(kf.w, qf[t][g].w, s[t]);
        }
      }
      const bool tail = key0 + 32 > nk;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        // mask the tail keys, running max
        if (tail) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (key0 + acc_row(r, h) >= nk) s[t][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[t], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[t] - m_new) * scale_log2e);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[t][r] = __builtin_amdgcn_exp2f((s[t][r] - m_new) * scale_log2e);  // raw v_exp_f32
          rs += s[t][r];
        }
        rs += __shfl_xor(rs, 32);
        l_run[t] = l_run[t] * alpha + rs;
        // the running maximum rarely moves after the first tiles: skip the 32 rescale multiplies when
        // no lane of the wave needs them (alpha == 1 exactly -> bit-identical result)
        if (!__all(m_new == m_run[t])) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[t][0][r] *= alpha; o[t][1][r] *= alpha; }
        }
        m_run[t] = m_new;
      }
      // O^T += V^T P^T
      const float* vp = Vs + (half * 32 + 4 * h) * AD + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int krow = (r & 3) + 8 * (r >> 2);
        const float v0 = vp[krow * AD];
        const float v1 = vp[krow * AD + 32];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          o[t][0] = mfma32(v0, s[t][r], o[t][0]);
          o[t][1] = mfma32(v1, s[t][r], o[t][1]);
        }
      }
    }
    if (has_next) ATT_STORE((kt + 1) & 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[q][db*32 + 8*(r>>2) + 4h + (r&3)] ----
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (q[t] < nq) {
      const float inv = 1.f / l_run[t];
      float* op = O + (size_t)(q_row0 + q[t]) * ldo + head * AD + 4 * h;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(o[t][db][4 * g] * inv, o[t][db][4 * g + 1] * inv, o[t][db][4 * g + 2] * inv,
                                 o[t][db][4 * g + 3] * inv);
          *reinterpret_cast<float4*>(op + db * 32 + 8 * g) = v;
        }
    }
  }
}

extern "C" int gfc_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                             int ldo, const int32_t* problems, int n_problems, int max_nq, int heads, float scale,
                             void* stream) {
  if (!Q || !K || !V || !O || !problems || n_problems <= 0 || max_nq <= 0 || heads <= 0) return GFC_ERR_INVALID;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4) return GFC_ERR_INVALID;
  // tuning knob (tools/bench_kernels.py): GFC_ATTN_QT=1|2 q-tiles per wave
  static const int forced = [] { const char* e = getenv("GFC_ATTN_QT"); return e ? atoi(e) : 0; }();
  // two q-tiles per wave pay off once the grid still fills the chip twice over (256 CUs x 2 workgroups)
  const long long wgs2 = (long long)((max_nq + 255) / 256) * heads * n_problems;
  const int qt = forced ? forced : (wgs2 >= 1024 ? 2 : 1);
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  if (qt == 2) {
    dim3 grid((max_nq + 255) / 256, heads, n_problems);
    hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo,
                       reinterpret_cast<const int4*>(problems), sl2);
  } else {
    dim3 grid((max_nq + 127) / 128, heads, n_problems);
    hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo,
                       reinterpret_cast<const int4*>(problems), sl2);
  }
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
