// EXPERIMENTAL (opt-in, not on the default path): flash-style attention on the bf16 matrix pipe at fp32 accuracy.
//
// Same algorithm and data flow as attention.hip (S^T = K.Q^T so that soft-max statistics are lane-local and the
// probabilities are already the B operand of O^T += V^T.P^T), with every fp32 product evaluated as six bf16 MFMA
// products of three bf16 planes per operand (see conv_split.hip).  What changes:
//   * K tile in LDS as [key][plane][64 d] bf16, V tile TRANSPOSED as [d][plane][64 key slots] bf16 (both split while
//     they are staged); Q as bf16 planes in registers; the probabilities are split in registers after the exponent.
//   * v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane half; the accumulator of S^T holds, for lane half h,
//     the keys kappa(b, h, j) = (j & 3) + 8 (2 b + (j >> 2)) + 4 h in registers r = 8 b + j.  Any k order is a valid
//     dot product as long as both operands agree, so V^T's key slots are stored in exactly that order and the
//     accumulator registers 8b .. 8b+7 become the B fragment of k block b without any lane movement.
// One workgroup = 4 waves = 128 queries of one (problem, head); key tiles of 64; single-buffered LDS (51 KB).
#include "common.h"

#define SAK 64
#define SAD 64
#define SAROW 400  // bytes per LDS row: 3 planes x 64 bf16 (384 B) + 16 B pad (pitch / 16 odd)

typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void asplit3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

// split 8 floats into the three bf16x8 planes of an MFMA fragment
#define ASPLIT8(f0_, f1_, P0_, P1_, P2_)                                                        \
  do {                                                                                          \
    __bf16 h_[8], m_[8], l_[8];                                                                 \
    asplit3((f0_).x, h_[0], m_[0], l_[0]); asplit3((f0_).y, h_[1], m_[1], l_[1]);               \
    asplit3((f0_).z, h_[2], m_[2], l_[2]); asplit3((f0_).w, h_[3], m_[3], l_[3]);               \
    asplit3((f1_).x, h_[4], m_[4], l_[4]); asplit3((f1_).y, h_[5], m_[5], l_[5]);               \
    asplit3((f1_).z, h_[6], m_[6], l_[6]); asplit3((f1_).w, h_[7], m_[7], l_[7]);               \
    P0_ = abf16x8{h_[0], h_[1], h_[2], h_[3], h_[4], h_[5], h_[6], h_[7]};                      \
    P1_ = abf16x8{m_[0], m_[1], m_[2], m_[3], m_[4], m_[5], m_[6], m_[7]};                      \
    P2_ = abf16x8{l_[0], l_[1], l_[2], l_[3], l_[4], l_[5], l_[6], l_[7]};                      \
  } while (0)

#define AMM6(acc_, A_, B_)                                                                      \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_##2, B_##0, acc_, 0, 0, 0);                  \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_##0, B_##2, acc_, 0, 0, 0);                  \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_##1, B_##1, acc_, 0, 0, 0);                  \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_##1, B_##0, acc_, 0, 0, 0);                  \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_##0, B_##1, acc_, 0, 0, 0);                  \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_##0, B_##0, acc_, 0, 0, 0);

__global__ __launch_bounds__(256, 2) void attention_split_kernel(const float* __restrict__ Q, int ldq,
                                                                 const float* __restrict__ Kp, int ldk,
                                                                 const float* __restrict__ V, int ldv,
                                                                 float* __restrict__ O, int ldo,
                                                                 const int4* __restrict__ problems, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char asm_[];
  char* Ks = asm_;                  // [64 keys][SAROW]
  char* Vt = asm_ + SAK * SAROW;    // [64 d][SAROW]
  const int4 pb = problems[blockIdx.z];
  const int q_row0 = pb.x, nq = pb.y, kv_row0 = pb.z, nk = pb.w;
  const int qt0 = blockIdx.x * 128;
  if (qt0 >= nq) return;  // uniform for the whole workgroup
  const int head = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- Q planes: lane (q, h) keeps Q[q][16 blk + 8 h + j], j = 0..7, for blk = 0..3 ----
  const int q = qt0 + wave * 32 + l31;
  abf16x8 q00, q01, q02, q10, q11, q12, q20, q21, q22, q30, q31, q32;  // q<blk><plane>
  {
    const float* qp = Q + (size_t)(q_row0 + min(q, nq - 1)) * ldq + head * SAD + 8 * h;
    float4 a, b;
    a = *reinterpret_cast<const float4*>(qp + 0);  b = *reinterpret_cast<const float4*>(qp + 4);  ASPLIT8(a, b, q00, q01, q02);
    a = *reinterpret_cast<const float4*>(qp + 16); b = *reinterpret_cast<const float4*>(qp + 20); ASPLIT8(a, b, q10, q11, q12);
    a = *reinterpret_cast<const float4*>(qp + 32); b = *reinterpret_cast<const float4*>(qp + 36); ASPLIT8(a, b, q20, q21, q22);
    a = *reinterpret_cast<const float4*>(qp + 48); b = *reinterpret_cast<const float4*>(qp + 52); ASPLIT8(a, b, q30, q31, q32);
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  // ---- staging: thread -> keys st_key + 16 i (i < 4), 4 floats at c4; split while writing to LDS ----
  const int st_key = tid >> 4, st_c4 = (tid & 15) * 4;
  const float* kbase = Kp + head * SAD + st_c4;
  const float* vbase = V + head * SAD + st_c4;
  float4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;
#define SA_LOAD(kt_)                                                                                         \
  do {                                                                                                       \
    const int kb_ = (kt_) * SAK + st_key;                                                                    \
    { const size_t r_ = kv_row0 + min(kb_ + 0, nk - 1);  kr0 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr0 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); } \
    { const size_t r_ = kv_row0 + min(kb_ + 16, nk - 1); kr1 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr1 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); } \
    { const size_t r_ = kv_row0 + min(kb_ + 32, nk - 1); kr2 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr2 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); } \
    { const size_t r_ = kv_row0 + min(kb_ + 48, nk - 1); kr3 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr3 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); } \
  } while (0)
  // key slot of V^T (see the header): key kk in [0, 32) of a half -> position (b*2 + hh)*8 + j
#define SA_SLOT(key_) ((((key_) >> 5) << 5) + (((((key_) & 31) >> 4) * 2 + ((((key_) & 15) >> 2) & 1)) << 3) + \
                       (((((key_) & 15) >> 3) << 2) | ((key_) & 3)))
#define SA_STORE_ONE(kr_, vr_, key_)                                                                         \
  do {                                                                                                       \
    __bf16 h0_, m0_, l0_, h1_, m1_, l1_, h2_, m2_, l2_, h3_, m3_, l3_;                                       \
    asplit3((kr_).x, h0_, m0_, l0_); asplit3((kr_).y, h1_, m1_, l1_);                                        \
    asplit3((kr_).z, h2_, m2_, l2_); asplit3((kr_).w, h3_, m3_, l3_);                                        \
    char* kd_ = Ks + (key_) * SAROW + st_c4 * 2;                                                             \
    *reinterpret_cast<abf16x4*>(kd_) = abf16x4{h0_, h1_, h2_, h3_};                                          \
    *reinterpret_cast<abf16x4*>(kd_ + 128) = abf16x4{m0_, m1_, m2_, m3_};                                    \
    *reinterpret_cast<abf16x4*>(kd_ + 256) = abf16x4{l0_, l1_, l2_, l3_};                                    \
    asplit3((vr_).x, h0_, m0_, l0_); asplit3((vr_).y, h1_, m1_, l1_);                                        \
    asplit3((vr_).z, h2_, m2_, l2_); asplit3((vr_).w, h3_, m3_, l3_);                                        \
    char* vd_ = Vt + st_c4 * SAROW + SA_SLOT(key_) * 2;                                                      \
    *reinterpret_cast<__bf16*>(vd_) = h0_;               *reinterpret_cast<__bf16*>(vd_ + 128) = m0_;               *reinterpret_cast<__bf16*>(vd_ + 256) = l0_; \
    *reinterpret_cast<__bf16*>(vd_ + SAROW) = h1_;       *reinterpret_cast<__bf16*>(vd_ + SAROW + 128) = m1_;       *reinterpret_cast<__bf16*>(vd_ + SAROW + 256) = l1_; \
    *reinterpret_cast<__bf16*>(vd_ + 2 * SAROW) = h2_;   *reinterpret_cast<__bf16*>(vd_ + 2 * SAROW + 128) = m2_;   *reinterpret_cast<__bf16*>(vd_ + 2 * SAROW + 256) = l2_; \
    *reinterpret_cast<__bf16*>(vd_ + 3 * SAROW) = h3_;   *reinterpret_cast<__bf16*>(vd_ + 3 * SAROW + 128) = m3_;   *reinterpret_cast<__bf16*>(vd_ + 3 * SAROW + 256) = l3_; \
  } while (0)
#define SA_STORE()                                                                                           \
  do {                                                                                                       \
    SA_STORE_ONE(kr0, vr0, st_key);                                                                          \
    SA_STORE_ONE(kr1, vr1, st_key + 16);                                                                     \
    SA_STORE_ONE(kr2, vr2, st_key + 32);                                                                     \
    SA_STORE_ONE(kr3, vr3, st_key + 48);                                                                     \
  } while (0)

  const int ntiles = (nk + SAK - 1) / SAK;
  SA_LOAD(0);
  SA_STORE();
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const bool has_next = kt + 1 < ntiles;
    if (has_next) SA_LOAD(kt + 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int key0 = kt * SAK + half * 32;
      if (key0 >= nk) break;  // uniform
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      // S^T[key][q] = K . Q^T: A = K planes of key (half*32 + l31), d = 16 blk + 8 h + j
      const char* kp = Ks + (half * 32 + l31) * SAROW + h * 16;
#define SA_QK(blk_, Q_)                                                                                      \
  {                                                                                                          \
    const abf16x8 k0 = *reinterpret_cast<const abf16x8*>(kp + 32 * (blk_));                                  \
    const abf16x8 k1 = *reinterpret_cast<const abf16x8*>(kp + 128 + 32 * (blk_));                            \
    const abf16x8 k2 = *reinterpret_cast<const abf16x8*>(kp + 256 + 32 * (blk_));                            \
    AMM6(s, k, Q_)                                                                                           \
  }
      SA_QK(0, q0) SA_QK(1, q1) SA_QK(2, q2) SA_QK(3, q3)
#undef SA_QK
      if (key0 + 32 > nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + acc_row(r, h) >= nk) s[r] = -INFINITY;
      }
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
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
      m_run = m_new;
      // probabilities -> bf16 planes: registers 8b .. 8b+7 are the B fragment of key block b
      abf16x8 p00, p01, p02, p10, p11, p12;
      {
        const float4 a0 = make_float4(s[0], s[1], s[2], s[3]), a1 = make_float4(s[4], s[5], s[6], s[7]);
        const float4 b0 = make_float4(s[8], s[9], s[10], s[11]), b1 = make_float4(s[12], s[13], s[14], s[15]);
        ASPLIT8(a0, a1, p00, p01, p02);
        ASPLIT8(b0, b1, p10, p11, p12);
      }
      // O^T[d][q] += V^T[d][key slots] . P^T: A = V^T planes of row d = dt*32 + l31, slots (half*32 + (b*2 + h)*8 ..)
      const char* vp = Vt + l31 * SAROW + (half * 32 + h * 8) * 2;
#define SA_PV(dt_, b_, O_, P_)                                                                               \
  {                                                                                                          \
    const char* v_ = vp + (dt_) * 32 * SAROW + (b_) * 32;                                                    \
    const abf16x8 v0 = *reinterpret_cast<const abf16x8*>(v_);                                                \
    const abf16x8 v1 = *reinterpret_cast<const abf16x8*>(v_ + 128);                                          \
    const abf16x8 v2 = *reinterpret_cast<const abf16x8*>(v_ + 256);                                          \
    AMM6(O_, v, P_)                                                                                          \
  }
      SA_PV(0, 0, o0, p0) SA_PV(0, 1, o0, p1) SA_PV(1, 0, o1, p0) SA_PV(1, 1, o1, p1)
#undef SA_PV
    }
    __syncthreads();  // every wave is done with this tile
    if (has_next) {
      SA_STORE();
      __syncthreads();
    }
  }

  // ---- normalise and store: lane holds O[q][db*32 + 8*(r>>2) + 4h + (r&3)] ----
  if (q < nq) {
    const float inv = 1.f / l_run;
    float* op = O + (size_t)(q_row0 + q) * ldo + head * SAD + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<float4*>(op + 8 * g) =
          make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<float4*>(op + 32 + 8 * g) =
          make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
}

extern "C" int gfc_attention_split(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                                   int ldo, const int32_t* problems, int n_problems, int max_nq, int heads, float scale,
                                   void* stream) {
  if (!Q || !K || !V || !O || !problems || n_problems <= 0 || max_nq <= 0 || heads <= 0) return GFC_ERR_INVALID;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4) return GFC_ERR_INVALID;
  const float sl2 = scale * 1.4426950408889634f;
  const size_t lds = 2 * SAK * SAROW;
  hipLaunchKernelGGL(attention_split_kernel, dim3((max_nq + 127) / 128, heads, n_problems), dim3(256), lds,
                     (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, O, ldo, reinterpret_cast<const int4*>(problems), sl2);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
