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

#ifdef ATT_DIAG  // diagnostic build (tools/micro/attn_timeline.py): per-wave cycle accounting, 8 words per wave
__device__ unsigned long long* g_att_diag_dev = nullptr;
extern "C" void gfc_diag_set_attn_stamps(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_att_diag_dev), &q, sizeof(q));
}
#define ATT_T(v_) const unsigned long long v_ = __builtin_readcyclecounter()
#else
#define ATT_T(v_) do {} while (0)
#endif

// QT = 32-query tiles per wave (1 or 2): a workgroup covers 128*QT queries.  With QT = 2 every K / V
// fragment read from LDS feeds two independent score tiles, the barrier count per query halves, and
// the MFMAs of one tile can issue while the softmax of the other runs on the VALU.
// AW = waves per workgroup (4, or 2 for small problems: twice the workgroups for the same queries).
template <int QT, int AW>
__global__ __launch_bounds__(64 * AW, AW == 4 ? 2 : 1) void attention_kernel(const float* __restrict__ Q, int ldq,
                                                           const float* __restrict__ Kp, int ldk,
                                                           const float* __restrict__ V, int ldv,
                                                           float* __restrict__ O, int ldo,
                                                           const int4* __restrict__ problems, float scale_log2e,
                                                           int ksplit, float* __restrict__ part, int max_nq, int xcd_remap) {
  // double-buffered K / V tiles: [2][AK*AKLD] keys, then [2][AK*AD] values (67.6 KB -> 2 workgroups / CU)
  __shared__ __attribute__((aligned(16))) float smem[2 * AK * AKLD + 2 * AK * AD];
  constexpr int AQ = 32 * AW * QT;
  constexpr int T = 64 * AW;
  constexpr int NLD = (AK * AD / 4) / T;  // float4 of K (and of V) per thread per tile

  // XCD-aware order (common.h): the query blocks (and key splits) of one (problem, head) run at the same time behind
  // one L2, so its K and V rows come from HBM once instead of once per query block
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (xcd_remap) {
    const unsigned per_z = gridDim.x * gridDim.y;
    unsigned t = gfc_xcd_chunk(bx + gridDim.x * (by + gridDim.y * bz), per_z * gridDim.z);
    bz = t / per_z;
    t -= bz * per_z;
    by = t / gridDim.x;
    bx = t - by * gridDim.x;
  }
  const int4 pb = problems[bz];
  const int q_row0 = pb.x, nq = pb.y, kv_row0 = pb.z, nk = pb.w;
  // key split (small problems only): bx = q-block * ksplit + s; split s walks its share of the key
  // tiles and leaves an un-normalised partial (O, m, l) for attention_merge_kernel
  const int ks = bx % ksplit;
  const int qt0 = (bx / ksplit) * AQ;
  if (qt0 >= nq) return;  // uniform for the whole workgroup
  const int head = by;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
#ifdef ATT_DIAG
  unsigned long long d_bar = 0, d_store = 0;
#endif
  ATT_T(t_entry);

  // ---- Q fragments: lane (q, h) keeps Q[q][8g + 4h + s], g = 0..7, s = 0..3 ----
  int q[QT];
  float4 qf[QT][8];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    q[t] = qt0 + (wave * QT + t) * 32 + l31;
    const int qc = min(q[t], nq - 1);
    const float* qp = Q + (size_t)(q_row0 + qc) * ldq + head * AD + 4 * h;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      // scale * log2(e) folded into the query fragment once (round 5): the scores then come out of the matrix pipe in
      // the soft-max's own units and the per-element multiply in front of every v_exp_f32 is gone -- on this part every
      // VALU instruction costs the fp32 matrix pipe its issue cycles (tools/micro/mfma_valu_hybrid.hip)
      const float4 qv = *reinterpret_cast<const float4*>(qp + 8 * g);
      qf[t][g] = make_float4(qv.x * scale_log2e, qv.y * scale_log2e, qv.z * scale_log2e, qv.w * scale_log2e);
    }
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

  const int ntiles_all = (nk + AK - 1) / AK;
  const int tiles_per = (ntiles_all + ksplit - 1) / ksplit;
  const int kt0 = ks * tiles_per;
  const int kt1 = min(ntiles_all, kt0 + tiles_per);
  // staging: thread -> (key = tid>>4 (+16 i), 4 floats at c4); the next tile is prefetched into
  // registers while the current one is multiplied, then written to the other LDS buffer
  const int st_key = tid >> 4, st_c4 = (tid & 15) * 4;  // keys st_key + (T/16) * i
  const float* kbase = Kp + head * AD + st_c4;
  const float* vbase = V + head * AD + st_c4;
  // named prefetch registers (arrays were demoted to private memory by hipcc)
  float4 kr0, kr1, kr2, kr3, kr4, kr5, kr6, kr7, vr0, vr1, vr2, vr3, vr4, vr5, vr6, vr7;
  kr4 = kr5 = kr6 = kr7 = vr4 = vr5 = vr6 = vr7 = make_float4(0.f, 0.f, 0.f, 0.f);
  static_assert(NLD == 4 || NLD == 8, "staging layout");
#define ATT_LOAD(kt_)                                                                                                                                                                       \
  do {                                                                                                                                                                                      \
    const int kb_ = (kt_) * AK + st_key;                                                                                                                                                    \
    { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 0, nk - 1); kr0 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr0 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }    \
    { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 1, nk - 1); kr1 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr1 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }    \
    { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 2, nk - 1); kr2 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr2 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }    \
    { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 3, nk - 1); kr3 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr3 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }    \
    if constexpr (NLD == 8) {                                                                                                                                                               \
      { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 4, nk - 1); kr4 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr4 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }  \
      { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 5, nk - 1); kr5 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr5 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }  \
      { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 6, nk - 1); kr6 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr6 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }  \
      { const size_t r_ = kv_row0 + min(kb_ + (T / 16) * 7, nk - 1); kr7 = *reinterpret_cast<const float4*>(kbase + r_ * ldk); vr7 = *reinterpret_cast<const float4*>(vbase + r_ * ldv); }  \
    }                                                                                                                                                                                       \
  } while (0)
#define ATT_STORE(buf_)                                                                                                                                    \
  do {                                                                                                                                                     \
    float* kd_ = smem + (buf_) * AK * AKLD + st_c4;                                                                                                        \
    float* vd_ = smem + 2 * AK * AKLD + (buf_) * AK * AD + st_c4;                                                                                          \
    { const int key_ = st_key + (T / 16) * 0; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr0; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr0; }    \
    { const int key_ = st_key + (T / 16) * 1; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr1; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr1; }    \
    { const int key_ = st_key + (T / 16) * 2; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr2; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr2; }    \
    { const int key_ = st_key + (T / 16) * 3; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr3; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr3; }    \
    if constexpr (NLD == 8) {                                                                                                                              \
      { const int key_ = st_key + (T / 16) * 4; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr4; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr4; }  \
      { const int key_ = st_key + (T / 16) * 5; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr5; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr5; }  \
      { const int key_ = st_key + (T / 16) * 6; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr6; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr6; }  \
      { const int key_ = st_key + (T / 16) * 7; *reinterpret_cast<float4*>(kd_ + key_ * AKLD) = kr7; *reinterpret_cast<float4*>(vd_ + key_ * AD) = vr7; }  \
    }                                                                                                                                                      \
  } while (0)

  if (kt0 < kt1) {
    ATT_LOAD(kt0);
    ATT_STORE(0);
  }
  __syncthreads();
  ATT_T(t_loop);
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool has_next = kt + 1 < kt1;
    if (has_next) ATT_LOAD(kt + 1);
    const int buf = (kt - kt0) & 1;
    const float* Ks = smem + buf * AK * AKLD;
    const float* Vs = smem + 2 * AK * AKLD + buf * AK * AD;

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
        const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m_new);  // raw v_exp_f32 (scores already in log2 units)
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
#ifdef ATT_DIAG
    {
      ATT_T(t_s0);
      if (has_next) ATT_STORE(buf ^ 1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      ATT_T(t_s1);
      __syncthreads();
      ATT_T(t_s2);
      d_store += t_s1 - t_s0;
      d_bar += t_s2 - t_s1;
    }
#else
    if (has_next) ATT_STORE(buf ^ 1);
    __syncthreads();
#endif
  }
  ATT_T(t_end);

  // ---- key-split partials: [problem][head][q][split][64 O | m | l] ----
  if (part != nullptr) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (q[t] < nq) {
        float* pp = part + ((((size_t)bz * gridDim.y + head) * max_nq + q[t]) * ksplit + ks) * 66;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float* dst = pp + db * 32 + 8 * g + 4 * h;  // 8-byte aligned (66 floats per record)
            dst[0] = o[t][db][4 * g]; dst[1] = o[t][db][4 * g + 1];
            dst[2] = o[t][db][4 * g + 2]; dst[3] = o[t][db][4 * g + 3];
          }
        if (h == 0) { pp[64] = m_run[t]; pp[65] = l_run[t]; }
      }
    }
    return;
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
#ifdef ATT_DIAG
  if (g_att_diag_dev && lane == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long* dd = g_att_diag_dev +
        ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * AW + wave) * 8;
    dd[0] = t_loop - t_entry; dd[1] = t_end - t_loop; dd[2] = d_bar; dd[3] = d_store;
    dd[4] = __builtin_readcyclecounter() - t_end; dd[5] = hw; dd[6] = t_entry;
  }
#endif
}

// combine the key-split partials: one wave per (query, head); lane = channel
__global__ __launch_bounds__(256) void attention_merge_kernel(const float* __restrict__ part, float* __restrict__ O,
                                                              int ldo, const int4* __restrict__ problems, int ksplit,
                                                              int max_nq, float scale_log2e) {
  const int4 pb = problems[blockIdx.z];
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), head = blockIdx.y, lane = threadIdx.x & 63;
  if (q >= pb.y) return;
  const float* pp = part + (((size_t)blockIdx.z * gridDim.y + head) * max_nq + q) * ksplit * 66;
  float m = -INFINITY;
  for (int s = 0; s < ksplit; ++s) m = fmaxf(m, pp[s * 66 + 64]);
  float acc = 0.f, l = 0.f;
  for (int s = 0; s < ksplit; ++s) {
    const float ms = pp[s * 66 + 64];
    const float w = (ms == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ms - m);  // (partials carry the maximum in log2 units)
    acc += w * pp[s * 66 + lane];
    l += w * pp[s * 66 + 65];
  }
  O[(size_t)(pb.x + q) * ldo + head * AD + lane] = acc / l;
}

// Scratch for the key-split path (0 when the problem set is large enough not to need it).
extern "C" size_t gfc_attention_workspace_bytes(int n_problems, int max_nq, int heads) {
  if (n_problems <= 0 || max_nq <= 0 || heads <= 0) return 0;
  const long long wgs128 = (long long)((max_nq + 127) / 128) * heads * n_problems;
  if (wgs128 >= 256) return 0;
  return gfc_align((size_t)n_problems * heads * max_nq * 8 * 66 * sizeof(float));
}

extern "C" int gfc_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                             int ldo, const int32_t* problems, int n_problems, int max_nq, int heads, float scale,
                             void* ws, size_t ws_bytes, void* stream) {
  if (!Q || !K || !V || !O || !problems || n_problems <= 0 || max_nq <= 0 || heads <= 0) return GFC_ERR_INVALID;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4) return GFC_ERR_INVALID;
  // tuning knob (tools/bench_kernels.py): GFC_ATTN_CFG = 1: 2 q-tiles/wave, 4 waves (256 queries / workgroup)
  //                                                      2: 1 q-tile/wave, 4 waves (128);  3: 1 q-tile, 2 waves (64)
  const int forced = gfc_knobs().attn_cfg;
  auto wgs = [&](int aq) { return (long long)((max_nq + aq - 1) / aq) * heads * n_problems; };
  // the largest query block that still fills the chip twice over (256 CUs x 2 workgroups)
  // (cfg 3, 64 queries per 2-wave workgroup, measured no faster than cfg 2 at batch 1: knob only)
  const int cfg = forced ? forced : (wgs(256) >= 1024 ? 1 : 2);
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  const int4* pt = reinterpret_cast<const int4*>(problems);
  // key split for small problem sets (batch 1..2): few 128-query blocks cannot fill 1024 SIMDs, so each block's
  // keys are shared out over up to 8 workgroups and a tiny merge kernel combines the partial soft-maxes
  const int xcd = gfc_knobs().xcd_remap != 0;
  int ksplit = 1;
  if (cfg == 2 && ws != nullptr) {
    const long long w = wgs(128);
    int want = w >= 256 ? 1 : (int)((511 + w) / w);
    if (want > 8) want = 8;
    while (want > 1 && ws_bytes < (size_t)n_problems * heads * max_nq * want * 66 * sizeof(float)) --want;
    ksplit = want < 1 ? 1 : want;
  }
  if (cfg == 1) {
    hipLaunchKernelGGL((attention_kernel<2, 4>), dim3((max_nq + 255) / 256, heads, n_problems), dim3(256), 0, st, Q,
                       ldq, K, ldk, V, ldv, O, ldo, pt, sl2, 1, (float*)nullptr, max_nq, xcd);
  } else if (cfg == 2) {
    float* part = ksplit > 1 ? (float*)ws : nullptr;
    hipLaunchKernelGGL((attention_kernel<1, 4>), dim3(((max_nq + 127) / 128) * ksplit, heads, n_problems), dim3(256), 0,
                       st, Q, ldq, K, ldk, V, ldv, O, ldo, pt, sl2, ksplit, part, max_nq, xcd);
    if (ksplit > 1)
      hipLaunchKernelGGL(attention_merge_kernel, dim3((max_nq + 3) / 4, heads, n_problems), dim3(256), 0, st, part, O,
                         ldo, pt, ksplit, max_nq, sl2);
  } else {
    hipLaunchKernelGGL((attention_kernel<1, 2>), dim3((max_nq + 63) / 64, heads, n_problems), dim3(128), 0, st, Q, ldq,
                       K, ldk, V, ldv, O, ldo, pt, sl2, 1, (float*)nullptr, max_nq, xcd);
  }
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
