// HPatches match metrics on the GPU ("next" row of the scope table: the caller right after the matcher).
//
// Replaces eval_matches_homography (reference gluefactory/eval/utils.py:141-185) with its helpers
// sym_homography_error (gluefactory/geometry/homography.py:314-323) and gt_matches_from_homography
// (gluefactory/geometry/gt_generation.py:730-801, pos_th = neg_th as the evaluation calls it).
// One workgroup per pair; the M x N distance matrix is never materialised: a row sweep and a column
// sweep keep running (min, argmin) per key point in registers (first index wins ties, like torch.min).
#include "common.h"

#define EM_THREADS 256

__device__ __forceinline__ void warp_pt(const float* Hm, float x, float y, float eps, float& ox, float& oy) {
  // to_homogeneous(p) @ H^T then division by (w + eps): einsum order x*H[r][0] + y*H[r][1] + 1*H[r][2]
  const float wx = x * Hm[0] + y * Hm[1] + Hm[2];
  const float wy = x * Hm[3] + y * Hm[4] + Hm[5];
  const float ww = x * Hm[6] + y * Hm[7] + Hm[8];
  ox = wx / (ww + eps);
  oy = wy / (ww + eps);
}

__global__ __launch_bounds__(EM_THREADS) void eval_matches_kernel(const float* __restrict__ kp0,
                                                                  const float* __restrict__ kp1,
                                                                  const long long* __restrict__ m0,
                                                                  const float* __restrict__ H,
                                                                  const float* __restrict__ Hinv, int M, int N,
                                                                  float pos_th, float neg_th, float* __restrict__ out,
                                                                  long long* __restrict__ gt_m0_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* k01 = sm;              // [M][2] kp0 warped into image 1 (eps 1e-5, warp_points_torch)
  float* k10 = k01 + 2 * M;     // [N][2] kp1 warped into image 0
  float* a0 = k10 + 2 * N;      // [M][2] local copy of kp0
  float* a1 = a0 + 2 * M;       // [N][2] local copy of kp1
  int* min1 = reinterpret_cast<int*>(a1 + 2 * N);  // [N] argmin over rows
  __shared__ float red[6];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* p0 = kp0 + (size_t)b * M * 2;
  const float* p1 = kp1 + (size_t)b * N * 2;
  const long long* mm = m0 + (size_t)b * M;
  float Hm[9], Hi[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Hm[i] = H[b * 9 + i]; Hi[i] = Hinv[b * 9 + i]; }
  if (tid < 6) red[tid] = 0.f;
  for (int i = tid; i < M; i += EM_THREADS) {
    const float x = p0[2 * i], y = p0[2 * i + 1];
    a0[2 * i] = x; a0[2 * i + 1] = y;
    warp_pt(Hm, x, y, 1e-5f, k01[2 * i], k01[2 * i + 1]);
  }
  for (int j = tid; j < N; j += EM_THREADS) {
    const float x = p1[2 * j], y = p1[2 * j + 1];
    a1[2 * j] = x; a1[2 * j + 1] = y;
    warp_pt(Hi, x, y, 1e-5f, k10[2 * j], k10[2 * j + 1]);
  }
  __syncthreads();
  // column sweep: argmin_i max(d0, d1)
  for (int j = tid; j < N; j += EM_THREADS) {
    const float x1 = a1[2 * j], y1 = a1[2 * j + 1], xb = k10[2 * j], yb = k10[2 * j + 1];
    float best = INFINITY;
    int bi = 0;
    for (int i = 0; i < M; ++i) {
      const float dx0 = k01[2 * i] - x1, dy0 = k01[2 * i + 1] - y1;
      const float dx1 = a0[2 * i] - xb, dy1 = a0[2 * i + 1] - yb;
      const float d = fmaxf(dx0 * dx0 + dy0 * dy0, dx1 * dx1 + dy1 * dy1);
      if (d < best) { best = d; bi = i; }
    }
    min1[j] = bi;
  }
  __syncthreads();
  // row sweep + per-row verdicts
  float s_match = 0.f, s_p1 = 0.f, s_p3 = 0.f, s_gt = 0.f, s_rec = 0.f, s_mask = 0.f, s_prec = 0.f;
  for (int i = tid; i < M; i += EM_THREADS) {
    const float x0 = a0[2 * i], y0 = a0[2 * i + 1], xa = k01[2 * i], ya = k01[2 * i + 1];
    float best = INFINITY, best_d0 = INFINITY;
    int bj = 0;
    for (int j = 0; j < N; ++j) {
      const float dx0 = xa - a1[2 * j], dy0 = ya - a1[2 * j + 1];
      const float dx1 = x0 - k10[2 * j], dy1 = y0 - k10[2 * j + 1];
      const float d0 = dx0 * dx0 + dy0 * dy0;
      const float d = fmaxf(d0, dx1 * dx1 + dy1 * dy1);
      if (d < best) { best = d; bj = j; }
      best_d0 = fminf(best_d0, d0);
    }
    long long gt = -2;  // ignore
    if (N > 0 && min1[bj] == i && best < pos_th * pos_th) gt = bj;
    if (N == 0 || best_d0 > neg_th * neg_th) gt = -1;  // unmatched
    if (N == 0) gt = -1;
    if (gt_m0_out) gt_m0_out[(size_t)b * M + i] = gt;
    const long long m = mm[i];
    if (m > -1) {
      // symmetric transfer error of the predicted match (plain division, homography.py:314-323)
      float ax, ay, bx, by;
      warp_pt(Hm, x0, y0, 0.f, ax, ay);
      const float x1 = a1[2 * m], y1 = a1[2 * m + 1];
      warp_pt(Hi, x1, y1, 0.f, bx, by);
      const float e01 = sqrtf((ax - x1) * (ax - x1) + (ay - y1) * (ay - y1));
      const float e10 = sqrtf((bx - x0) * (bx - x0) + (by - y0) * (by - y0));
      const float err = (e01 + e10) / 2.f;
      s_match += 1.f;
      s_p1 += err < 1.f ? 1.f : 0.f;
      s_p3 += err < 3.f ? 1.f : 0.f;
    }
    if (gt > -1) { s_gt += 1.f; s_rec += (m == gt) ? 1.f : 0.f; }
    if (m > -1 && gt >= -1) { s_mask += 1.f; s_prec += (m == gt) ? 1.f : 0.f; }
  }
  // block reduction (7 sums): wave shuffle then LDS atomics
  float v[7] = {s_match, s_p1, s_p3, s_gt, s_rec, s_mask, s_prec};
  __shared__ float acc7[7];
  if (tid < 7) acc7[tid] = 0.f;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    float t = wave_sum(v[q]);
    if ((tid & 63) == 0) atomicAdd(&acc7[q], t);
  }
  __syncthreads();
  if (tid == 0) {
    float* o = out + (size_t)b * 6;
    const float nm = acc7[0];
    o[0] = nm > 0.f ? acc7[1] / nm : 0.f;  // prec@1px (mean over matches, nan -> 0)
    o[1] = nm > 0.f ? acc7[2] / nm : 0.f;  // prec@3px
    o[2] = nm;                             // num_matches
    o[3] = (M + N) / 2.f;                  // num_keypoints
    o[4] = acc7[4] / (1e-8f + acc7[3]);    // gt_match_recall
    o[5] = acc7[6] / (1e-8f + acc7[5]);    // gt_match_precision
  }
}

extern "C" int gfc_eval_matches_homography(const float* kp0, const float* kp1, const int64_t* m0, const float* H,
                                           const float* Hinv, int B, int M, int N, float pos_th, float neg_th,
                                           float* out, int64_t* gt_m0_out, void* stream) {
  if (!kp0 || !kp1 || !m0 || !H || !Hinv || !out || B <= 0 || M < 0 || N < 0) return GFC_ERR_INVALID;
  const size_t lds = (size_t)(4 * (M + N)) * sizeof(float) + (size_t)N * sizeof(int) + 64;
  if (lds > 160 * 1024) return GFC_ERR_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)eval_matches_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(eval_matches_kernel, dim3(B), dim3(EM_THREADS), lds, (hipStream_t)stream, kp0, kp1,
                     (const long long*)m0, H, Hinv, M, N, pos_th, neg_th, out, (long long*)gt_m0_out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ---- weighted DLT homography + corner error ("next" row rank 3) ---------------------------------------------
// Replaces eval_homography_dlt (reference gluefactory/eval/utils.py:276-302): kornia's find_homography_dlt
// (Hartley-normalised DLT, 2 rows per match, A^T diag(w) A, eigenvector of the smallest eigenvalue, de-normalise,
// divide by H[2][2] + 1e-8) followed by homography_corner_error (gluefactory/geometry/homography.py:336-342).
// One workgroup per pair; the 2n x 9 design matrix is never formed: each thread accumulates the 45 unique entries
// of the 9x9 normal matrix in fp64, a cyclic Jacobi iteration in LDS (thread 0) finds the eigenvector.

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of NV doubles per thread; result valid in every thread (via LDS)
template <int NV>
__device__ __forceinline__ void block_sum_f64(double* v, double* lds /* [4][NV] */, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const double t = wave_sum_f64(v[q]);
    if ((tid & 63) == 0) lds[wave * NV + q] = t;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = lds[q] + lds[NV + q] + lds[2 * NV + q] + lds[3 * NV + q];
  __syncthreads();
}

__global__ __launch_bounds__(EM_THREADS) void dlt_kernel(const float* __restrict__ kp0, const float* __restrict__ kp1,
                                                         const long long* __restrict__ m0,
                                                         const float* __restrict__ sc0, const float* __restrict__ Hgt,
                                                         const float* __restrict__ size0, int M, int N,
                                                         float* __restrict__ Hout, float* __restrict__ err_out) {
  __shared__ double red[4 * 45];
  __shared__ double A[81], V[81];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* p0 = kp0 + (size_t)b * M * 2;
  const float* p1 = kp1 + (size_t)b * N * 2;
  const long long* mm = m0 + (size_t)b * M;
  const float* ss = sc0 + (size_t)b * M;
  // pass 1: count and centroids
  double s5[5] = {0, 0, 0, 0, 0};
  for (int i = tid; i < M; i += EM_THREADS) {
    const long long j = mm[i];
    if (j > -1 && j < N) {
      s5[0] += 1.0; s5[1] += p0[2 * i]; s5[2] += p0[2 * i + 1]; s5[3] += p1[2 * j]; s5[4] += p1[2 * j + 1];
    }
  }
  block_sum_f64<5>(s5, red, tid);
  const double n = s5[0];
  if (n < 4.0) {  // find_homography_dlt asserts >= 4 points; the caller turns that into H = inf (utils.py:291-292)
    if (tid < 9) Hout[b * 9 + tid] = INFINITY;
    if (tid == 0) err_out[b] = INFINITY;
    return;
  }
  const double mx0 = s5[1] / n, my0 = s5[2] / n, mx1 = s5[3] / n, my1 = s5[4] / n;
  // pass 2: mean distance to the centroid
  double d2[2] = {0, 0};
  for (int i = tid; i < M; i += EM_THREADS) {
    const long long j = mm[i];
    if (j > -1 && j < N) {
      const double ax = p0[2 * i] - mx0, ay = p0[2 * i + 1] - my0, bx = p1[2 * j] - mx1, by = p1[2 * j + 1] - my1;
      d2[0] += sqrt(ax * ax + ay * ay);
      d2[1] += sqrt(bx * bx + by * by);
    }
  }
  block_sum_f64<2>(d2, red, tid);
  const double sc_a = 1.4142135623730951 / (d2[0] / n + 1e-8), sc_b = 1.4142135623730951 / (d2[1] / n + 1e-8);
  // pass 3: normal matrix, upper triangle row-major (r <= c)
  double acc[45];
#pragma unroll
  for (int q = 0; q < 45; ++q) acc[q] = 0.0;
  for (int i = tid; i < M; i += EM_THREADS) {
    const long long j = mm[i];
    if (j > -1 && j < N) {
      const double w = ss[i];
      const double x1 = sc_a * (p0[2 * i] - mx0), y1 = sc_a * (p0[2 * i + 1] - my0);
      const double x2 = sc_b * (p1[2 * j] - mx1), y2 = sc_b * (p1[2 * j + 1] - my1);
      const double rx[9] = {0, 0, 0, -x1, -y1, -1.0, y2 * x1, y2 * y1, y2};
      const double ry[9] = {x1, y1, 1.0, 0, 0, 0, -x2 * x1, -x2 * y1, -x2};
      int q = 0;
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int c = r; c < 9; ++c) acc[q++] += w * (rx[r] * rx[c] + ry[r] * ry[c]);
    }
  }
  block_sum_f64<45>(acc, red, tid);
  if (tid == 0) {
    int q = 0;
    for (int r = 0; r < 9; ++r)
      for (int c = r; c < 9; ++c) { A[r * 9 + c] = acc[q]; A[c * 9 + r] = acc[q]; ++q; }
    for (int r = 0; r < 81; ++r) V[r] = (r % 10 == 0) ? 1.0 : 0.0;
    // cyclic Jacobi: A <- J^T A J, V <- V J
    for (int sweep = 0; sweep < 60; ++sweep) {
      double off = 0.0, dg = 0.0;
      for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) { if (r == c) dg += A[r * 9 + c] * A[r * 9 + c]; else off += A[r * 9 + c] * A[r * 9 + c]; }
      if (!(off > 1e-40 * dg)) break;  // also leaves on NaN
      for (int p = 0; p < 8; ++p)
        for (int r = p + 1; r < 9; ++r) {
          const double apq = A[p * 9 + r];
          if (fabs(apq) < 1e-300) continue;
          const double theta = (A[r * 9 + r] - A[p * 9 + p]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
          for (int k = 0; k < 9; ++k) {  // columns p, r
            const double akp = A[k * 9 + p], akr = A[k * 9 + r];
            A[k * 9 + p] = c * akp - s * akr;
            A[k * 9 + r] = s * akp + c * akr;
          }
          for (int k = 0; k < 9; ++k) {  // rows p, r
            const double apk = A[p * 9 + k], ark = A[r * 9 + k];
            A[p * 9 + k] = c * apk - s * ark;
            A[r * 9 + k] = s * apk + c * ark;
          }
          for (int k = 0; k < 9; ++k) {
            const double vkp = V[k * 9 + p], vkr = V[k * 9 + r];
            V[k * 9 + p] = c * vkp - s * vkr;
            V[k * 9 + r] = s * vkp + c * vkr;
          }
        }
    }
    int best = 0;
    for (int r = 1; r < 9; ++r)
      if (A[r * 9 + r] < A[best * 9 + best]) best = r;
    double h[9];
    for (int r = 0; r < 9; ++r) h[r] = V[r * 9 + best];
    // H = T2^-1 (Hn T1), T = [[s,0,-s mx],[0,s,-s my],[0,0,1]]
    double g[9];
    for (int r = 0; r < 3; ++r) {
      g[r * 3 + 0] = h[r * 3 + 0] * sc_a;
      g[r * 3 + 1] = h[r * 3 + 1] * sc_a;
      g[r * 3 + 2] = -h[r * 3 + 0] * sc_a * mx0 - h[r * 3 + 1] * sc_a * my0 + h[r * 3 + 2];
    }
    double f[9];
    for (int c = 0; c < 3; ++c) {
      f[0 * 3 + c] = g[0 * 3 + c] / sc_b + mx1 * g[2 * 3 + c];
      f[1 * 3 + c] = g[1 * 3 + c] / sc_b + my1 * g[2 * 3 + c];
      f[2 * 3 + c] = g[2 * 3 + c];
    }
    const double den = f[8] + 1e-8;
    float Hf[9];
    bool finite = true;
    for (int r = 0; r < 9; ++r) { Hf[r] = (float)(f[r] / den); finite = finite && isfinite(Hf[r]); }
    float err = INFINITY;
    if (finite) {
      // homography_corner_error: corners (0,0) (W,0) (W,H) (0,H), plain division, mean distance, fp32 like the reference
      const float Wd = size0[b * 2], Hd = size0[b * 2 + 1];
      const float cx[4] = {0.f, Wd, Wd, 0.f}, cy[4] = {0.f, 0.f, Hd, Hd};
      float Hg[9];
      for (int r = 0; r < 9; ++r) Hg[r] = Hgt[b * 9 + r];
      float sum = 0.f;
      for (int k = 0; k < 4; ++k) {
        float ax, ay, gx, gy;
        warp_pt(Hf, cx[k], cy[k], 0.f, ax, ay);
        warp_pt(Hg, cx[k], cy[k], 0.f, gx, gy);
        sum += sqrtf((ax - gx) * (ax - gx) + (ay - gy) * (ay - gy));
      }
      err = sum / 4.f;
      if (!isfinite(err)) err = INFINITY;
    } else {
      for (int r = 0; r < 9; ++r) Hf[r] = INFINITY;
    }
    for (int r = 0; r < 9; ++r) Hout[b * 9 + r] = Hf[r];
    err_out[b] = err;
  }
}

extern "C" int gfc_eval_homography_dlt(const float* kp0, const float* kp1, const int64_t* m0, const float* scores0,
                                       const float* H_gt, const float* image_size0, int B, int M, int N, float* H_out,
                                       float* err_out, void* stream) {
  if (!kp0 || !kp1 || !m0 || !scores0 || !H_gt || !image_size0 || !H_out || !err_out || B <= 0 || M < 0 || N < 0)
    return GFC_ERR_INVALID;
  hipLaunchKernelGGL(dlt_kernel, dim3(B), dim3(EM_THREADS), 0, (hipStream_t)stream, kp0, kp1, (const long long*)m0,
                     scores0, H_gt, image_size0, M, N, H_out, err_out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
