// LightGlue: positional encoding, LayerNorm+GELU, matchability, dual log-softmax assignment and
// mutual-argmax match filtering.  HBM-bound row/column reductions on wavefronts.
#include "common.h"

// ------------------------------------------------------------------------------------------
// normalize_keypoints + LearnableFourierPositionalEncoding (lightglue.py:28-40,53-66).
// cos/sin [rows][64]: value of frequency f stored at 2f and 2f+1 (repeat_interleave(2)).
// ------------------------------------------------------------------------------------------
__global__ void posenc_kernel(const float* __restrict__ kpts, const float* __restrict__ scale_ori,
                              const float* __restrict__ sizes, const int* __restrict__ row0,
                              const int* __restrict__ nrows, const float* __restrict__ wr, int dim,
                              float* __restrict__ cos_out, float* __restrict__ sin_out, float* __restrict__ cs_out) {
  const int img = blockIdx.y;
  const int n = nrows[img];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // (row, freq)
  const int i = t >> 5, f = t & 31;
  if (i >= n) return;
  const size_t row = (size_t)row0[img] + i;
  const float sw = sizes[2 * img], sh = sizes[2 * img + 1];
  const float half_extent = fmaxf(sw, sh) / 2.f;
  const float kx = (kpts[2 * row] - sw / 2.f) / half_extent;
  const float ky = (kpts[2 * row + 1] - sh / 2.f) / half_extent;
  float p = kx * wr[dim * f];
  p += ky * wr[dim * f + 1];
  if (dim == 4) {  // add_scale_ori: [x, y, scale, orientation] (lightglue.py:436-453), scale / orientation un-normalised
    p += scale_ori[2 * row] * wr[4 * f + 2];
    p += scale_ori[2 * row + 1] * wr[4 * f + 3];
  }
  const float c = cosf(p), s = sinf(p);
  *reinterpret_cast<float2*>(cos_out + row * 64 + 2 * f) = make_float2(c, c);
  *reinterpret_cast<float2*>(sin_out + row * 64 + 2 * f) = make_float2(s, s);
  // packed table for the QKV GEMM's epilogue (library-internal): (cos, sin) of frequency f side by side, so that ONE
  // float4 holds everything the rotation of four consecutive channels needs
  if (cs_out) *reinterpret_cast<float2*>(cs_out + row * 64 + 2 * f) = make_float2(c, s);
}

int gfc_lg_posenc_packed(const float* kpts, const float* scale_ori, const float* sizes, const int32_t* row0,
                         const int32_t* n, int n_images, int max_n, const float* wr, int dim, float* cos_out, float* sin_out,
                         float* cs_out, void* stream) {
  if (!kpts || !sizes || !row0 || !n || !wr || !cos_out || !sin_out || n_images <= 0 || max_n <= 0) return GFC_ERR_INVALID;
  if ((dim != 2 && dim != 4) || ((dim == 4) != (scale_ori != nullptr))) return GFC_ERR_INVALID;
  dim3 grid((max_n * 32 + 255) / 256, n_images);
  hipLaunchKernelGGL(posenc_kernel, grid, dim3(256), 0, (hipStream_t)stream, kpts, scale_ori, sizes, row0, n, wr, dim,
                     cos_out, sin_out, cs_out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_lg_posenc(const float* kpts, const float* scale_ori, const float* sizes, const int32_t* row0,
                             const int32_t* n, int n_images, int max_n, const float* wr, int dim, float* cos_out,
                             float* sin_out, void* stream) {
  return gfc_lg_posenc_packed(kpts, scale_ori, sizes, row0, n, n_images, max_n, wr, dim, cos_out, sin_out, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// LayerNorm(512, eps 1e-5) + GELU(erf), in place; one wave per row (lightglue.py:143-148).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_gelu_kernel(float* __restrict__ x, int ld, int rows,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = x + (size_t)row * ld;
  float4 v[2];
  v[0] = *reinterpret_cast<const float4*>(p + lane * 4);
  v[1] = *reinterpret_cast<const float4*>(p + 256 + lane * 4);
  float s = (v[0].x + v[0].y) + (v[0].z + v[0].w) + (v[1].x + v[1].y) + (v[1].z + v[1].w);
  const float mean = wave_sum(s) * (1.f / 512.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += a * a + b * b + c * c + d * d;
  }
  const float var = wave_sum(q) * (1.f / 512.f);
  const float rstd = 1.f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c0 = i * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c0);
    const float4 bb = *reinterpret_cast<const float4*>(beta + c0);
    float y[4] = {(v[i].x - mean) * rstd * g.x + bb.x, (v[i].y - mean) * rstd * g.y + bb.y,
                  (v[i].z - mean) * rstd * g.z + bb.z, (v[i].w - mean) * rstd * g.w + bb.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = gfc_gelu(y[j]);
    *reinterpret_cast<float4*>(p + c0) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

extern "C" int gfc_layernorm_gelu(float* x, int ld, int rows, int width, const float* gamma, const float* beta,
                                  void* stream) {
  if (!x || !gamma || !beta || rows <= 0 || ld % 4) return GFC_ERR_INVALID;
  if (width != 512) return GFC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(layernorm_gelu_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, rows, gamma,
                     beta);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// matchability logit z[row] = x[row,:256] . w + b    (lightglue.py:286-287)
__global__ __launch_bounds__(256) void rowdot256_kernel(const float* __restrict__ x, int ld, int rows,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * ld + lane * 4);
  const float4 b = *reinterpret_cast<const float4*>(w + lane * 4);
  float s = wave_sum(a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w);
  if (lane == 0) z[row] = s + bias[0];
}

int gfc_rowdot256(const float* x, int ld, int rows, const float* w, const float* bias, float* z, hipStream_t st) {
  hipLaunchKernelGGL(rowdot256_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ld, rows, w, bias, z);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ------------------------------------------------------------------------------------------
// sigmoid_log_double_softmax (lightglue.py:257-269):
//   S[i,j] = log_softmax_j(sim)[i,j] + log_softmax_i(sim)[i,j] + logsig(z0_i) + logsig(z1_j)
//   S[i,N] = logsig(-z0_i),  S[M,j] = logsig(-z1_j),  S[M,N] = 0
// log_softmax(x) = (x - max) - log(sum(exp(x - max))).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float logsigmoid(float x) {
  // F.logsigmoid: min(x, 0) - log1p(exp(-|x|))
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// rows: one wave per row.  stats[0] = max, stats[1] = log(sum exp)
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ sim, long long stride_b, int ld,
                                                       int M, int N, float* __restrict__ rmax,
                                                       float* __restrict__ rlog) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= M) return;
  const float* p = sim + b * stride_b + (size_t)i * ld;
  float m = -INFINITY;
  for (int j = lane; j < N; j += 64) m = fmaxf(m, p[j]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < N; j += 64) s += expf(p[j] - m);
  s = wave_sum(s);
  if (lane == 0) {
    rmax[(size_t)b * M + i] = m;
    rlog[(size_t)b * M + i] = logf(s);
  }
}

// columns: block = 32 columns x 8 row groups, online (max, sum) per thread, LDS combine
__global__ __launch_bounds__(256) void lse_cols_kernel(const float* __restrict__ sim, long long stride_b, int ld,
                                                       int M, int N, float* __restrict__ cmax,
                                                       float* __restrict__ clog) {
  __shared__ float sm[8][32], ss[8][32];
  const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cx, b = blockIdx.y;
  const float* p = sim + b * stride_b;
  float m = -INFINITY, s = 0.f;
  if (j < N)
    for (int i = rg; i < M; i += 8) {
      float v = p[(size_t)i * ld + j];
      if (v > m) { s = s * expf(m - v) + 1.f; m = v; } else { s += expf(v - m); }
    }
  sm[rg][cx] = m;
  ss[rg][cx] = s;
  __syncthreads();
  if (rg == 0 && j < N) {
    float mm = -INFINITY;
    for (int g = 0; g < 8; ++g) mm = fmaxf(mm, sm[g][cx]);
    float t = 0.f;
    for (int g = 0; g < 8; ++g) t += (ss[g][cx] > 0.f) ? ss[g][cx] * expf(sm[g][cx] - mm) : 0.f;
    cmax[(size_t)b * N + j] = mm;
    clog[(size_t)b * N + j] = logf(t);
  }
}

__global__ __launch_bounds__(256) void assign_finalize_kernel(const float* __restrict__ sim, long long stride_sim,
                                                              int lds, const float* __restrict__ z0,
                                                              const float* __restrict__ z1, int M, int N,
                                                              const float* __restrict__ rmax,
                                                              const float* __restrict__ rlog,
                                                              const float* __restrict__ cmax,
                                                              const float* __restrict__ clog, float* __restrict__ out) {
  // grid: (ceil((N+1)/256), M+1, B).  out[b][i][j], ld = N+1.  May run in place (sim == out, lds == N+1):
  // every element is read and written by the same thread.
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j > N) return;
  float* o = out + ((size_t)b * (M + 1) + i) * (N + 1) + j;
  float v;
  if (i < M && j < N) {
    const float x = sim[b * stride_sim + (size_t)i * lds + j];
    const float s0 = (x - rmax[(size_t)b * M + i]) - rlog[(size_t)b * M + i];
    const float s1 = (x - cmax[(size_t)b * N + j]) - clog[(size_t)b * N + j];
    const float cert = logsigmoid(z0[(size_t)b * M + i]) + logsigmoid(z1[(size_t)b * N + j]);
    v = (s0 + s1) + cert;
  } else if (i < M) {
    v = logsigmoid(-z0[(size_t)b * M + i]);
  } else if (j < N) {
    v = logsigmoid(-z1[(size_t)b * N + j]);
  } else {
    v = 0.f;
  }
  *o = v;
}

// ------------------------------------------------------------------------------------------
// Two-pass tail of the assignment head (lightglue.py:257-269 + the arg-max half of :294-319): the
// [B,M,N] similarity block is read ONCE for the row and column soft-max statistics and once more for the final
// scores, which are written in place together with the row / column arg-max of the finished matrix.
// HBM traffic: 1 read + (1 read + 1 write) of the matrix; the five-pass form above (kept for
// gfc_lg_log_assignment / gfc_nn_match and behind GFC_ASSIGN_MODE=1) moved ~8.4x the matrix
// (profiles/r01_pmc_summary.json: strided column walks read it twice each).
//
// Workgroup = one band of AS_RB rows of one pair, 4 waves; a wave walks whole rows with its lanes along the
// columns (column j = c0 + lane + 64 q, q < AS_NC: contiguous 256-byte segments per load), so
//   * row statistics are lane-local + one wave reduction per row,
//   * column statistics live in registers (AS_NC running values per lane) across the rows of the wave, are merged
//     over the 4 waves in LDS and leave the workgroup as one partial per (band, column); a tiny kernel merges the
//     bands.  Columns beyond AS_NC*64 are handled in further chunks (row state carried in LDS).
// Tie breaking as torch.max on CPU: the lowest index among equal maxima.
// ------------------------------------------------------------------------------------------
__global__ void mutual_kernel(const float* __restrict__ max0, const int* __restrict__ i0, const int* __restrict__ i1,
                              int M, int N, float th, long long* __restrict__ m0, long long* __restrict__ m1,
                              float* __restrict__ ms0, float* __restrict__ ms1);
#define AS_RB 64  // rows per band (16 per wave)
#define AS_NC 16  // columns per lane per chunk (1024-column chunks)

__global__ __launch_bounds__(256) void assign_stats_kernel(const float* __restrict__ sim, long long stride_b, int ld,
                                                           int M, int N, float* __restrict__ rmax,
                                                           float* __restrict__ rlog, float* __restrict__ cpart) {
  __shared__ float cm[4][AS_NC * 64], cs[4][AS_NC * 64];
  __shared__ float row_m[AS_RB], row_s[AS_RB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int band = blockIdx.x, nbands = gridDim.x, b = blockIdx.y;
  const float* p = sim + b * stride_b;
  const int i0 = band * AS_RB;
  for (int c0 = 0; c0 < N; c0 += AS_NC * 64) {
    float m[AS_NC], sacc[AS_NC];
#pragma unroll
    for (int q = 0; q < AS_NC; ++q) { m[q] = -INFINITY; sacc[q] = 0.f; }
    for (int r = wave; r < AS_RB; r += 4) {  // wave-uniform
      const int i = i0 + r;
      if (i >= M) break;
      const float* row = p + (size_t)i * ld + c0 + lane;
      float v[AS_NC];
      float lm = -INFINITY;
#pragma unroll
      for (int q = 0; q < AS_NC; ++q) {
        v[q] = (c0 + lane + 64 * q < N) ? row[64 * q] : -INFINITY;
        lm = fmaxf(lm, v[q]);
      }
      const float old_m = c0 ? row_m[r] : -INFINITY;
      const float rm = fmaxf(wave_max(lm), old_m);
      float ls = 0.f;
#pragma unroll
      for (int q = 0; q < AS_NC; ++q) ls += (c0 + lane + 64 * q < N) ? expf(v[q] - rm) : 0.f;
      ls = wave_sum(ls);
      if (c0) ls += row_s[r] * expf(old_m - rm);
      if (lane == 0) { row_m[r] = rm; row_s[r] = ls; }
      // column statistics: online (max, sum of exp) per owned column
#pragma unroll
      for (int q = 0; q < AS_NC; ++q) {
        // one exponential, no divergent branch: exp(m - x) for x > m and exp(x - m) otherwise are both exp(-|x - m|)
        const float x = v[q];
        const bool up = x > m[q];
        const float e = expf(up ? m[q] - x : x - m[q]);
        sacc[q] = up ? sacc[q] * e + 1.f : sacc[q] + e;
        m[q] = up ? x : m[q];
      }
    }
#pragma unroll
    for (int q = 0; q < AS_NC; ++q) { cm[wave][lane + 64 * q] = m[q]; cs[wave][lane + 64 * q] = sacc[q]; }
    __syncthreads();
    for (int jj = threadIdx.x; jj < AS_NC * 64; jj += 256) {
      const int j = c0 + jj;
      if (j >= N) break;
      float mm = fmaxf(fmaxf(cm[0][jj], cm[1][jj]), fmaxf(cm[2][jj], cm[3][jj]));
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += (cs[w][jj] > 0.f) ? cs[w][jj] * expf(cm[w][jj] - mm) : 0.f;
      float* o = cpart + (((size_t)b * nbands + band) * N + j) * 2;
      o[0] = mm; o[1] = t;
    }
    __syncthreads();
  }
  for (int r = threadIdx.x; r < AS_RB; r += 256) {
    const int i = i0 + r;
    if (i < M) { rmax[(size_t)b * M + i] = row_m[r]; rlog[(size_t)b * M + i] = logf(row_s[r]); }
  }
}

// merge the band partials of the column statistics: cmax, clog [B][N]
__global__ void assign_colmerge_kernel(const float* __restrict__ cpart, int nbands, int N, float* __restrict__ cmax,
                                       float* __restrict__ clog) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j >= N) return;
  const float* pp = cpart + ((size_t)b * nbands * N + j) * 2;
  float mm = -INFINITY;
  for (int g = 0; g < nbands; ++g) mm = fmaxf(mm, pp[(size_t)g * N * 2]);
  float t = 0.f;
  for (int g = 0; g < nbands; ++g) {
    const float sg = pp[(size_t)g * N * 2 + 1];
    t += (sg > 0.f) ? sg * expf(pp[(size_t)g * N * 2] - mm) : 0.f;
  }
  cmax[(size_t)b * N + j] = mm;
  clog[(size_t)b * N + j] = logf(t);
}

// final scores in place (sc = [B][M+1][N+1], the inner block holds sim) + row arg-max (complete) and column arg-max
// (one partial per band) of the finished inner block
__global__ __launch_bounds__(256) void assign_finalize_argmax_kernel(
    float* __restrict__ sc, int M, int N, const float* __restrict__ z0, const float* __restrict__ z1,
    const float* __restrict__ rmax, const float* __restrict__ rlog, const float* __restrict__ cmax,
    const float* __restrict__ clog, float* __restrict__ row_best, int* __restrict__ row_arg,
    float* __restrict__ cpart_v, int* __restrict__ cpart_i) {
  __shared__ float cv[4][AS_NC * 64];
  __shared__ int ci[4][AS_NC * 64];
  __shared__ float row_v[AS_RB];
  __shared__ int row_i[AS_RB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int band = blockIdx.x, nbands = gridDim.x, b = blockIdx.y;
  const int ld = N + 1;
  float* p = sc + (size_t)b * (M + 1) * ld;
  const int i0 = band * AS_RB;
  for (int c0 = 0; c0 < N; c0 += AS_NC * 64) {
    float cmx[AS_NC], clg[AS_NC], lz1[AS_NC], bv[AS_NC];
    int bi[AS_NC];
#pragma unroll
    for (int q = 0; q < AS_NC; ++q) {
      const int j = c0 + lane + 64 * q;
      const bool ok = j < N;
      cmx[q] = ok ? cmax[(size_t)b * N + j] : 0.f;
      clg[q] = ok ? clog[(size_t)b * N + j] : 0.f;
      lz1[q] = ok ? logsigmoid(z1[(size_t)b * N + j]) : 0.f;
      bv[q] = -INFINITY;
      bi[q] = 0x7FFFFFFF;
    }
    // whole chunk inside the matrix (N = 1024: always): the 16 elements of a row are requested together and the NEXT
    // row's before this row is stored -- in the per-element form (load, wait, store; vmcnt retires in order and counts
    // stores) every load waited for the previous store's acknowledgement
    const bool full_chunk = c0 + AS_NC * 64 <= N;
    float xs[AS_NC], xn[AS_NC];
    if (full_chunk && i0 + wave < M) {
      const float* row = p + (size_t)(i0 + wave) * ld + c0 + lane;
#pragma unroll
      for (int q = 0; q < AS_NC; ++q) xs[q] = row[64 * q];
    }
    for (int r = wave; r < AS_RB; r += 4) {
      const int i = i0 + r;
      if (i >= M) break;
      const float rm = rmax[(size_t)b * M + i], rl = rlog[(size_t)b * M + i];
      const float zi = z0[(size_t)b * M + i];
      const float lz0 = logsigmoid(zi);
      float* row = p + (size_t)i * ld + c0 + lane;
      float best = -INFINITY;
      int arg = 0x7FFFFFFF;
      if (full_chunk) {
        if (r + 4 < AS_RB && i + 4 < M) {
#pragma unroll
          for (int q = 0; q < AS_NC; ++q) xn[q] = row[(size_t)4 * ld + 64 * q];
        }
#pragma unroll
        for (int q = 0; q < AS_NC; ++q) {
          const int j = c0 + lane + 64 * q;
          const float x = xs[q];
          const float s0 = (x - rm) - rl;
          const float s1 = (x - cmx[q]) - clg[q];
          const float cert = lz0 + lz1[q];
          const float v = (s0 + s1) + cert;
          row[64 * q] = v;
          if (v > best || arg == 0x7FFFFFFF) { best = v; arg = j; }
          if (v > bv[q] || bi[q] == 0x7FFFFFFF) { bv[q] = v; bi[q] = i; }
        }
#pragma unroll
        for (int q = 0; q < AS_NC; ++q) xs[q] = xn[q];
      } else
#pragma unroll
      for (int q = 0; q < AS_NC; ++q) {
        const int j = c0 + lane + 64 * q;
        if (j < N) {
          const float x = row[64 * q];
          const float s0 = (x - rm) - rl;
          const float s1 = (x - cmx[q]) - clg[q];
          const float cert = lz0 + lz1[q];
          const float v = (s0 + s1) + cert;
          row[64 * q] = v;
          if (v > best || arg == 0x7FFFFFFF) { best = v; arg = j; }
          if (v > bv[q] || bi[q] == 0x7FFFFFFF) { bv[q] = v; bi[q] = i; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(arg, o);
        if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
      }
      if (lane == 0) {
        if (c0 == 0 || best > row_v[r]) { row_v[r] = best; row_i[r] = arg; }  // earlier chunks hold the lower indices
        if (c0 + AS_NC * 64 >= N) p[(size_t)i * ld + N] = logsigmoid(-zi);     // dustbin column
      }
    }
#pragma unroll
    for (int q = 0; q < AS_NC; ++q) { cv[wave][lane + 64 * q] = bv[q]; ci[wave][lane + 64 * q] = bi[q]; }
    __syncthreads();
    for (int jj = threadIdx.x; jj < AS_NC * 64; jj += 256) {
      const int j = c0 + jj;
      if (j >= N) break;
      float v = cv[0][jj];
      int ix = ci[0][jj];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float ov = cv[w][jj];
        const int oi = ci[w][jj];
        if (oi != 0x7FFFFFFF && (ix == 0x7FFFFFFF || ov > v || (ov == v && oi < ix))) { v = ov; ix = oi; }
      }
      const size_t o = ((size_t)b * nbands + band) * N + j;
      cpart_v[o] = v;
      cpart_i[o] = ix;
      if (band == nbands - 1) p[(size_t)M * ld + j] = logsigmoid(-z1[(size_t)b * N + j]);  // dustbin row
    }
    __syncthreads();
  }
  for (int r = threadIdx.x; r < AS_RB; r += 256) {
    const int i = i0 + r;
    if (i < M) { row_best[(size_t)b * M + i] = row_v[r]; row_arg[(size_t)b * M + i] = row_i[r]; }
  }
  if (band == nbands - 1 && threadIdx.x == 0) p[(size_t)M * ld + N] = 0.f;
}

// column arg-max: merge of the band partials (bands ascend with the row index: strictly greater wins)
__global__ void assign_colarg_merge_kernel(const float* __restrict__ cpart_v, const int* __restrict__ cpart_i, int nbands,
                                           int N, int* __restrict__ col_arg) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j >= N) return;
  const size_t base = (size_t)b * nbands * N + j;
  float v = cpart_v[base];
  int ix = cpart_i[base];
  for (int g = 1; g < nbands; ++g) {
    const float ov = cpart_v[base + (size_t)g * N];
    const int oi = cpart_i[base + (size_t)g * N];
    if (oi != 0x7FFFFFFF && (ix == 0x7FFFFFFF || ov > v)) { v = ov; ix = oi; }
  }
  col_arg[(size_t)b * N + j] = ix;
}

// In-place variant used by gfc_lg_forward: sim was written by the GEMM straight into the inner block of
// the [M+1][N+1] output (ld = N+1).
// scratch of the two-pass tail behind `stats`: column partials of both passes, row / column arg-max
size_t gfc_assign_tail_bytes(int B, int M, int N) {
  const size_t nb = (size_t)(M + AS_RB - 1) / AS_RB;
  return gfc_align((size_t)B * nb * N * 2 * 4) + 2 * gfc_align((size_t)B * nb * N * 4) + gfc_align((size_t)B * M * 8) +
         gfc_align((size_t)B * N * 4);
}

// sim (inner block of scores, ld = N+1) -> final scores in place + filter_matches, two sweeps over the matrix.
// stats: 2*B*(M+N) floats; tail: gfc_assign_tail_bytes().
int gfc_assign_filter_fused(float* scores, const float* z0, const float* z1, int B, int M, int N, float threshold,
                            int64_t* m0, int64_t* m1, float* ms0, float* ms1, float* stats, void* tail,
                            hipStream_t st) {
  float* rmax = stats;
  float* rlog = rmax + (size_t)B * M;
  float* cmax = rlog + (size_t)B * M;
  float* clog = cmax + (size_t)B * N;
  const int nb = (M + AS_RB - 1) / AS_RB;
  char* t = (char*)tail;
  float* cpart = (float*)t;                t += gfc_align((size_t)B * nb * N * 2 * 4);
  float* cpv = (float*)t;                  t += gfc_align((size_t)B * nb * N * 4);
  int* cpi = (int*)t;                      t += gfc_align((size_t)B * nb * N * 4);
  float* rbest = (float*)t;
  int* rarg = (int*)(rbest + (size_t)B * M); t += gfc_align((size_t)B * M * 8);
  int* carg = (int*)t;
  const long long sb = (long long)(M + 1) * (N + 1);
  hipLaunchKernelGGL(assign_stats_kernel, dim3(nb, B), dim3(256), 0, st, scores, sb, N + 1, M, N, rmax, rlog, cpart);
  hipLaunchKernelGGL(assign_colmerge_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, cpart, nb, N, cmax, clog);
  hipLaunchKernelGGL(assign_finalize_argmax_kernel, dim3(nb, B), dim3(256), 0, st, scores, M, N, z0, z1, rmax, rlog,
                     cmax, clog, rbest, rarg, cpv, cpi);
  hipLaunchKernelGGL(assign_colarg_merge_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, cpv, cpi, nb, N, carg);
  const int mn = M > N ? M : N;
  hipLaunchKernelGGL(mutual_kernel, dim3((mn + 255) / 256, B), dim3(256), 0, st, rbest, rarg, carg, M, N, threshold,
                     (long long*)m0, (long long*)m1, ms0, ms1);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

int gfc_assign_inplace(float* scores, const float* z0, const float* z1, int B, int M, int N, float* stats,
                       hipStream_t st) {
  float* rmax = stats;
  float* rlog = rmax + (size_t)B * M;
  float* cmax = rlog + (size_t)B * M;
  float* clog = cmax + (size_t)B * N;
  const long long sb = (long long)(M + 1) * (N + 1);
  hipLaunchKernelGGL(lse_rows_kernel, dim3((M + 3) / 4, B), dim3(256), 0, st, scores, sb, N + 1, M, N, rmax, rlog);
  hipLaunchKernelGGL(lse_cols_kernel, dim3((N + 31) / 32, B), dim3(256), 0, st, scores, sb, N + 1, M, N, cmax, clog);
  hipLaunchKernelGGL(assign_finalize_kernel, dim3((N + 1 + 255) / 256, M + 1, B), dim3(256), 0, st, scores, sb, N + 1,
                     z0, z1, M, N, rmax, rlog, cmax, clog, scores);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_lg_log_assignment(const float* sim, const float* z0, const float* z1, int B, int M, int N,
                                     float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!sim || !z0 || !z1 || !out || !ws || B <= 0 || M <= 0 || N <= 0) return GFC_ERR_INVALID;
  if (ws_bytes < (size_t)2 * B * (M + N) * sizeof(float)) return GFC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* rmax = (float*)ws;
  float* rlog = rmax + (size_t)B * M;
  float* cmax = rlog + (size_t)B * M;
  float* clog = cmax + (size_t)B * N;
  const long long sb = (long long)M * N;
  hipLaunchKernelGGL(lse_rows_kernel, dim3((M + 3) / 4, B), dim3(256), 0, st, sim, sb, N, M, N, rmax, rlog);
  hipLaunchKernelGGL(lse_cols_kernel, dim3((N + 31) / 32, B), dim3(256), 0, st, sim, sb, N, M, N, cmax, clog);
  hipLaunchKernelGGL(assign_finalize_kernel, dim3((N + 1 + 255) / 256, M + 1, B), dim3(256), 0, st, sim, sb, N, z0, z1,
                     M, N, rmax, rlog, cmax, clog, out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ------------------------------------------------------------------------------------------
// filter_matches (lightglue.py:294-319).  Integer outputs are bit-exact: max / argmax over the
// inner [M,N] block with first-index tie break (torch.max on CPU), mutual check by gather.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowargmax_kernel(const float* __restrict__ sc, int M, int N,
                                                        float* __restrict__ vmax, int* __restrict__ imax) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= M) return;
  const float* p = sc + ((size_t)b * (M + 1) + i) * (N + 1);
  float m = -INFINITY;
  int idx = 0x7FFFFFFF;
  for (int j = lane; j < N; j += 64) {
    float v = p[j];
    if (v > m || idx == 0x7FFFFFFF) { m = v; idx = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(m, o);
    int oi = __shfl_xor(idx, o);
    if (om > m || (om == m && oi < idx)) { m = om; idx = oi; }
  }
  if (lane == 0) { vmax[(size_t)b * M + i] = m; imax[(size_t)b * M + i] = idx; }
}

__global__ __launch_bounds__(256) void colargmax_kernel(const float* __restrict__ sc, int M, int N,
                                                        int* __restrict__ imax) {
  __shared__ float sm[8][32];
  __shared__ int si[8][32];
  const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cx, b = blockIdx.y;
  const float* p = sc + (size_t)b * (M + 1) * (N + 1);
  float m = -INFINITY;
  int idx = 0x7FFFFFFF;
  if (j < N)
    for (int i = rg; i < M; i += 8) {
      float v = p[(size_t)i * (N + 1) + j];
      if (v > m || idx == 0x7FFFFFFF) { m = v; idx = i; }
    }
  sm[rg][cx] = m;
  si[rg][cx] = idx;
  __syncthreads();
  if (rg == 0 && j < N) {
    for (int g = 1; g < 8; ++g) {
      float om = sm[g][cx];
      int oi = si[g][cx];
      if (om > m || (om == m && oi < idx)) { m = om; idx = oi; }
    }
    imax[(size_t)b * N + j] = idx;
  }
}

__global__ void mutual_kernel(const float* __restrict__ max0, const int* __restrict__ i0, const int* __restrict__ i1,
                              int M, int N, float th, long long* __restrict__ m0, long long* __restrict__ m1,
                              float* __restrict__ ms0, float* __restrict__ ms1) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int* a0 = i0 + (size_t)b * M;
  const int* a1 = i1 + (size_t)b * N;
  const float* mx = max0 + (size_t)b * M;
  if (t < M) {
    const int j = a0[t];
    const bool mutual = a1[j] == t;
    const float s = mutual ? expf(mx[t]) : 0.f;
    ms0[(size_t)b * M + t] = s;
    m0[(size_t)b * M + t] = (mutual && s > th) ? (long long)j : -1ll;
  }
  if (t < N) {
    const int i = a1[t];
    const bool mutual1 = a0[i] == t;
    const bool mutual0 = a1[a0[i]] == i;
    const float s0 = mutual0 ? expf(mx[i]) : 0.f;
    ms1[(size_t)b * N + t] = mutual1 ? s0 : 0.f;
    m1[(size_t)b * N + t] = (mutual1 && mutual0 && s0 > th) ? (long long)i : -1ll;
  }
}

extern "C" int gfc_lg_filter_matches(const float* scores, int B, int M, int N, float threshold, int64_t* m0,
                                     int64_t* m1, float* ms0, float* ms1, void* ws, size_t ws_bytes, void* stream) {
  if (!scores || !m0 || !m1 || !ms0 || !ms1 || !ws || B <= 0 || M <= 0 || N <= 0) return GFC_ERR_INVALID;
  if (ws_bytes < (size_t)B * (M + N) * 8) return GFC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* max0 = (float*)ws;
  int* i0 = (int*)(max0 + (size_t)B * M);
  int* i1 = i0 + (size_t)B * M;
  hipLaunchKernelGGL(rowargmax_kernel, dim3((M + 3) / 4, B), dim3(256), 0, st, scores, M, N, max0, i0);
  hipLaunchKernelGGL(colargmax_kernel, dim3((N + 31) / 32, B), dim3(256), 0, st, scores, M, N, i1);
  const int mn = M > N ? M : N;
  hipLaunchKernelGGL(mutual_kernel, dim3((mn + 255) / 256, B), dim3(256), 0, st, max0, i0, i1, M, N, threshold,
                     (long long*)m0, (long long*)m1, ms0, ms1);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ------------------------------------------------------------------------------------------
// Nearest-neighbour matcher (reference gluefactory/models/matchers/nearest_neighbor_matcher.py:15-79):
// top-2 similarities per row / column, ratio and distance tests on d = 2(1 - sim), mutual check.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void top2_insert(float v, int j, float& b1, int& i1, float& b2) {
  if (v > b1 || (v == b1 && j < i1)) { b2 = b1; b1 = v; i1 = j; } else if (v > b2) { b2 = v; }
}

// rows: one wave per row -> (best, argbest, second best)
__global__ __launch_bounds__(256) void nn_top2_rows_kernel(const float* __restrict__ sim, int M, int N,
                                                           float* __restrict__ best, int* __restrict__ arg,
                                                           float* __restrict__ second) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= M) return;
  const float* p = sim + ((size_t)b * M + i) * N;
  float b1 = -INFINITY, b2 = -INFINITY;
  int i1 = 0x7FFFFFFF;
  for (int j = lane; j < N; j += 64) top2_insert(p[j], j, b1, i1, b2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob1 = __shfl_xor(b1, o), ob2 = __shfl_xor(b2, o);
    const int oi1 = __shfl_xor(i1, o);
    // merge two (best, second) pairs
    if (ob1 > b1 || (ob1 == b1 && oi1 < i1)) { b2 = fmaxf(b1, ob2); b1 = ob1; i1 = oi1; } else { b2 = fmaxf(b2, ob1); }
  }
  if (lane == 0) { best[(size_t)b * M + i] = b1; arg[(size_t)b * M + i] = i1; second[(size_t)b * M + i] = b2; }
}

// columns: 32 columns x 8 row groups per block
__global__ __launch_bounds__(256) void nn_top2_cols_kernel(const float* __restrict__ sim, int M, int N,
                                                           float* __restrict__ best, int* __restrict__ arg,
                                                           float* __restrict__ second) {
  __shared__ float s1[8][32], s2[8][32];
  __shared__ int si[8][32];
  const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cx, b = blockIdx.y;
  const float* p = sim + (size_t)b * M * N;
  float b1 = -INFINITY, b2 = -INFINITY;
  int i1 = 0x7FFFFFFF;
  if (j < N)
    for (int i = rg; i < M; i += 8) top2_insert(p[(size_t)i * N + j], i, b1, i1, b2);
  s1[rg][cx] = b1; s2[rg][cx] = b2; si[rg][cx] = i1;
  __syncthreads();
  if (rg == 0 && j < N) {
    for (int g = 1; g < 8; ++g) {
      const float ob1 = s1[g][cx], ob2 = s2[g][cx];
      const int oi1 = si[g][cx];
      if (ob1 > b1 || (ob1 == b1 && oi1 < i1)) { b2 = fmaxf(b1, ob2); b1 = ob1; i1 = oi1; } else { b2 = fmaxf(b2, ob1); }
    }
    best[(size_t)b * N + j] = b1; arg[(size_t)b * N + j] = i1; second[(size_t)b * N + j] = b2;
  }
}

__device__ __forceinline__ int nn_accept(float s1, float s2, int idx, int ncand, float ratio, float dist_th) {
  // find_nn, nearest_neighbor_matcher.py:15-31
  if (ncand == 0) return -1;
  const float d1 = 2.f * (1.f - s1), d2 = 2.f * (1.f - s2);
  bool ok = true;
  if (ratio > 0.f && ncand > 1) ok = ok && (d1 <= (ratio * ratio) * d2);
  if (dist_th > 0.f) ok = ok && (d1 <= dist_th * dist_th);
  return ok ? idx : -1;
}

__global__ void nn_match_kernel(const float* __restrict__ rb, const int* __restrict__ ra, const float* __restrict__ rs,
                                const float* __restrict__ cb, const int* __restrict__ ca, const float* __restrict__ cs,
                                int M, int N, float ratio, float dist_th, int mutual, long long* __restrict__ m0,
                                long long* __restrict__ m1, float* __restrict__ ms0, float* __restrict__ ms1) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  rb += (size_t)b * M; ra += (size_t)b * M; rs += (size_t)b * M;
  cb += (size_t)b * N; ca += (size_t)b * N; cs += (size_t)b * N;
  if (t < M) {
    int a = nn_accept(rb[t], rs[t], ra[t], N, ratio, dist_th);
    if (mutual && a > -1) {
      const int back = nn_accept(cb[a], cs[a], ca[a], M, ratio, dist_th);
      if (back != t) a = -1;
    }
    m0[(size_t)b * M + t] = a;
    ms0[(size_t)b * M + t] = a > -1 ? 1.f : 0.f;
  }
  if (t < N) {
    int a = nn_accept(cb[t], cs[t], ca[t], M, ratio, dist_th);
    if (mutual && a > -1) {
      const int back = nn_accept(rb[a], rs[a], ra[a], N, ratio, dist_th);
      if (back != t) a = -1;
    }
    m1[(size_t)b * N + t] = a;
    ms1[(size_t)b * N + t] = a > -1 ? 1.f : 0.f;
  }
}

// la[:, :M, :N] = log_softmax(sim, -1) + log_softmax(sim, -2); last row / column zero
__global__ __launch_bounds__(256) void nn_log_assignment_kernel(const float* __restrict__ sim, int M, int N,
                                                                const float* __restrict__ rmax,
                                                                const float* __restrict__ rlog,
                                                                const float* __restrict__ cmax,
                                                                const float* __restrict__ clog,
                                                                float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j > N) return;
  float v = 0.f;
  if (i < M && j < N) {
    const float x = sim[((size_t)b * M + i) * N + j];
    v = ((x - rmax[(size_t)b * M + i]) - rlog[(size_t)b * M + i]) +
        ((x - cmax[(size_t)b * N + j]) - clog[(size_t)b * N + j]);
  }
  out[((size_t)b * (M + 1) + i) * (N + 1) + j] = v;
}

extern "C" size_t gfc_nn_workspace_bytes(int B, int M, int N) {
  if (B <= 0 || M <= 0 || N <= 0) return 0;
  return gfc_align((size_t)B * (M + N) * 5 * 4);
}

extern "C" int gfc_nn_match(const float* desc0, const float* desc1, int B, int M, int N, int D, float ratio_thresh,
                            float distance_thresh, int mutual, int64_t* m0, int64_t* m1, float* ms0, float* ms1,
                            float* sim, float* log_assignment, void* ws, size_t ws_bytes, void* stream) {
  if (!desc0 || !desc1 || !m0 || !m1 || !ms0 || !ms1 || !sim || !ws || B <= 0 || M <= 0 || N <= 0) return GFC_ERR_INVALID;
  if (D <= 0 || D % 32) return GFC_ERR_UNSUPPORTED;
  if (ws_bytes < gfc_nn_workspace_bytes(B, M, N)) return GFC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int s = gfc_batched_nt(desc0, D, (long long)M * D, desc1, D, (long long)N * D, sim, N, (long long)M * N, M, N, D, B,
                         st);
  if (s != GFC_OK) return s;
  float* rb = (float*)ws;
  float* rs = rb + (size_t)B * M;
  int* ra = (int*)(rs + (size_t)B * M);
  float* cb = (float*)(ra + (size_t)B * M);
  float* cs = cb + (size_t)B * N;
  int* ca = (int*)(cs + (size_t)B * N);
  float* stats = (float*)(ca + (size_t)B * N);  // 2*(M+N) floats per batch entry
  hipLaunchKernelGGL(nn_top2_rows_kernel, dim3((M + 3) / 4, B), dim3(256), 0, st, sim, M, N, rb, ra, rs);
  hipLaunchKernelGGL(nn_top2_cols_kernel, dim3((N + 31) / 32, B), dim3(256), 0, st, sim, M, N, cb, ca, cs);
  const int mn = M > N ? M : N;
  hipLaunchKernelGGL(nn_match_kernel, dim3((mn + 255) / 256, B), dim3(256), 0, st, rb, ra, rs, cb, ca, cs, M, N,
                     ratio_thresh, distance_thresh, mutual, (long long*)m0, (long long*)m1, ms0, ms1);
  if (log_assignment) {
    float* rmax = stats;
    float* rlog = rmax + (size_t)B * M;
    float* cmax = rlog + (size_t)B * M;
    float* clog = cmax + (size_t)B * N;
    hipLaunchKernelGGL(lse_rows_kernel, dim3((M + 3) / 4, B), dim3(256), 0, st, sim, (long long)M * N, N, M, N, rmax,
                       rlog);
    hipLaunchKernelGGL(lse_cols_kernel, dim3((N + 31) / 32, B), dim3(256), 0, st, sim, (long long)M * N, N, M, N, cmax,
                       clog);
    hipLaunchKernelGGL(nn_log_assignment_kernel, dim3((N + 1 + 255) / 256, M + 1, B), dim3(256), 0, st, sim, M, N,
                       rmax, rlog, cmax, clog, log_assignment);
  }
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
