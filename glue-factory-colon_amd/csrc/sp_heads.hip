// SuperPoint detector head, fused: 1x1 convolution 256 -> 65 (+ BN affine in the open variant), softmax over the 65
// logits of every 8x8 cell, dustbin dropped, depth-to-space -- reference
// gluefactory/models/extractors/superpoint_open.py:111-114,138-144 and gluefactory_nonfree/superpoint.py:193-194,229-235:
//     S[b, 8y+i, 8x+j] = softmax_c( Wp . hidden[b, y, x, :] + bp )[8i+j]
// Until round 4 this was a GEMM launch (N = 65 padded to a 128-column tile: half of its MFMAs multiplied padding)
// writing [rows, 65] logits to HBM plus softmax_d2s_kernel reading them back (158 MB round trip
// per 64 VGA images).  Here one workgroup owns 128 cells: the 64 heat-map channels on the matrix pipe (two 32-column
// tiles), the dustbin channel as one fmaf chain per cell on the VALU, the logits tile in LDS, statistics and outputs
// with the arithmetic of the softmax_d2s_kernel of rounds 1-4 (max, then the sum of expf(l - max) over c = 0..64 in order).
//
// Built to give the bits of the two-launch form, which was measured identical on the round-5 workloads when that form was
// retired and no longer exists to compare with; the test (test_detector_head_fused_softmax_d2s) holds the kernel to
// 1e-7 and equal per-cell arg-max against an fp32 restatement.  The construction: the K loop feeds the MFMAs exactly like gemm_nt_kernel (lane half h supplies
// k = 8g + 4h + s in step s of k group g), v_mfma_f32_32x32x2_f32 accumulates its two products in k order with one
// rounding each (the conv1a-on-the-matrix-pipe finding of round 3), so the dustbin's VALU chain
// fmaf(a[8g+s], w[8g+s], .) then fmaf(a[8g+4+s], w[8g+4+s], .) reproduces the value the GEMM's third column tile
// computed; the epilogue is ((acc + bias) * scale + shift), the soft-max sums exp(l - max) over c = 0..64 in order.
#include "common.h"

#define DH_ROWS 128                 // cells per workgroup
#define DH_BK 32                    // K tile
#define DH_LD (DH_BK + 4)           // LDS pitch of the A / B tiles
#define DH_K 256
#define DH_TILE ((DH_ROWS + 64) * DH_LD)  // floats per staged K tile: A [128][36] then B [64][36]
#define DH_LP 65                    // pitch of the logits tile

__global__ __launch_bounds__(256, 2) void det_head_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long rows, int h8, int w8,
                                                          float* __restrict__ heat) {
  __shared__ __attribute__((aligned(16))) float smem[2 * DH_TILE + DH_K + 2 * DH_ROWS];
  __shared__ long long cbase[DH_ROWS];  // per cell: offset of its 8x8 block in the heat-map (-1: beyond the last cell)
  float* w64 = smem + 2 * DH_TILE;   // dustbin filter [256]
  float* mx = w64 + DH_K;            // per cell: max, sum
  float* sinv = mx + DH_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const long long row0 = (long long)blockIdx.x * DH_ROWS;

  w64[tid] = Wp[(size_t)64 * DH_K + tid];

  // staging: A tile 128 x 32 floats = 1024 float4 (4 per thread), B tile 64 x 32 = 512 float4 (2 per thread)
  const int s_r = tid >> 3, s_c4 = (tid & 7) * 4;  // rows s_r + 32 i
  const float* ap[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long r = row0 + s_r + 32 * i;
    if (r > rows - 1) r = rows - 1;  // tail cells: a valid row, results dropped
    ap[i] = A + (size_t)r * lda + s_c4;
  }
  const float* wp0 = Wp + (size_t)s_r * DH_K + s_c4;
  const float* wp1 = Wp + (size_t)(s_r + 32) * DH_K + s_c4;
  float4 ar0, ar1, ar2, ar3, wr0, wr1;
#define DH_LOAD(kt_)                                                       \
  do {                                                                     \
    const int k0_ = (kt_) * DH_BK;                                         \
    ar0 = *reinterpret_cast<const float4*>(ap[0] + k0_);                   \
    ar1 = *reinterpret_cast<const float4*>(ap[1] + k0_);                   \
    ar2 = *reinterpret_cast<const float4*>(ap[2] + k0_);                   \
    ar3 = *reinterpret_cast<const float4*>(ap[3] + k0_);                   \
    wr0 = *reinterpret_cast<const float4*>(wp0 + k0_);                     \
    wr1 = *reinterpret_cast<const float4*>(wp1 + k0_);                     \
  } while (0)
#define DH_STORE(buf_)                                                     \
  do {                                                                     \
    float* as_ = smem + (buf_) * DH_TILE + s_r * DH_LD + s_c4;             \
    float* bs_ = smem + (buf_) * DH_TILE + DH_ROWS * DH_LD + s_r * DH_LD + s_c4; \
    *reinterpret_cast<float4*>(as_) = ar0;                                 \
    *reinterpret_cast<float4*>(as_ + 32 * DH_LD) = ar1;                    \
    *reinterpret_cast<float4*>(as_ + 64 * DH_LD) = ar2;                    \
    *reinterpret_cast<float4*>(as_ + 96 * DH_LD) = ar3;                    \
    *reinterpret_cast<float4*>(bs_) = wr0;                                 \
    *reinterpret_cast<float4*>(bs_ + 32 * DH_LD) = wr1;                    \
  } while (0)

  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  float dust = 0.f;  // threads 0..127: the dustbin logit of cell tid
  const int a_off = (wave * 32 + l31) * DH_LD + 4 * h;
  const int b_off = DH_ROWS * DH_LD + l31 * DH_LD + 4 * h;
  constexpr int KT = DH_K / DH_BK;
  DH_LOAD(0);
  DH_STORE(0);
  DH_LOAD(1);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) {
      DH_STORE((kt + 1) & 1);
      if (kt + 2 < KT) DH_LOAD(kt + 2);
    }
    const float* tile = smem + (kt & 1) * DH_TILE;
#pragma unroll
    for (int gk = 0; gk < DH_BK / 8; ++gk) {
      const float4 af = *reinterpret_cast<const float4*>(tile + a_off + 8 * gk);
      const float4 b0 = *reinterpret_cast<const float4*>(tile + b_off + 8 * gk);
      const float4 b1 = *reinterpret_cast<const float4*>(tile + b_off + 32 * DH_LD + 8 * gk);
      acc[0] = mfma32(af.x, b0.x, acc[0]);
      acc[1] = mfma32(af.x, b1.x, acc[1]);
      acc[0] = mfma32(af.y, b0.y, acc[0]);
      acc[1] = mfma32(af.y, b1.y, acc[1]);
      acc[0] = mfma32(af.z, b0.z, acc[0]);
      acc[1] = mfma32(af.z, b1.z, acc[1]);
      acc[0] = mfma32(af.w, b0.w, acc[0]);
      acc[1] = mfma32(af.w, b1.w, acc[1]);
    }
    if (tid < DH_ROWS) {
      // the dustbin channel of cell tid, in the matrix pipe's order: k = 8g + s, then 8g + 4 + s
      const float* arow = tile + tid * DH_LD;
      const float* wk = w64 + kt * DH_BK;
#pragma unroll
      for (int gk = 0; gk < DH_BK / 8; ++gk) {
        const float4 lo = *reinterpret_cast<const float4*>(arow + 8 * gk), hi = *reinterpret_cast<const float4*>(arow + 8 * gk + 4);
        const float4 wl = *reinterpret_cast<const float4*>(wk + 8 * gk), wh = *reinterpret_cast<const float4*>(wk + 8 * gk + 4);
        dust = fmaf(lo.x, wl.x, dust); dust = fmaf(hi.x, wh.x, dust);
        dust = fmaf(lo.y, wl.y, dust); dust = fmaf(hi.y, wh.y, dust);
        dust = fmaf(lo.z, wl.z, dust); dust = fmaf(hi.z, wh.z, dust);
        dust = fmaf(lo.w, wl.w, dust); dust = fmaf(hi.w, wh.w, dust);
      }
    }
    __syncthreads();
  }
#undef DH_LOAD
#undef DH_STORE

  // ---- logits tile L[cell][65] in LDS (the staging buffers are free after the loop's last barrier) ----
  float* L = smem;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ch = nt * 32 + l31;
    const float bi = bias[ch], sc = scale ? scale[ch] : 1.f, sh = shift ? shift[ch] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) L[(wave * 32 + acc_row(r, h)) * DH_LP + ch] = (acc[nt][r] + bi) * sc + sh;
  }
  if (tid < DH_ROWS) L[tid * DH_LP + 64] = (dust + bias[64]) * (scale ? scale[64] : 1.f) + (shift ? shift[64] : 0.f);
  __syncthreads();
  // statistics with the arithmetic of softmax_d2s_kernel -- max over the 65 logits, then the sum of expf(l - max) over
  // c = 0..64 IN ORDER -- split so that every thread works: the maximum is order-independent (two threads per cell), the 65
  // exponentials of a cell are evaluated once by all threads and kept in the tile (the outputs below are these same values
  // divided by the sum: expf is not evaluated a second time), only the ordered sum is one thread per cell
  {
    const int q = tid >> 1, part = tid & 1;
    float m = -INFINITY;
    for (int c = part * 33; c < (part ? 65 : 33); ++c) m = fmaxf(m, L[q * DH_LP + c]);
    m = fmaxf(m, __shfl_xor(m, 1));
    if (part == 0) mx[q] = m;
  }
  __syncthreads();
  for (int idx = tid; idx < DH_ROWS * 65; idx += 256) {
    const int q = idx / 65;
    L[idx] = expf(L[idx] - mx[q]);
  }
  __syncthreads();
  if (tid < DH_ROWS) {
    float s = 0.f;
    for (int c = 0; c < 65; ++c) s += L[tid * DH_LP + c];
    sinv[tid] = s;
    const long long cell = row0 + tid, per_img = (long long)h8 * w8;
    const long long b = cell / per_img;
    const int t = (int)(cell - b * per_img), y = t / w8, x = t - y * w8;
    cbase[tid] = cell < rows ? ((b * h8 * 8 + (long long)y * 8) * (w8 * 8) + x * 8) : -1;
  }
  __syncthreads();
  // ---- depth-to-space: heat[b, 8y + i, 8x + j] = exp(l[cell][8i + j] - max) / sum; consecutive threads walk (cell, j) ----
  const int W8 = w8 * 8;
#pragma unroll 4
  for (int idx = tid; idx < 8 * DH_ROWS * 8; idx += 256) {
    const int i = idx >> 10, q = (idx >> 3) & (DH_ROWS - 1), j = idx & 7;
    const long long base = cbase[q];
    if (base >= 0) heat[base + (long long)i * W8 + j] = L[q * DH_LP + 8 * i + j] / sinv[q];
  }
}

extern "C" int gfc_sp_detector_head(const float* hidden, int lda, const float* w, const float* bias, const float* scale,
                                    const float* shift, int B, int h8, int w8, float* heat, void* stream);

// hidden: [rows][lda] with the detector's 256 hidden channels first; Wp [65][256]; bias / scale / shift [65] (scale,
// shift nullable together); heat [B][8 h8][8 w8]
int gfc_det_head_softmax_d2s(const float* hidden, int lda, const float* wp, const float* bias, const float* scale,
                             const float* shift, int B, int h8, int w8, float* heat, hipStream_t st) {
  if (!hidden || !wp || !bias || !heat || B <= 0 || h8 <= 0 || w8 <= 0 || lda < DH_K || lda % 4) return GFC_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return GFC_ERR_INVALID;
  // float4 loads of both operands: 16-byte aligned bases (lda % 4 keeps every row aligned)
  if ((reinterpret_cast<size_t>(hidden) | reinterpret_cast<size_t>(wp)) & 15) return GFC_ERR_INVALID;
  const long long rows = (long long)B * h8 * w8;
  const long long grid = (rows + DH_ROWS - 1) / DH_ROWS;
  if (grid >= (1ll << 31)) return GFC_ERR_UNSUPPORTED;
  static_assert(DH_ROWS * DH_LP <= 2 * DH_TILE, "the logits tile aliases the staging buffers");
  hipLaunchKernelGGL(det_head_kernel, dim3((unsigned)grid), dim3(256), 0, st, hidden, lda, wp, bias, scale, shift, rows, h8,
                     w8, heat);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_sp_detector_head(const float* hidden, int lda, const float* w, const float* bias, const float* scale,
                                    const float* shift, int B, int h8, int w8, float* heat, void* stream) {
  return gfc_det_head_softmax_d2s(hidden, lda, w, bias, scale, shift, B, h8, w8, heat, (hipStream_t)stream);
}
