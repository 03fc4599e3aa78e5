// Image preprocessing on the GPU ("next" row rank 1 of the scope table: the step right before the extractor).
//
// Replaces ImagePreprocessor.__call__'s resize (reference gluefactory/utils/image.py:33-47) and the uint8 -> float
// conversion of numpy_image_to_torch (image.py:148-156).  The resize the reference calls is kornia's
// geometry.transform.resize (third-party, absent here -> parity unpinned): when down-scaling with antialias=True a
// Gaussian blur (sigma = max((factor-1)/2, 0.001) per axis, kernel size int(max(4 sigma, 3)) made odd, reflect
// border) and then torch's bilinear interpolation (align_corners False/True).  Both steps are fused: each output
// pixel evaluates the blurred image at its 4 bilinear taps (horizontal pass, then vertical, like a separable filter);
// nothing intermediate is written.  HBM-bound: one read of the source (L2 serves the window overlap), one write.
#include "common.h"

#define PP_MAX_KS 63

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // torch 'reflect' padding: -1 -> 1, n -> n-2 (pad < n is checked on the host)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// One source value.  U8: interleaved HWC bytes through a 256-entry table of (float)((double)v / 255.0) -- value / 255 in
// double, rounded once, as numpy_image_to_torch does -- built once per workgroup in LDS.
template <bool U8>
__device__ __forceinline__ float load_px(const void* src, const float* lut, int H, int W, int C, int c, int y, int x, int bgr) {
  if (U8) {
    const unsigned char* s = static_cast<const unsigned char*>(src);
    const int cc = bgr ? (C - 1 - c) : c;
    return lut[s[((size_t)y * W + x) * C + cc]];
  }
  return static_cast<const float*>(src)[((size_t)c * H + y) * W + x];  // planar CHW float
}

template <bool U8>
__global__ __launch_bounds__(256) void resize_kernel(const void* __restrict__ src, int H, int W, int C, int bgr,
                                                     float* __restrict__ dst, int OH, int OW, int align_corners,
                                                     int ksy, int ksx, float sgy, float sgx, long long src_stride,
                                                     long long dst_stride) {
  __shared__ float wy[PP_MAX_KS], wx[PP_MAX_KS];
  __shared__ float lut[256];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if (U8) lut[tid] = (float)((double)tid / 255.0);
  if (ksy > 0) {
    // kornia gaussian(): x = arange(ks) - ks//2 (ks is odd here), exp(-x^2 / (2 sigma^2)), normalised
    if (tid < ksy) { const float d = (float)(tid - ksy / 2); wy[tid] = expf(-(d * d) / (2.f * sgy * sgy)); }
    if (tid >= 64 && tid < 64 + ksx) { const float d = (float)(tid - 64 - ksx / 2); wx[tid - 64] = expf(-(d * d) / (2.f * sgx * sgx)); }
    __syncthreads();
    float sy = 0.f, sx = 0.f;
    for (int i = 0; i < ksy; ++i) sy += wy[i];
    for (int i = 0; i < ksx; ++i) sx += wx[i];
    __syncthreads();
    if (tid < ksy) wy[tid] = wy[tid] / sy;
    if (tid >= 64 && tid < 64 + ksx) wx[tid - 64] = wx[tid - 64] / sx;
  }
  __syncthreads();
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= OW || oy >= OH) return;
  const char* sb = static_cast<const char*>(src) + (size_t)blockIdx.z * src_stride;
  float* db = dst + (size_t)blockIdx.z * dst_stride;
  // torch upsample_bilinear2d source coordinates (area_pixel_compute_source_index)
  float fy, fx;
  if (align_corners) {
    const float scy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, scx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    fy = scy * oy; fx = scx * ox;
  } else {
    const float scy = (float)H / (float)OH, scx = (float)W / (float)OW;
    // one rounding, as torch's CPU kernel evaluates scale * (dst + 0.5) - 0.5 (its translation unit is built with FMA
    // contraction: checked bit for bit against F.interpolate, tools/micro/fuzz_shapes.py; with two roundings the source
    // coordinate is off by an ulp at coordinates > 64, i.e. the tap weight by ~1e-5)
    fy = fmaxf(fmaf(scy, oy + 0.5f, -0.5f), 0.f); fx = fmaxf(fmaf(scx, ox + 0.5f, -0.5f), 0.f);
  }
  const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  for (int c = 0; c < C; ++c) {
    float v00, v01, v10, v11;
    if (ksy > 0) {
      // The blurred image at the four bilinear taps.  Each tap is sum_i wy[i] * (sum_j wx[j] * px) in that order (the
      // separable filter of the restatement: horizontal pass, then vertical); the horizontal sums of a source row at
      // x0 and x1 serve both vertical taps, so every source row of the (ksy + 1) x (ksx + 1) window is read once.
      const int ry = ksy / 2, rx = ksx / 2;
      float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
      const bool two_y = y1 != y0, two_x = x1 != x0;
      for (int i = 0; i <= ksy; ++i) {           // window row i <-> source row y0 - ry + i (before reflection)
        if (i == ksy && !two_y) break;
        // row y0 - ry + i is tap i of (y0, .) and tap i - 1 of (y1, .) when y1 = y0 + 1.  Reflection must be applied
        // per (tap centre, offset) as the padded image defines it: reflect(cy + i' - ry) -- for y1 = y0 + 1 the two
        // expressions name the same un-reflected row, hence the same reflected row.
        const int yy = reflect_idx(y0 - ry + i, H);
        float r0 = 0.f, r1 = 0.f;
        for (int j = 0; j <= ksx; ++j) {
          if (j == ksx && !two_x) break;
          const float px = load_px<U8>(sb, lut, H, W, C, c, yy, reflect_idx(x0 - rx + j, W), bgr);
          if (j < ksx) r0 += wx[j] * px;
          if (two_x && j > 0) r1 += wx[j - 1] * px;
        }
        if (!two_x) r1 = r0;
        if (i < ksy) { a00 += wy[i] * r0; a01 += wy[i] * r1; }
        if (two_y && i > 0) { a10 += wy[i - 1] * r0; a11 += wy[i - 1] * r1; }
      }
      if (!two_y) { a10 = a00; a11 = a01; }
      v00 = a00; v01 = a01; v10 = a10; v11 = a11;
    } else {
      v00 = load_px<U8>(sb, lut, H, W, C, c, y0, x0, bgr); v01 = load_px<U8>(sb, lut, H, W, C, c, y0, x1, bgr);
      v10 = load_px<U8>(sb, lut, H, W, C, c, y1, x0, bgr); v11 = load_px<U8>(sb, lut, H, W, C, c, y1, x1, bgr);
    }
    db[((size_t)c * OH + oy) * OW + ox] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}

extern "C" int gfc_preprocess_resize(const void* src, int src_is_u8_hwc, int bgr, int B, int C, int H, int W, float* dst,
                                     int OH, int OW, int align_corners, int antialias, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return GFC_ERR_INVALID;
  int ksy = 0, ksx = 0;
  float sgy = 0.f, sgx = 0.f;
  const float facy = (float)H / (float)OH, facx = (float)W / (float)OW;
  if (antialias && fmaxf(facy, facx) > 1.f) {  // kornia blurs only when some axis is down-scaled
    sgy = fmaxf((facy - 1.f) / 2.f, 0.001f);
    sgx = fmaxf((facx - 1.f) / 2.f, 0.001f);
    ksy = (int)fmaxf(2.f * 2.f * sgy, 3.f);
    ksx = (int)fmaxf(2.f * 2.f * sgx, 3.f);
    if (ksy % 2 == 0) ++ksy;
    if (ksx % 2 == 0) ++ksx;
    if (ksy > PP_MAX_KS || ksx > PP_MAX_KS) return GFC_ERR_UNSUPPORTED;    // down-scaling by more than ~30x
    if (ksy / 2 >= H || ksx / 2 >= W) return GFC_ERR_INVALID;              // reflect padding needs pad < size
  }
  const dim3 block(32, 8), grid((OW + 31) / 32, (OH + 7) / 8, B);
  const long long sstride = (long long)C * H * W * (src_is_u8_hwc ? 1 : 4), dstride = (long long)C * OH * OW;
  if (src_is_u8_hwc)
    hipLaunchKernelGGL(resize_kernel<true>, grid, block, 0, (hipStream_t)stream, src, H, W, C, bgr, dst, OH, OW,
                       align_corners, ksy, ksx, sgy, sgx, sstride, dstride);
  else
    hipLaunchKernelGGL(resize_kernel<false>, grid, block, 0, (hipStream_t)stream, src, H, W, C, 0, dst, OH, OW,
                       align_corners, ksy, ksx, sgy, sgx, sstride, dstride);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
