// EXPERIMENTAL (opt-in, not on the default path): 3x3 convolution on the bf16 matrix pipe at fp32 accuracy.
//
// Every fp32 operand is split into three bf16 planes  x = hi + mid + lo  (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid): 24+ mantissa bits together) and each product a*b is evaluated as the six bf16 products
//   lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi        (dropped terms <= 2^-24 |a||b|)
// with v_mfma_f32_32x32x16_bf16, whose products are exact in fp32 and which accumulates in fp32.  bf16 MFMA runs at
// 16x the fp32-MFMA rate, six products per fp32 product leave 2.67x.  Same implicit-GEMM structure, tile mapping and
// epilogues as conv.hip (reference superpoint_open.py:61-77; the arithmetic differs from the fp32 MFMA path by
// rounding only -- see tests/test_gpu_primitives.py::test_conv3x3_split_accuracy for the measured error).
// Weights are split once at pack time (gfc_pack_conv3x3_split); activations stay fp32 in HBM and are split while
// they are staged into LDS.
#include <stdlib.h>

#include "common.h"

#define ST 16               // output tile edge
#define SH (ST + 2)         // halo tile edge
#define SKC 16              // channels per chunk = one 32x32x16 MFMA k block
#define SNB 64              // output channels per workgroup
#define SROW 112            // bytes per LDS row: 3 planes x 16 bf16 (96 B) + 16 B pad (row pitch / 16 odd: conflict-free b128)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

// w [cout][cin][3][3] fp32 -> [cin/16][9][cout][3 planes][16] bf16
__global__ void pack_conv3x3_split_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * cin * 9) return;
  const int c = i % 16;
  int t = i / 16;
  const int co = t % cout; t /= cout;
  const int tap = t % 9;
  const int chunk = t / 9;
  const float v = w[((size_t)co * cin + chunk * 16 + c) * 9 + tap];
  __bf16 hi, mid, lo;
  split3(v, hi, mid, lo);
  __bf16* o = out + ((((size_t)chunk * 9 + tap) * cout + co) * 3) * 16 + c;
  o[0] = hi; o[16] = mid; o[32] = lo;
}

struct SplitArgs {
  const float* x;
  const __bf16* w;
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;
  int B, H, W, cin, cout, relu, tiles_x, tiles_y;
  // STEM variant only (x = 1-channel image): the cin = 1 layer in front, evaluated on the halo tile (as in conv.hip)
  const float* w1;   // [9][64]
  const float* b1;
  const float* s1;   // nullable
  const float* t1;
};
#define SIM (ST + 4)

template <bool POOL, bool STEM>
__global__ __launch_bounds__(256, STEM ? 2 : 3) void conv3x3_split_kernel(SplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* in_s = smem_raw;                       // [SH*SH][SROW]
  char* w_s = smem_raw + SH * SH * SROW;       // [2][SNB][SROW]
  float* img_s = reinterpret_cast<float*>(w_s + 2 * SNB * SROW);  // STEM: [SIM*SIM] image patch
  float* c1_s = img_s + SIM * SIM;                                // STEM: w1 [9][64], b1, s1, t1 [64] each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y, nb = blockIdx.y;
  const int x0 = tx * ST, y0 = ty * ST, cin = a.cin;
  const int nchunks = cin / SKC, nsteps = nchunks * 9;
  const float* xin = a.x + (size_t)b * a.H * a.W * (STEM ? 1 : cin);
  // weights of step s = chunk*9 + tap for this output-channel block: 64 rows x 96 B, contiguous
  const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)nb * SNB * 96;
  const size_t wstep = (size_t)a.cout * 96;

  // ---- staging ----
  constexpr int NI = (SH * SH * 4 + 255) / 256;  // float4 (4 channels) per thread per chunk
  float4 ireg[NI];
  float4 wreg0, wreg1;
  wreg1 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int c4 = (tid & 3) * 4;
#define SP_LOAD_IN(chunk_)                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                            \
    const int idx_ = tid + 256 * i_;                                                             \
    const int p_ = idx_ >> 2;                                                                    \
    const int gy_ = y0 - 1 + p_ / SH, gx_ = x0 - 1 + p_ % SH;                                    \
    float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                 \
    if (idx_ < SH * SH * 4 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W)                    \
      v_ = *reinterpret_cast<const float4*>(xin + ((size_t)gy_ * a.W + gx_) * cin + (chunk_) * SKC + c4); \
    ireg[i_] = v_;                                                                               \
  }
  // STEM: conv1a (1 -> 64, conv + ReLU + BN affine) of 4 channels of a halo pixel from the image patch in LDS
#define SP_FILL_IN(chunk_)                                                                       \
  {                                                                                              \
    const int c0_ = (chunk_) * SKC + c4;                                                         \
    float4 wv_[9];                                                                               \
    _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_)                                             \
        wv_[t_] = *reinterpret_cast<const float4*>(c1_s + t_ * 64 + c0_);                        \
    const float4 b1_ = *reinterpret_cast<const float4*>(c1_s + 576 + c0_);                       \
    const float4 s1_ = *reinterpret_cast<const float4*>(c1_s + 640 + c0_);                       \
    const float4 t1_ = *reinterpret_cast<const float4*>(c1_s + 704 + c0_);                       \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                          \
      const int idx_ = tid + 256 * i_;                                                           \
      const int p_ = idx_ >> 2;                                                                  \
      const int py_ = p_ / SH, px_ = p_ % SH;                                                    \
      const int gy_ = y0 - 1 + py_, gx_ = x0 - 1 + px_;                                          \
      float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                               \
      if (idx_ < SH * SH * 4 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W) {                \
        _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) {                                       \
          const float f_ = img_s[(py_ + t_ / 3) * SIM + px_ + t_ % 3];                           \
          v_.x = fmaf(f_, wv_[t_].x, v_.x);                                                      \
          v_.y = fmaf(f_, wv_[t_].y, v_.y);                                                      \
          v_.z = fmaf(f_, wv_[t_].z, v_.z);                                                      \
          v_.w = fmaf(f_, wv_[t_].w, v_.w);                                                      \
        }                                                                                        \
        v_.x = fmaxf(v_.x + b1_.x, 0.f) * s1_.x + t1_.x;                                         \
        v_.y = fmaxf(v_.y + b1_.y, 0.f) * s1_.y + t1_.y;                                         \
        v_.z = fmaxf(v_.z + b1_.z, 0.f) * s1_.z + t1_.z;                                         \
        v_.w = fmaxf(v_.w + b1_.w, 0.f) * s1_.w + t1_.w;                                         \
      }                                                                                          \
      ireg[i_] = v_;                                                                             \
    }                                                                                            \
  }
#define SP_STORE_IN()                                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                            \
    const int idx_ = tid + 256 * i_;                                                             \
    if (idx_ < SH * SH * 4) {                                                                    \
      __bf16 h0_, m0_, l0_, h1_, m1_, l1_, h2_, m2_, l2_, h3_, m3_, l3_;                          \
      split3(ireg[i_].x, h0_, m0_, l0_);                                                         \
      split3(ireg[i_].y, h1_, m1_, l1_);                                                         \
      split3(ireg[i_].z, h2_, m2_, l2_);                                                         \
      split3(ireg[i_].w, h3_, m3_, l3_);                                                         \
      const bf16x4 hi_ = {h0_, h1_, h2_, h3_}, mid_ = {m0_, m1_, m2_, m3_}, lo_ = {l0_, l1_, l2_, l3_}; \
      char* d_ = in_s + (idx_ >> 2) * SROW + c4 * 2;                                             \
      *reinterpret_cast<bf16x4*>(d_) = hi_;                                                      \
      *reinterpret_cast<bf16x4*>(d_ + 32) = mid_;                                                \
      *reinterpret_cast<bf16x4*>(d_ + 64) = lo_;                                                 \
    }                                                                                            \
  }
  // weight slice: 384 pieces of 16 B; thread t copies piece t and (t < 128) piece t + 256
#define SP_LOAD_W(step_)                                                                         \
  do {                                                                                           \
    const char* s_ = wbase + (size_t)(step_) * wstep;                                            \
    wreg0 = *reinterpret_cast<const float4*>(s_ + tid * 16);                                     \
    if (tid < 128) wreg1 = *reinterpret_cast<const float4*>(s_ + (tid + 256) * 16);              \
  } while (0)
#define SP_STORE_W(buf_)                                                                         \
  do {                                                                                           \
    char* d_ = w_s + (buf_) * SNB * SROW;                                                        \
    *reinterpret_cast<float4*>(d_ + (tid / 6) * SROW + (tid % 6) * 16) = wreg0;                  \
    if (tid < 128) *reinterpret_cast<float4*>(d_ + ((tid + 256) / 6) * SROW + ((tid + 256) % 6) * 16) = wreg1; \
  } while (0)

  // MFMA tile (wave, mt) = image rows 2*wave + mt and that + 8 (conflict-free, see conv.hip); lane = pixel / channel row
  int a_off[2], b_off[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) a_off[mt] = ((2 * wave + mt + 8 * (l31 >> 4)) * SH + (l31 & 15)) * SROW + h * 16;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) b_off[nt] = (nt * 32 + l31) * SROW + h * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  if constexpr (STEM) {
    for (int i = tid; i < SIM * SIM; i += 256) {
      const int gy = y0 - 2 + i / SIM, gx = x0 - 2 + i % SIM;
      img_s[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? xin[(size_t)gy * a.W + gx] : 0.f;
    }
    for (int i = tid; i < 768; i += 256)
      c1_s[i] = i < 576 ? a.w1[i] : i < 640 ? a.b1[i - 576] : i < 704 ? (a.s1 ? a.s1[i - 640] : 1.f)
                                                                         : (a.t1 ? a.t1[i - 704] : 0.f);
    __syncthreads();
    SP_FILL_IN(0);
  } else {
    SP_LOAD_IN(0);
  }
  SP_LOAD_W(0);
  SP_STORE_IN();
  SP_STORE_W(0);
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / 9, tap = step - chunk * 9;
    const bool has_next = step + 1 < nsteps;
    const bool new_chunk = has_next && tap == 8;
    if (has_next) SP_LOAD_W(step + 1);
    if constexpr (STEM) {
      if (new_chunk) SP_FILL_IN(chunk + 1);
    } else {
      if (tap == 2 && chunk + 1 < nchunks) { SP_LOAD_IN(chunk + 1); }
    }
    const int dy = tap / 3, dx = tap - dy * 3;
    const char* ap = in_s + (dy * SH + dx) * SROW;
    const char* bp = w_s + (step & 1) * SNB * SROW;
    bf16x8 af[2][3], bf[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt][pl] = *reinterpret_cast<const bf16x8*>(ap + a_off[mt] + 32 * pl);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bf[nt][pl] = *reinterpret_cast<const bf16x8*>(bp + b_off[nt] + 32 * pl);
    }
    // smallest terms first; planes: 0 = hi, 1 = mid, 2 = lo
#define SP_MM(pa_, pb_)                                                                               \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[0][pb_], acc[0][0], 0, 0, 0);    \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[1][pb_], acc[0][1], 0, 0, 0);    \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[0][pb_], acc[1][0], 0, 0, 0);    \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[1][pb_], acc[1][1], 0, 0, 0);
    SP_MM(2, 0) SP_MM(0, 2) SP_MM(1, 1) SP_MM(1, 0) SP_MM(0, 1) SP_MM(0, 0)
#undef SP_MM
    if (has_next) SP_STORE_W((step + 1) & 1);
    if (new_chunk) {
      __syncthreads();
      SP_STORE_IN();
    }
    __syncthreads();
  }

  // ---- epilogue (as conv.hip) ----
  if (!POOL) {
    constexpr int ELD = 68;
    float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * ELD;
    const int er = lane >> 4, ec = (lane & 15) * 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int co = nb * SNB + nt * 32 + l31;
        const float bi = a.bias[co];
        const float sc = a.scale ? a.scale[co] : 1.f;
        const float sh = a.shift ? a.shift[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = acc[mt][nt][r] + bi;
          if (a.relu) t = fmaxf(t, 0.f);
          patch[acc_row(r, h) * ELD + nt * 32 + l31] = t * sc + sh;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = er + 4 * i;
        const int gy = y0 + 2 * wave + mt + 8 * (p >> 4), gx = x0 + (p & 15);
        const float4 v = *reinterpret_cast<const float4*>(patch + p * ELD + ec);
        if (gy < a.H && gx < a.W)
          *reinterpret_cast<float4*>(a.y + (((size_t)b * a.H + gy) * a.W + gx) * a.cout + nb * SNB + ec) = v;
      }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int co = nb * SNB + nt * 32 + l31;
      const float bi = a.bias[co];
      const float sc = a.scale ? a.scale[co] : 1.f;
      const float sh = a.shift ? a.shift[co] : 0.f;
      float v0[16], v1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float t0 = acc[0][nt][r] + bi, t1 = acc[1][nt][r] + bi;
        if (a.relu) { t0 = fmaxf(t0, 0.f); t1 = fmaxf(t1, 0.f); }
        v0[r] = t0 * sc + sh;
        v1[r] = t1 * sc + sh;
      }
      const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 2 * q;
        const float m = fmaxf(fmaxf(v0[r], v0[r + 1]), fmaxf(v1[r], v1[r + 1]));
        const int pxl = (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;
        const int oy = (y0 >> 1) + wave + 4 * (r >> 3), ox = (x0 >> 1) + (pxl >> 1);
        if (oy < Ho && ox < Wo) a.y[(((size_t)b * Ho + oy) * Wo + ox) * a.cout + co] = m;
      }
    }
  }
}

extern "C" int gfc_pack_conv3x3_split(const float* w_oihw, void* w_split, int cout, int cin, void* stream) {
  if (!w_oihw || !w_split || cout <= 0 || cin <= 0 || cin % SKC) return GFC_ERR_INVALID;
  const int total = cout * cin * 9;
  hipLaunchKernelGGL(pack_conv3x3_split_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (__bf16*)w_split, cout, cin);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_conv3x3_split(const float* x, const void* w_split, const float* bias, const float* scale,
                                 const float* shift, float* y, int B, int H, int W, int cin, int cout, int relu,
                                 int pool, void* stream) {
  if (!x || !w_split || !bias || !y || B <= 0 || H <= 0 || W <= 0) return GFC_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return GFC_ERR_INVALID;
  if (cin % SKC != 0 || cout % SNB != 0) return GFC_ERR_UNSUPPORTED;
  SplitArgs a = {};
  a.x = x; a.w = (const __bf16*)w_split; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y;
  a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.tiles_x = (W + ST - 1) / ST; a.tiles_y = (H + ST - 1) / ST;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * B), cout / SNB);
  const size_t lds = (size_t)SH * SH * SROW + 2 * SNB * SROW;
  if (pool) hipLaunchKernelGGL((conv3x3_split_kernel<true, false>), grid, dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv3x3_split_kernel<false, false>), grid, dim3(256), lds, (hipStream_t)stream, a);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// conv1a (1 -> 64, fp32 VALU on the halo tile) + conv1b (64 -> 64, split bf16 MFMA) + 2x2 max-pool in one launch
extern "C" int gfc_sp_stem_split(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                                 const void* w2_split, const float* b2, const float* s2, const float* t2, float* y,
                                 int B, int H, int W, void* stream) {
  if (!image || !w1 || !b1 || !w2_split || !b2 || !y || B <= 0 || H < 2 || W < 2) return GFC_ERR_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return GFC_ERR_INVALID;
  SplitArgs a = {};
  a.x = image; a.w = (const __bf16*)w2_split; a.bias = b2; a.scale = s2; a.shift = t2; a.y = y;
  a.B = B; a.H = H; a.W = W; a.cin = 64; a.cout = 64; a.relu = 1;
  a.w1 = w1; a.b1 = b1; a.s1 = s1; a.t1 = t1;
  a.tiles_x = (W + ST - 1) / ST; a.tiles_y = (H + ST - 1) / ST;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * B), 1);
  const size_t lds = (size_t)SH * SH * SROW + 2 * SNB * SROW + (SIM * SIM + 768) * sizeof(float);
  hipLaunchKernelGGL((conv3x3_split_kernel<true, true>), grid, dim3(256), lds, (hipStream_t)stream, a);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
