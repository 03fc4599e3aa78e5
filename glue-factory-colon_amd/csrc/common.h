// Shared device helpers for the gfx950 kernels (wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gfc_amd.h"
#include "runtime.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D(32x32) += A(32x2) * B(2x32).  Lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] lives in register r of lane l.
// Exact fp32 (a k-ordered fmaf chain), 64 cycles per SIMD.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Row of accumulator register r for lane half h in a 32x32 tile.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// XCD-aware work order.  Workgroups whose flat dispatch ids agree modulo 8 run on the same XCD, i.e. behind the same
// 4 MB L2 (observed placement, MI355X_MICROARCH.md: a label, used for speed only -- results never depend on it).
// gfc_xcd_chunk maps flat id -> logical index such that each XCD walks ONE contiguous chunk of the logical index space
// in dispatch order: workgroups that share operands (the column tiles of a GEMM row panel, the query blocks of an
// attention head, the output-channel blocks and neighbours of a convolution tile) then hit the L2 that already holds
// them.  Bijective on [0, n) for every n.
__device__ __forceinline__ unsigned gfc_xcd_chunk(unsigned id, unsigned n) {
  const unsigned q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// erff, branch-free.  The device library's erff evaluates one of two polynomials behind a divergent branch (|x| < 1 or
// not): inlined into an epilogue that is ~55 instructions and an exec-mask round trip per element, each element its own
// basic block.  This is the SAME arithmetic -- the same coefficients, the same fma chain, the same expf(-p) reduction
// (checked bit for bit against erff over all 2^32 inputs: tools/micro/erf_check.hip) -- with both polynomials evaluated
// and one select, so that consecutive elements schedule into each other.
__device__ __forceinline__ float gfc_erff(float x) {
  const float ax = fabsf(x);
  // |x| >= 1: 1 - exp(-p(|x|))
  float p = fmaf(ax, __uint_as_float(0x378e98abu), __uint_as_float(0xb9c68948u));
  p = fmaf(ax, p, __uint_as_float(0x3b7cd369u));
  p = fmaf(ax, p, __uint_as_float(0xbcc618b2u));
  p = fmaf(ax, p, __uint_as_float(0x3dda74e4u));
  p = fmaf(ax, p, __uint_as_float(0x3f228afdu));
  p = fmaf(ax, p, __uint_as_float(0x3e03c728u));
  p = fmaf(ax, p, ax);
  const float nl2e = __uint_as_float(0xbfb8aa3bu);  // -log2(e), high part
  const float ph = nl2e * p;
  float pl = fmaf(p, nl2e, -ph);
  const float pr = rintf(ph);
  pl = fmaf(p, __uint_as_float(0xb2a5705fu), pl);  // -log2(e), low part
  float e = __builtin_amdgcn_exp2f((ph - pr) + pl);
  e = ldexpf(e, (int)pr);
  e = (__uint_as_float(0x42ce8ed0u) < p) ? 0.f : e;       // exp(-p) underflows
  e = (__uint_as_float(0xc2b17218u) > p) ? INFINITY : e;  // (never for p >= 0; kept for bit-equality on NaN / -x paths)
  const float big = 1.0f - e;
  // |x| < 1: |x| + |x| q(x^2)
  const float t = x * x;
  float q = fmaf(t, __uint_as_float(0xba1345e1u), __uint_as_float(0x3ba10414u));
  q = fmaf(t, q, __uint_as_float(0xbcdac9b8u));
  q = fmaf(t, q, __uint_as_float(0x3de703beu));
  q = fmaf(t, q, __uint_as_float(0xbec09330u));
  q = fmaf(t, q, __uint_as_float(0x3e0375d0u));
  const float small = fmaf(ax, q, ax);
  const float r = !(ax < 1.0f) ? big : small;
  return copysignf(r, x);
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 in exact arithmetic,
// 5.6e-7 evaluated in fp32; on the GELU: 4.7e-7 absolute over [-9, 9] against 4.5e-7 for an exactly rounded fp32 erf --
// both are the fp32 rounding of the product): ~14 instructions instead of the ~45 of gfc_erff.  GFC_EXACT_ERF = 1
// builds the library with gfc_erff (bit-identical to the device library's erff) in the GELU instead.
__device__ __forceinline__ float gfc_gelu(float x) {
#if defined(GFC_EXACT_ERF) && GFC_EXACT_ERF
  return 0.5f * x * (1.f + gfc_erff(x * 0.70710678118654752440f));
#else
  const float z = x * 0.70710678118654752440f, a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-(a * a) * 1.4426950408889634f);  // exp(-z^2)
  const float r = fmaf(-p, e, 1.f);                                         // erf(|z|)
  return 0.5f * x * (1.f + copysignf(r, z));
#endif
}

#define GFC_LAUNCH_CHECK()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return GFC_ERR_LAUNCH; \
  } while (0)

// library-internal variants (not part of the C ABI): rotary operands as ONE packed table [rows][32 frequencies][cos, sin]
int gfc_lg_posenc_packed(const float* kpts, const float* scale_ori, const float* sizes, const int32_t* row0,
                         const int32_t* n, int n_images, int max_n, const float* wr, int dim, float* cos_out, float* sin_out,
                         float* cs_out, void* stream);
int gfc_linear_rot_packed(const float* A0, int lda0, int K0, const float* W, int ldw, const float* bias, const float* rot_cs,
                          int rot_cols, float* Y, int ldy, int M, int N, void* stream);

static inline size_t gfc_align(size_t x) { return (x + 255) & ~(size_t)255; }
