// 3x3 convolutions of the SuperPoint encoder as Winograd F(2x2, 3x3) on the fp32 matrix pipe (NHWC).
//
// Same function as conv.hip (reference gluefactory/models/extractors/superpoint_open.py:61-77,100-118 and
// gluefactory_nonfree/superpoint.py:214-241: conv3x3 -> ReLU [-> BatchNorm(eval)] [-> MaxPool2d(2,2)]), computed
// with 16 multiplications per 2x2 output block instead of 36:
//     Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A
// Every product is an fp32 product accumulated in fp32 (v_mfma_f32_32x32x2_f32); the transforms are fp32
// additions (B^T, A^T have entries 0, +-1) and the filter transform G g G^T is evaluated once, in float64, when the
// weights are packed.  Through the whole SuperPoint-open stack the heat-map error against a float64 evaluation is the
// same as the direct fp32 convolution's (2.4e-6 vs 3.2e-6, tools/micro/winograd_numerics.py); the MFMA work drops by
// 2.25x.
//
// Mapping.  The 16 transform positions (xi, nu) are 16 independent GEMMs  M[xi,nu][tile, cout] = V[xi,nu][tile, cin] .
// U[xi,nu][cin, cout].  One workgroup = 4 waves = a 16 x 8 pixel output tile (8 x 4 = 32 Winograd tiles = one MFMA M
// tile; lane <-> tile) x 64 output channels; wave xi owns the four positions (xi, 0..3) x 2 cout tiles = 8
// accumulators.
//   * A operand: the 18 x 10 input halo patch (16-channel chunks) sits in LDS, double buffered.  A wave builds its
//     fragments on the fly: row transform (two patch rows, +-) then column transform, 8 ds_read_b128 + 8 float4 VALU
//     operations per 8-deep k group and 32 MFMAs.  Patch rows / columns are stored de-interleaved (even rows first,
//     even columns first) at a 12-pixel row pitch with the 16-byte channel slots XOR-swizzled by (row & 3): every
//     ds_read_b128 service group of 16 lanes then hits 16 distinct 4-bank slots (conflict-free, no padding floats).
//   * B operand: the transformed filters never enter LDS.  They are packed in MFMA-fragment order, so a wave reads
//     the fragment of (position, cout tile, k group) as one fully coalesced 1 KB load straight into registers, one
//     position ahead of its use (L2 / L1 resident: 0.26 - 4 MB per layer).
//   * Epilogue: column transform lane-local (the four positions of a wave), row transform across the four waves
//     through LDS; bias, ReLU, BN affine, optional 2x2 max-pool (one Winograd tile = one pooling window), NHWC store.
// STEM = true additionally evaluates conv1a (1 -> 64, conv + ReLU + BN) on the halo patch from a 20 x 12 image patch
// in LDS (as conv.hip's stem does), so the [B,H,W,64] activation never exists in HBM.
#include "common.h"

#define WT_Y 16                    // output tile rows
#define WT_X 8                     // output tile columns
#define WP_R 18                    // halo patch rows
#define WP_C 12                    // patch row pitch in pixels (10 used: 5 even + 5 odd columns)
#define WKC 16                     // channels per input chunk
#define WPATCH (WP_R * WP_C * WKC)  // floats per patch buffer
#define WIM_R 20                   // STEM: image patch rows
#define WIM_C 12                   // STEM: image patch columns

struct WinoArgs {
  const float* x;
  const float* w;  // packed by gfc_pack_conv3x3_wino
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;
  int B, H, W, cin, cout, relu;
  int tiles_x, tiles_y;
  const float* w1;  // STEM: conv1a [9][64], bias, BN scale / shift (nullable)
  const float* b1;
  const float* s1;
  const float* t1;
};

// U = G g G^T in float64, rounded once; scattered into MFMA-fragment order:
//   out[nb][xi][kg][nu][nt][lane = 32 h + l31][s]  <-  U[xi][nu] of (cout = 64 nb + 32 nt + l31, cin = 8 kg + 4 h + s)
__global__ void pack_conv3x3_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cout * cin) return;
  const int ci = idx % cin, co = idx / cin;
  double g[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) g[r][c] = (double)w[((size_t)co * cin + ci) * 9 + r * 3 + c];
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  double t[4][3];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 3; ++c) t[i][c] = G[i][0] * g[0][c] + G[i][1] * g[1][c] + G[i][2] * g[2][c];
  const int nb = co / 64, nt = (co % 64) / 32, l31 = co % 32;
  const int kg = ci / 8, h = (ci % 8) / 4, s = ci % 4;
  const int lane = 32 * h + l31;
  for (int xi = 0; xi < 4; ++xi)
    for (int nu = 0; nu < 4; ++nu) {
      const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
      const size_t o = ((((((size_t)nb * 4 + xi) * (cin / 8) + kg) * 4 + nu) * 2 + nt) * 64 + lane) * 4 + s;
      out[o] = (float)u;
    }
}

extern "C" int gfc_pack_conv3x3_wino(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
  if (!w_oihw || !w_packed || cout <= 0 || cin <= 0 || cout % 64 || cin % WKC) return GFC_ERR_INVALID;
  const int total = cout * cin;
  hipLaunchKernelGGL(pack_conv3x3_wino_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_packed, cout, cin);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

__device__ __forceinline__ float4 f4_axpy(float s, float4 a, float4 b) {  // s * a + b with s = +-1: exact
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <bool POOL, bool STEM>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_s = smem;                      // [2][WPATCH]; the epilogue exchange buffer (16384 floats) aliases it
  float* img_s = smem + 2 * WPATCH;        // STEM: [WIM_R][WIM_C] image patch
  float* c1_s = img_s + WIM_R * WIM_C;     // STEM: conv1a w [9][64], b [64], s [64], t [64]

  const int tid = threadIdx.x, lane = tid & 63, xi = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int cin = a.cin, nchunks = cin / WKC;

  // work item: tile + ntiles * (output-channel block)
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  int t_ = blockIdx.x % ntiles;
  const int nb = blockIdx.x / ntiles;
  const int x0 = (t_ % a.tiles_x) * WT_X;
  t_ /= a.tiles_x;
  const int y0 = (t_ % a.tiles_y) * WT_Y;
  const int b = t_ / a.tiles_y;
  const float* xin = a.x + (size_t)b * a.H * a.W * (STEM ? 1 : cin);

  // ---- B stream: fragments of this wave's four positions, contiguous per k group (8 x 64 float4) ----
  const float4* wp = reinterpret_cast<const float4*>(a.w) + (((size_t)nb * 4 + xi) * (cin / 8)) * 512 + lane;
  float4 bq[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bq[nu][nt] = wp[(nu * 2 + nt) * 64];

  // ---- patch staging: 18 x 10 pixels x 4 float4 = 720 float4 per chunk, three per thread ----
  float4 ireg[3];
  int st_off[3];   // LDS offset (floats) of this thread's three pieces; -1 = none
  int st_gofs[3];  // global offset (floats, without the chunk term) or -1 when the pixel lies outside the image
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i;
    const int pix = idx >> 2, c4 = idx & 3;
    const int py = pix / 10, px = pix - py * 10;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const int R = (py >> 1) + (py & 1) * 9, C = (px >> 1) + (px & 1) * 5;
    st_off[i] = idx < 720 ? (R * WP_C + C) * WKC + ((c4 ^ (R & 3)) << 2) : -1;
    const bool inside = idx < 720 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    if constexpr (STEM) st_gofs[i] = inside ? (py * WIM_C + px) : -1;  // offset into the image patch (tap (0,0) corner)
    else st_gofs[i] = inside ? (int)(((size_t)gy * a.W + gx) * cin + c4 * 4) : -1;
  }
#define WINO_LOAD_IN(chunk_)                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                               \
    float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
    if (st_gofs[i_] >= 0) v_ = *reinterpret_cast<const float4*>(xin + st_gofs[i_] + (chunk_) * WKC); \
    ireg[i_] = v_;                                                                                 \
  }
#define WINO_FILL_IN(chunk_)                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                               \
    float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
    if (st_gofs[i_] >= 0) {                                                                        \
      const int c0_ = (chunk_) * WKC + (((tid + 256 * i_) & 3) << 2);                              \
      const float* ip_ = img_s + st_gofs[i_];                                                      \
      _Pragma("unroll") for (int t2_ = 0; t2_ < 9; ++t2_) {                                        \
        const float f_ = ip_[(t2_ / 3) * WIM_C + t2_ % 3];                                         \
        const float4 w_ = *reinterpret_cast<const float4*>(c1_s + t2_ * 64 + c0_);                 \
        v_.x = fmaf(f_, w_.x, v_.x); v_.y = fmaf(f_, w_.y, v_.y);                                  \
        v_.z = fmaf(f_, w_.z, v_.z); v_.w = fmaf(f_, w_.w, v_.w);                                  \
      }                                                                                            \
      const float4 b1_ = *reinterpret_cast<const float4*>(c1_s + 576 + c0_);                       \
      const float4 s1_ = *reinterpret_cast<const float4*>(c1_s + 640 + c0_);                       \
      const float4 t1_ = *reinterpret_cast<const float4*>(c1_s + 704 + c0_);                       \
      v_.x = fmaxf(v_.x + b1_.x, 0.f) * s1_.x + t1_.x; v_.y = fmaxf(v_.y + b1_.y, 0.f) * s1_.y + t1_.y; \
      v_.z = fmaxf(v_.z + b1_.z, 0.f) * s1_.z + t1_.z; v_.w = fmaxf(v_.w + b1_.w, 0.f) * s1_.w + t1_.w; \
    }                                                                                              \
    ireg[i_] = v_;                                                                                 \
  }
#define WINO_STORE_IN(buf_)                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                                 \
    if (st_off[i_] >= 0) *reinterpret_cast<float4*>(in_s + (buf_) * WPATCH + st_off[i_]) = ireg[i_];

  if constexpr (STEM) {
    if (tid < WIM_R * WIM_C) {
      const int gy = y0 - 2 + tid / WIM_C, gx = x0 - 2 + tid % WIM_C;
      img_s[tid] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? xin[(size_t)gy * a.W + gx] : 0.f;
    }
    for (int i = tid; i < 768; i += 256)
      c1_s[i] = i < 576 ? a.w1[i] : i < 640 ? a.b1[i - 576] : i < 704 ? (a.s1 ? a.s1[i - 640] : 1.f)
                                                                         : (a.t1 ? a.t1[i - 704] : 0.f);
    __syncthreads();
    WINO_FILL_IN(0);
  } else {
    WINO_LOAD_IN(0);
  }
  WINO_STORE_IN(0);

  // ---- A fragments: this wave's row transform takes patch rows a1, a2 of every tile: d[a1] + sg * d[a2] ----
  //   xi = 0: d0 - d2   xi = 1: d1 + d2   xi = 2: d2 - d1   xi = 3: d1 - d3      (B^T of F(2x2,3x3))
  const int a1 = xi == 0 ? 0 : xi == 2 ? 2 : 1;
  const int a2 = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
  const float sg = xi == 1 ? 1.f : -1.f;
  const int ty = l31 >> 2, tx = l31 & 3;
  const int R1 = ty + (a1 >> 1) + (a1 & 1) * 9, R2 = ty + (a2 >> 1) + (a2 & 1) * 9;
  // float offsets of column b = 0 for k group 0; k group 1 toggles bit 3 (slot index ^ 2); columns b = 1, 2, 3 are the
  // de-interleaved positions +5, +1, +6 pixels
  const int o1 = (R1 * WP_C + tx) * WKC + ((h ^ (R1 & 3)) << 2);
  const int o2 = (R2 * WP_C + tx) * WKC + ((h ^ (R2 & 3)) << 2);

  f32x16 acc[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][nt][r] = 0.f;

  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool has_next = c + 1 < nchunks;
    if constexpr (!STEM) {
      if (has_next) WINO_LOAD_IN(c + 1);
    }
    const float* ps = in_s + (c & 1) * WPATCH;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float* p1 = ps + (g ? (o1 ^ 8) : o1);
      const float* p2 = ps + (g ? (o2 ^ 8) : o2);
      // row transform of the four columns, then the four column transforms
      const float4 t0 = f4_axpy(sg, *reinterpret_cast<const float4*>(p2), *reinterpret_cast<const float4*>(p1));
      const float4 t1 = f4_axpy(sg, *reinterpret_cast<const float4*>(p2 + 5 * WKC), *reinterpret_cast<const float4*>(p1 + 5 * WKC));
      const float4 t2 = f4_axpy(sg, *reinterpret_cast<const float4*>(p2 + 1 * WKC), *reinterpret_cast<const float4*>(p1 + 1 * WKC));
      const float4 t3 = f4_axpy(sg, *reinterpret_cast<const float4*>(p2 + 6 * WKC), *reinterpret_cast<const float4*>(p1 + 6 * WKC));
      float4 v[4];
      v[0] = f4_sub(t0, t2);
      v[1] = f4_add(t1, t2);
      v[2] = f4_sub(t2, t1);
      v[3] = f4_sub(t1, t3);
      const int kg_next = 2 * c + g + 1;
      const bool more_b = kg_next < cin / 8;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        acc[nu][0] = mfma32(v[nu].x, bq[nu][0].x, acc[nu][0]);
        acc[nu][1] = mfma32(v[nu].x, bq[nu][1].x, acc[nu][1]);
        acc[nu][0] = mfma32(v[nu].y, bq[nu][0].y, acc[nu][0]);
        acc[nu][1] = mfma32(v[nu].y, bq[nu][1].y, acc[nu][1]);
        acc[nu][0] = mfma32(v[nu].z, bq[nu][0].z, acc[nu][0]);
        acc[nu][1] = mfma32(v[nu].z, bq[nu][1].z, acc[nu][1]);
        acc[nu][0] = mfma32(v[nu].w, bq[nu][0].w, acc[nu][0]);
        acc[nu][1] = mfma32(v[nu].w, bq[nu][1].w, acc[nu][1]);
        // the registers of this position are free: fetch its fragments of the next k group
        if (more_b) {
          bq[nu][0] = wp[(size_t)kg_next * 512 + (nu * 2 + 0) * 64];
          bq[nu][1] = wp[(size_t)kg_next * 512 + (nu * 2 + 1) * 64];
        }
      }
      if constexpr (STEM) {
        if (g == 0 && has_next) WINO_FILL_IN(c + 1);  // VALU work under the MFMAs of this chunk
      }
    }
    if (has_next) WINO_STORE_IN((c + 1) & 1);
    __syncthreads();
  }
#undef WINO_LOAD_IN
#undef WINO_FILL_IN
#undef WINO_STORE_IN

  // ---- output transform.  Column direction (nu) lane-local: z0 = M0 + M1 + M2, z1 = M1 - M2 - M3 ----
  float* ex = smem;  // [xi 4][j 2][nt 2][r 16][64 lanes]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      ex[(((xi * 2 + 0) * 2 + nt) * 16 + r) * 64 + lane] = (m0 + m1) + m2;
      ex[(((xi * 2 + 1) * 2 + nt) * 16 + r) * 64 + lane] = (m1 - m2) - m3;
    }
  __syncthreads();
  // row direction (xi) across the waves: wave w finishes cout tile nt = w >> 1, accumulator registers 8 (w & 1) .. + 7
  const int nt_w = xi >> 1, rh = xi & 1;
  const int co = nb * 64 + nt_w * 32 + l31;
  const float bi = a.bias[co];
  const float sc = a.scale ? a.scale[co] : 1.f;
  const float sh = a.shift ? a.shift[co] : 0.f;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int r = rh * 8 + rr;
    float yv[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float z0 = ex[(((0 * 2 + j) * 2 + nt_w) * 16 + r) * 64 + lane];
      const float z1 = ex[(((1 * 2 + j) * 2 + nt_w) * 16 + r) * 64 + lane];
      const float z2 = ex[(((2 * 2 + j) * 2 + nt_w) * 16 + r) * 64 + lane];
      const float z3 = ex[(((3 * 2 + j) * 2 + nt_w) * 16 + r) * 64 + lane];
      yv[0][j] = (z0 + z1) + z2;
      yv[1][j] = (z1 - z2) - z3;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float t = yv[i][j] + bi;
        if (a.relu) t = fmaxf(t, 0.f);
        yv[i][j] = t * sc + sh;
      }
    const int tile = acc_row(r, h);  // Winograd tile of this register: (tile >> 2, tile & 3)
    const int oy = (y0 >> 1) + (tile >> 2), ox = (x0 >> 1) + (tile & 3);
    if constexpr (POOL) {
      const int Ho = a.H >> 1, Wo = a.W >> 1;
      if (oy < Ho && ox < Wo)
        a.y[(((size_t)b * Ho + oy) * Wo + ox) * a.cout + co] = fmaxf(fmaxf(yv[0][0], yv[0][1]), fmaxf(yv[1][0], yv[1][1]));
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gy = 2 * oy + i, gx = 2 * ox + j;
          if (gy < a.H && gx < a.W) a.y[(((size_t)b * a.H + gy) * a.W + gx) * a.cout + co] = yv[i][j];
        }
    }
  }
}

template <bool POOL, bool STEM>
static int launch_wino(const WinoArgs& a, hipStream_t st) {
  constexpr size_t patches = (size_t)2 * WPATCH + (STEM ? WIM_R * WIM_C + 768 : 0), exch = 4 * 2 * 2 * 16 * 64;
  constexpr size_t lds = (patches > exch ? patches : exch) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};
  if (lds > 64 * 1024) gfc_allow_dynamic_lds((const void*)conv3x3_wino_kernel<POOL, STEM>, lds, lds_ok);
  const long long nitems = (long long)a.tiles_x * a.tiles_y * a.B * (a.cout / 64);
  hipLaunchKernelGGL((conv3x3_wino_kernel<POOL, STEM>), dim3((unsigned)nitems), dim3(256), lds, st, a);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_conv3x3_wino(const float* x, const float* w_wino, const float* bias, const float* scale,
                                const float* shift, float* y, int B, int H, int W, int cin, int cout, int relu, int pool,
                                void* stream) {
  if (!x || !w_wino || !bias || !y || B <= 0 || H <= 0 || W <= 0) return GFC_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return GFC_ERR_INVALID;
  if (cin % WKC != 0 || cout % 64 != 0 || cin <= 0 || cout <= 0) return GFC_ERR_UNSUPPORTED;
  if ((long long)H * W * cin >= (1ll << 31)) return GFC_ERR_UNSUPPORTED;  // 32-bit pixel offsets inside one image
  WinoArgs a = {};
  a.x = x; a.w = w_wino; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y;
  a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.tiles_x = (W + WT_X - 1) / WT_X;
  a.tiles_y = (H + WT_Y - 1) / WT_Y;
  return pool ? launch_wino<true, false>(a, (hipStream_t)stream) : launch_wino<false, false>(a, (hipStream_t)stream);
}

// conv1a (1 -> 64, direct) + conv1b (64 -> 64, Winograd) + 2x2 max-pool in one launch: image [B,H,W] -> [B,H/2,W/2,64]
extern "C" int gfc_sp_stem_wino(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                                const float* w2_wino, const float* b2, const float* s2, const float* t2, float* y, int B,
                                int H, int W, void* stream) {
  if (!image || !w1 || !b1 || !w2_wino || !b2 || !y || B <= 0 || H < 2 || W < 2) return GFC_ERR_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return GFC_ERR_INVALID;
  WinoArgs a = {};
  a.x = image; a.w = w2_wino; a.bias = b2; a.scale = s2; a.shift = t2; a.y = y;
  a.B = B; a.H = H; a.W = W; a.cin = 64; a.cout = 64; a.relu = 1;
  a.w1 = w1; a.b1 = b1; a.s1 = s1; a.t1 = t1;
  a.tiles_x = (W + WT_X - 1) / WT_X;
  a.tiles_y = (H + WT_Y - 1) / WT_Y;
  return launch_wino<true, true>(a, (hipStream_t)stream);
}
