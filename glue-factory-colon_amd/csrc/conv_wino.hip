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
#ifndef WINO_DIAG
#define WINO_DIAG 0                // diagnostic builds (tools/ab_build.sh): 1 no conv1a, 2 no B reloads, 8 no patch loads, 16 no column transform, 64 no chunk barrier, 256 phase accounting (tools/micro/wino_timeline.py), 512 no output stores
#endif

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
  int xcd_remap;  // walk the work items in XCD-contiguous order (runtime.h: GFC_XCD_REMAP)
#if WINO_DIAG & 256
  unsigned long long* diag;  // diagnostic build (tools/micro/wino_timeline.py): 8 words per wave
#endif
};
#if WINO_DIAG & 256
static unsigned long long* g_wino_diag = nullptr;
extern "C" void gfc_diag_set_wino_stamps(void* p) { g_wino_diag = (unsigned long long*)p; }
#define WINO_T(v_) const unsigned long long v_ = __builtin_readcyclecounter()
#else
#define WINO_T(v_) do {} while (0)
#endif

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
  float* in_s = smem;               // [2][WPATCH]; the epilogue exchange buffer (16384 floats) aliases it
  float* img_s = smem + 16384;      // STEM: [WIM_R][WIM_C] image patch (behind the exchange buffer: it is filled for the
  float* c1_s = img_s + WIM_R * WIM_C;  //   next item while the epilogue runs); conv1a w [9][64], b [64], s [64], t [64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, kept scalar: everything derived from it
  const int l31 = lane & 31, h = lane >> 5;                 // (filter stream base, transform rows) stays in SGPRs
  const int cin = a.cin, nchunks = cin / WKC;

  // Persistent workgroups: a workgroup walks the work items blockIdx.x, blockIdx.x + gridDim.x, ...  Item order
  // (XCD-aware, common.h: gfc_xcd_chunk -- gridDim.x is a multiple of 8 whenever the loop runs more than once): each
  // XCD owns a contiguous run of items = whole images; inside the run a PAIR of output-channel blocks is the fastest
  // index, then the tile column, the tile row, the image and last the remaining output-channel blocks -- the two
  // channel blocks of a tile and its neighbours' halos are read while the tile's input is still in that XCD's L2
  // (all blocks fastest would put cout/64 x 0.5 MB of filters into the L2 working set of the 512-channel head layer).
  // The first input patch, the first filter fragments (and, in the stem, the
  // image patch) of the NEXT item are requested during the last chunk of the current one and land under its epilogue.
  // Exit: the item index is a pure function of blockIdx / gridDim (no queue, no inter-workgroup dependency).
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  const int nitems = ntiles * (a.cout / 64);
  const int nbpair = (a.cout / 64) % 2 == 0 ? 2 : 1;
  int item = blockIdx.x;
  int x0, y0, b, nb;
  const float* xin;
  const float4* wp;  // B stream: fragments of this wave's four positions, contiguous per k group (8 x 64 float4)
#define WINO_DECODE(w_, x0_, y0_, b_, nb_, xin_, wp_)                                              \
  do {                                                                                             \
    unsigned l_ = a.xcd_remap ? gfc_xcd_chunk((unsigned)(w_), (unsigned)nitems) : (unsigned)(w_);  \
    const int lo_ = (int)(l_ % (unsigned)nbpair);                                                  \
    l_ /= (unsigned)nbpair;                                                                        \
    int t_ = (int)(l_ % (unsigned)ntiles);                                                         \
    nb_ = (int)(l_ / (unsigned)ntiles) * nbpair + lo_;                                             \
    x0_ = (t_ % a.tiles_x) * WT_X;                                                                 \
    t_ /= a.tiles_x;                                                                               \
    y0_ = (t_ % a.tiles_y) * WT_Y;                                                                 \
    b_ = t_ / a.tiles_y;                                                                           \
    xin_ = a.x + (size_t)b_ * a.H * a.W * (STEM ? 1 : cin);                                        \
    wp_ = reinterpret_cast<const float4*>(a.w) + (((size_t)nb_ * 4 + xi) * (cin / 8)) * 512;        \
  } while (0)
  WINO_DECODE(item, x0, y0, b, nb, xin, wp);

  // ---- patch staging: 18 x 10 pixels x 4 float4 = 720 float4 per chunk, three per thread ----
  float4 ireg[3];
  int st_off[3];   // LDS offset (floats) of this thread's three pieces; -1 = none
  int st_gofs[3];  // global offset (floats, without the chunk term) or -1 when the pixel lies outside the image
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i;
    const int pix = idx >> 2, c4 = idx & 3;
    const int py = pix / 10, px = pix - py * 10;
    const int R = (py >> 1) + (py & 1) * 9, C = (px >> 1) + (px & 1) * 5;
    st_off[i] = idx < 720 ? (R * WP_C + C) * WKC + ((c4 ^ (R & 3)) << 2) : -1;
  }
#define WINO_GOFS(y0_, x0_)                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                               \
    const int py_ = ((tid + 256 * i_) >> 2) / 10, px_ = ((tid + 256 * i_) >> 2) - py_ * 10;        \
    const int gy_ = (y0_) - 1 + py_, gx_ = (x0_) - 1 + px_;                                        \
    const bool in_ = st_off[i_] >= 0 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;            \
    if constexpr (STEM) st_gofs[i_] = in_ ? (py_ * WIM_C + px_) : -1; /* image-patch offset, tap (0,0) */ \
    else st_gofs[i_] = in_ ? (int)(((size_t)gy_ * a.W + gx_) * cin + ((tid & 3) << 2)) : -1;       \
  }
#define WINO_LOAD_IN(xin_, chunk_)                                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                               \
    float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
    if (st_gofs[i_] >= 0 && !(WINO_DIAG & 8)) v_ = *reinterpret_cast<const float4*>((xin_) + st_gofs[i_] + (chunk_) * WKC); \
    ireg[i_] = v_;                                                                                 \
  }
#define WINO_FILL_IN(chunk_)                                                                      \
  {                                                                                                \
    /* STEM: conv1a on this thread's three patch pieces.  The pieces share their channel quad (256 = 0 mod 4):   \
       the 12 constant quads are read from LDS once per chunk (re-reading them per piece costs 10 % of the       \
       kernel: the LDS pipe is shared with the fragment reads of all eight resident waves). */                 \
    const int c0_ = (chunk_) * WKC + ((tid & 3) << 2);                                             \
    float4 wv_[9];                                                                                 \
    _Pragma("unroll") for (int t2_ = 0; t2_ < 9; ++t2_)                                            \
        wv_[t2_] = *reinterpret_cast<const float4*>(c1_s + t2_ * 64 + c0_);                        \
    const float4 b1_ = *reinterpret_cast<const float4*>(c1_s + 576 + c0_);                         \
    const float4 s1_ = *reinterpret_cast<const float4*>(c1_s + 640 + c0_);                         \
    const float4 t1_ = *reinterpret_cast<const float4*>(c1_s + 704 + c0_);                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                             \
      float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);  /* outside the image: conv1b's zero padding */ \
      if (st_gofs[i_] >= 0 && !(WINO_DIAG & 1)) {                                                  \
        const float* ip_ = img_s + st_gofs[i_];                                                    \
        _Pragma("unroll") for (int t2_ = 0; t2_ < 9; ++t2_) {                                      \
          const float f_ = ip_[(t2_ / 3) * WIM_C + t2_ % 3];                                       \
          v_.x = fmaf(f_, wv_[t2_].x, v_.x); v_.y = fmaf(f_, wv_[t2_].y, v_.y);                    \
          v_.z = fmaf(f_, wv_[t2_].z, v_.z); v_.w = fmaf(f_, wv_[t2_].w, v_.w);                    \
        }                                                                                          \
        v_.x = fmaxf(v_.x + b1_.x, 0.f) * s1_.x + t1_.x; v_.y = fmaxf(v_.y + b1_.y, 0.f) * s1_.y + t1_.y; \
        v_.z = fmaxf(v_.z + b1_.z, 0.f) * s1_.z + t1_.z; v_.w = fmaxf(v_.w + b1_.w, 0.f) * s1_.w + t1_.w; \
      }                                                                                            \
      ireg[i_] = v_;                                                                               \
    }                                                                                              \
  }
#define WINO_STORE_IN(buf_)                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                                 \
    if (st_off[i_] >= 0) *reinterpret_cast<float4*>(in_s + (buf_) * WPATCH + st_off[i_]) = ireg[i_];
#define WINO_IMG(xin_, y0_, x0_)                                                                   \
  ((tid < WIM_R * WIM_C && (y0_) - 2 + tid / WIM_C >= 0 && (y0_) - 2 + tid / WIM_C < a.H &&        \
    (x0_) - 2 + tid % WIM_C >= 0 && (x0_) - 2 + tid % WIM_C < a.W)                                 \
       ? (xin_)[(size_t)((y0_) - 2 + tid / WIM_C) * a.W + (x0_) - 2 + tid % WIM_C] : 0.f)

  WINO_GOFS(y0, x0);
  if constexpr (STEM) {
    if (tid < WIM_R * WIM_C) img_s[tid] = WINO_IMG(xin, y0, x0);
    for (int i = tid; i < 768; i += 256)
      c1_s[i] = i < 576 ? a.w1[i] : i < 640 ? a.b1[i - 576] : i < 704 ? (a.s1 ? a.s1[i - 640] : 1.f)
                                                                         : (a.t1 ? a.t1[i - 704] : 0.f);
    __syncthreads();
    WINO_FILL_IN(0);
  } else {
    WINO_LOAD_IN(xin, 0);
  }
  float4 bq[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bq[nu][nt] = wp[(nu * 2 + nt) * 64 + lane];

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

#if WINO_DIAG & 256
  unsigned long long d_k = 0, d_x = 0, d_s = 0, d_n = 0, d_epi10 = 0;
  const unsigned long long d_t0 = __builtin_readcyclecounter();
#endif
  while (true) {
  WINO_T(t_a);
  WINO_STORE_IN(0);
  __syncthreads();
  f32x16 acc[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][nt][r] = 0.f;
  const int next_item = item + gridDim.x;
  // (the stem keeps one workgroup per item: with the next-item state live next to conv1a's 36 constant registers the
  // kernel spills, and conv1a with fewer live constants re-reads them from LDS: -6..-10 % either way)
  const bool more = !STEM && next_item < nitems;
  int nx0 = 0, ny0 = 0, nbb = 0, nnb = 0;
  const float* nxin = xin;
  const float4* nwp = wp;
  if (more) WINO_DECODE(next_item, nx0, ny0, nbb, nnb, nxin, nwp);

  for (int c = 0; c < nchunks; ++c) {
    const bool has_next = c + 1 < nchunks;
    const float* ps = in_s + (c & 1) * WPATCH;
#if WINO_DIAG & 128  // diagnostic: request the next patch at the top of the chunk instead of after group 0
    if constexpr (!STEM) {
      if (has_next) { WINO_LOAD_IN(xin, c + 1); }
      else if (more) { WINO_GOFS(ny0, nx0); WINO_LOAD_IN(nxin, 0); }
    }
#endif
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
      if (WINO_DIAG & 16) {  // diagnostic: fragments without the column transform
        v[0] = t0; v[1] = t1; v[2] = t2; v[3] = t3;
      } else {
        v[0] = f4_sub(t0, t2);
        v[1] = f4_add(t1, t2);
        v[2] = f4_sub(t2, t1);
        v[3] = f4_sub(t1, t3);
      }
      const int kg_next = 2 * c + g + 1;
      // next fragments: the following k group of this item, or group 0 of the next item
      const bool more_b = (kg_next < cin / 8 || more) && !(WINO_DIAG & 2);
      const float4* bsrc = kg_next < cin / 8 ? wp + (size_t)kg_next * 512 : nwp;
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
          bq[nu][0] = bsrc[(nu * 2 + 0) * 64 + lane];
          bq[nu][1] = bsrc[(nu * 2 + 1) * 64 + lane];
        }
      }
      if (g == 0) {
        // requested AFTER this group's fragment loads (vmcnt retires in order: the fragment waits of the next
        // group must not include these), stored to LDS at the end of the chunk
        if constexpr (STEM) {
          if (has_next) { WINO_FILL_IN(c + 1); }  // VALU work under the MFMAs of this chunk
        } else {
#if !(WINO_DIAG & 128)
          if (has_next) { WINO_LOAD_IN(xin, c + 1); }
          else if (more) { WINO_GOFS(ny0, nx0); WINO_LOAD_IN(nxin, 0); }
#endif
        }
      }
    }
    if (has_next) WINO_STORE_IN((c + 1) & 1);
    if (!(WINO_DIAG & 64)) __syncthreads();
  }

  WINO_T(t_b);
  // ---- output transform.  Column direction (nu) lane-local: z0 = M0 + M1 + M2, z1 = M1 - M2 - M3 ----
  // exchange buffer [xi 4][j 2][tile 32][cout 64]: written from the accumulator layout (cout on the lane: 128-byte
  // runs), read back with 4 consecutive channels per lane (ds_read_b128; 16 lanes = the 256 contiguous bytes of one
  // pixel's channel block; conflict-free at a 64-float pitch)
  float* ex = smem;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      const int o = acc_row(r, h) * 64 + nt * 32 + l31;
      ex[(xi * 2 + 0) * 2048 + o] = (m0 + m1) + m2;
      ex[(xi * 2 + 1) * 2048 + o] = (m1 - m2) - m3;
    }
  __syncthreads();
  WINO_T(t_c);
  // row direction (xi) across the waves, then bias / ReLU / BN / pool: thread -> channel quad q of tiles t, t + 16
  const int q4 = (tid & 15) * 4;
  const int co = nb * 64 + q4;
  const float4 bi = *reinterpret_cast<const float4*>(a.bias + co);
  const float4 sc = a.scale ? *reinterpret_cast<const float4*>(a.scale + co) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sh = a.shift ? *reinterpret_cast<const float4*>(a.shift + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool relu = a.relu != 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int tile = (tid >> 4) + 16 * k;  // Winograd tile (tile >> 2, tile & 3)
    const float* zp = ex + tile * 64 + q4;
    float4 yv[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 z0 = *reinterpret_cast<const float4*>(zp + (0 * 2 + j) * 2048);
      const float4 z1 = *reinterpret_cast<const float4*>(zp + (1 * 2 + j) * 2048);
      const float4 z2 = *reinterpret_cast<const float4*>(zp + (2 * 2 + j) * 2048);
      const float4 z3 = *reinterpret_cast<const float4*>(zp + (3 * 2 + j) * 2048);
      yv[0][j] = f4_add(f4_add(z0, z1), z2);
      yv[1][j] = f4_sub(f4_sub(z1, z2), z3);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 t = f4_add(yv[i][j], bi);
        if (relu) t = make_float4(fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f));
        yv[i][j] = make_float4(t.x * sc.x + sh.x, t.y * sc.y + sh.y, t.z * sc.z + sh.z, t.w * sc.w + sh.w);
      }
    const int oy = (y0 >> 1) + (tile >> 2), ox = (x0 >> 1) + (tile & 3);
    if constexpr (POOL) {
      const int Ho = a.H >> 1, Wo = a.W >> 1;
      if (oy < Ho && ox < Wo) {
        const float4 u = yv[0][0], v = yv[0][1], w = yv[1][0], z = yv[1][1];
        if (!(WINO_DIAG & 512) || oy < 0)  // diagnostic 512: no output stores
        *reinterpret_cast<float4*>(a.y + (((size_t)b * Ho + oy) * Wo + ox) * a.cout + co) =
            make_float4(fmaxf(fmaxf(u.x, v.x), fmaxf(w.x, z.x)), fmaxf(fmaxf(u.y, v.y), fmaxf(w.y, z.y)),
                        fmaxf(fmaxf(u.z, v.z), fmaxf(w.z, z.z)), fmaxf(fmaxf(u.w, v.w), fmaxf(w.w, z.w)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gy = 2 * oy + i, gx = 2 * ox + j;
          if (gy < a.H && gx < a.W && (!(WINO_DIAG & 512) || gy < 0))  // diagnostic 512: no output stores
            *reinterpret_cast<float4*>(a.y + (((size_t)b * a.H + gy) * a.W + gx) * a.cout + co) = yv[i][j];
        }
    }
  }
#if WINO_DIAG & 256
  {
    WINO_T(t_d);
    d_k += t_b - t_a; d_x += t_c - t_b; d_s += t_d - t_c; d_n += 1;
    if (d_n == 10) d_epi10 = t_b;  // absolute start of the 10th item's epilogue (phase of co-resident workgroups)
  }
#endif
  if (!more) break;
  item = next_item; x0 = nx0; y0 = ny0; b = nbb; nb = nnb; xin = nxin; wp = nwp;
  __syncthreads();  // the exchange buffer has been read: the next item's first patch may land in LDS
  }  // persistent loop over work items
#if WINO_DIAG & 256
  if (a.diag && lane == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = a.diag + ((size_t)blockIdx.x * 4 + xi) * 8;
    o[0] = d_k; o[1] = d_x; o[2] = d_s; o[3] = d_n; o[4] = __builtin_readcyclecounter() - d_t0;
    o[5] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32); o[6] = d_t0; o[7] = d_epi10;
  }
#endif
#undef WINO_DECODE
#undef WINO_GOFS
#undef WINO_LOAD_IN
#undef WINO_FILL_IN
#undef WINO_STORE_IN
#undef WINO_IMG
}

template <bool POOL, bool STEM>
static int launch_wino(const WinoArgs& a, hipStream_t st) {
  static_assert(2 * WPATCH <= 16384, "the exchange buffer covers the patch buffers");
  constexpr size_t lds = (size_t)(16384 + (STEM ? WIM_R * WIM_C + 768 : 0)) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};
  if (lds > 64 * 1024) gfc_allow_dynamic_lds((const void*)conv3x3_wino_kernel<POOL, STEM>, lds, lds_ok);
  const long long nitems = (long long)a.tiles_x * a.tiles_y * a.B * (a.cout / 64);
  long long resident = ((long long)gfc_device_cus() * 2) & ~7ll;  // two workgroups per CU (registers, 64-68 KB LDS);
  if (resident < 8) resident = 8;                               // a multiple of 8: the XCD label of an item = item % 8
  const int forced = gfc_knobs().conv_persist;                 // GFC_CONV_PERSIST=0: one workgroup per item
  const long long grid = (STEM || forced == 0 || nitems < resident) ? nitems : resident;
  WinoArgs wa = a;
  wa.xcd_remap = gfc_knobs().xcd_remap != 0 && nitems >= 16;
#if WINO_DIAG & 256
  wa.diag = g_wino_diag;
#endif
  hipLaunchKernelGGL((conv3x3_wino_kernel<POOL, STEM>), dim3((unsigned)grid), dim3(256), lds, st, wa);
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
