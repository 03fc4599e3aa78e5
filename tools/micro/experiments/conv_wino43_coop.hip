// SuperPoint stem (conv1a 1->64 + conv1b 64->64 + 2x2 max-pool) with conv1b as Winograd F(4x4, 3x3) on the fp32 matrix
// pipe, and conv1a on the matrix pipe as well.
//
// Same function as gfc_sp_stem / gfc_sp_stem_wino (reference gluefactory/models/extractors/superpoint_open.py:61-77,
// 100-108 and gluefactory_nonfree/superpoint.py:214-218: conv3x3 -> ReLU [-> BatchNorm(eval)] twice, MaxPool2d(2,2)):
//     Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        with 6x6 input tiles, 4x4 output tiles:
// 36 multiplications per 16 outputs and input channel instead of 64 (F(2x2,3x3)) or 144 (direct).  Every product is an
// fp32 product accumulated in fp32 (v_mfma_f32_32x32x2_f32); the filter transform is evaluated once in float64.
// F(4x4,3x3) has larger transform coefficients than F(2x2,3x3); the decision experiment
// (tools/micro/winograd_f43_numerics.py, profiles/r03_winograd_f43_numerics.txt) shows that for conv1b ALONE the
// whole-stack heat-map error stays at the direct fp32 convolution's level (<= 5.7e-6 vs 4.9e-6, 0 key-point flips on
// every case) while for deeper layers it does not -- so only the stem uses it.
//
// Mapping.  One workgroup = 12 waves = one 32 x 16 pixel output tile = 8 x 4 Winograd tiles (= one MFMA M tile: lane
// <-> Winograd tile) x 64 output channels.  The 36 transform positions (xi, nu) are 36 independent GEMMs
// M[xi,nu][tile, cout] = V[xi,nu][tile, cin] . U[xi,nu][cin, cout]; wave (xi, nt) owns the six positions (xi, 0..5) of
// output-channel tile nt: 6 accumulators.
//   * conv1a: the 18 x 34 halo patch of conv1b's input is a GEMM [612 pixels] x [9 taps + bias] x [64 channels]: M tiles
//     of 32 pixels, A = image taps read from a 20 x 36 image patch in LDS, B = conv1a's weights (tap 9 = the bias with
//     A = 1), 5 MFMAs per 32 pixels x 32 channels, the same k-ordered fmaf chain the VALU form evaluates (bit-identical).
//     ReLU / BN affine on the accumulators, then written as 16-channel chunks into three LDS patch buffers.
//   * A operand: a wave builds its fragments on the fly from the patch, two channels (one ds_read_b64) at a time so
//     that the transform temporaries stay small (12 waves per CU leave 168 registers per lane, 96 of them
//     accumulators): row transform xi (3 or 4 patch rows) of the six columns, accumulated straight into the six column
//     transforms; 36 / 48 ds_read_b64 per 8-deep k group and 24 MFMAs.  A patch buffer is two planes of 8 channels
//     ([plane][pixel][8]); patch columns are stored de-interleaved modulo 4 ((q & 3) * 9 + (q >> 2)) and the four
//     8-byte channel pairs of a pixel XOR-swizzled by (row >> 2) & 3: the 32 lanes of a ds_read_b64 group (4 tile
//     rows x 8 consecutive tile columns) hit 32 distinct bank pairs.
//   * B operand: transformed filters in MFMA-fragment order, streamed from L2 (3 KB per wave and half k group as six
//     coalesced 512-byte loads straight into registers, requested before the half's transforms and consumed after
//     them; the same 576 KB for every work item).
//   * Epilogue: column transform lane-local (6 accumulators -> 4 values), row transform across the six xi waves
//     through a 96 KB LDS exchange (one output-channel tile per pass, aliasing the patch buffers); bias, ReLU, BN
//     affine, 2x2 max-pool (a 4x4 Winograd tile = four pooling windows), NHWC store.
// Persistent workgroups, one per CU (130 KB LDS); the next item's image patch is requested two chunks ahead.
#include "common.h"

#define S4_TW 32                               // output tile width
#define S4_TH 16                               // output tile height
#define S4_PR 18                               // patch rows
#define S4_PC 34                               // patch columns
#define S4_PITCH 36                            // pixel slots per patch row (de-interleaved column positions < 36)
#define S4_KC 16                               // channels per chunk
#define S4_PLANE (S4_PR * S4_PITCH * 8)        // floats per 8-channel plane (5184)
#define S4_PATCH (2 * S4_PLANE)                // floats per patch buffer (10368)
#define S4_NPIX (S4_PR * S4_PC)                // 612
#define S4_IMR 20                              // image patch rows
#define S4_IMC 36                              // image patch columns
#define S4_EX (6 * 2 * 32 * 32)                // exchange buffer floats (one cout tile, one column pair): [xi][jj][tile][cout]
#define S4_THREADS 768
#ifndef S4_COOP
#define S4_COOP 1  // 1: cooperative input transform through an LDS fragment buffer (round 4); 0: every wave builds its own fragments
#endif
#define S4_VFLOATS (6 * 2 * 32 * 14)           // fragment buffer V[xi][h][tile][14]: six channel pairs per lane, 14-float pitch
#ifndef S4_DIAG
#define S4_DIAG 0  // diagnostic builds (tools/ab_build.sh WORKTREE s4 "-DS4_DIAG=1"): 1 = per-wave phase accounting (tools/micro/stem43_timeline.py); ablations with wrong results: 2 no conv1a stores, 4 no conv1a units inside the chunks, 8 no patch reads, 16 no filter-fragment loads, 32 no epilogue
#endif
#if S4_DIAG & 1
static unsigned long long* g_s4_diag = nullptr;
extern "C" void gfc_diag_set_stem43_stamps(void* p) { g_s4_diag = (unsigned long long*)p; }
#define S4_T(v_) const unsigned long long v_ = __builtin_readcyclecounter()
#else
#define S4_T(v_) do {} while (0)
#endif

struct Stem43Args {
  const float* x;      // image [B,H,W]
  const float* w;      // conv1b filters packed by gfc_pack_conv3x3_wino43
  const float* bias;   // conv1b
  const float* scale;  // nullable (no BN)
  const float* shift;
  float* y;            // [B,H/2,W/2,64]
  int B, H, W;
  int tiles_x, tiles_y;
  const float* w1;  // conv1a [9][64]
  const float* b1;
  const float* s1;  // nullable
  const float* t1;
#if S4_DIAG & 1
  unsigned long long* diag;  // 16 words per wave
#endif
};

// U = G g G^T (6x6) in float64, rounded once; scattered into MFMA-fragment order:
//   out[xi][kg][nt][hh][nu][lane = 32 h + l31][e]  <-  U[xi][nu] of (cout = 32 nt + l31, cin = 8 kg + 4 h + 2 hh + e)
// (a wave reads the six positions' fragments of one half k group = channels 2 hh, 2 hh + 1 of every lane's four as
// six consecutive 512-byte runs)
__global__ void pack_conv3x3_wino43_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cout * cin) return;
  const int ci = idx % cin, co = idx / cin;
  double g[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) g[r][c] = (double)w[((size_t)co * cin + ci) * 9 + r * 3 + c];
  const double G[6][3] = {{1.0 / 4, 0.0, 0.0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  double t[6][3];
  for (int i = 0; i < 6; ++i)
    for (int c = 0; c < 3; ++c) t[i][c] = G[i][0] * g[0][c] + G[i][1] * g[1][c] + G[i][2] * g[2][c];
  const int nt = co / 32, l31 = co % 32;
  const int kg = ci / 8, h = (ci % 8) / 4, s = ci % 4;
  const int lane = 32 * h + l31;
  const int nkg = cin / 8, ntn = cout / 32;
  for (int xi = 0; xi < 6; ++xi)
    for (int nu = 0; nu < 6; ++nu) {
      const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
      const size_t o = ((((((size_t)xi * nkg + kg) * ntn + nt) * 2 + (s >> 1)) * 6 + nu) * 64 + lane) * 2 + (s & 1);
      out[o] = (float)u;
    }
}

extern "C" int gfc_pack_conv3x3_wino43(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
  if (!w_oihw || !w_packed || cout != 64 || cin != 64) return GFC_ERR_INVALID;  // the stem's conv1b only
  const int total = cout * cin;
  hipLaunchKernelGGL(pack_conv3x3_wino43_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_packed, cout, cin);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

__device__ __forceinline__ float4 f4_scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 b) {
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_addv(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_subv(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_max(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f v2_fma(float s, v2f a, v2f b) { return v2f{fmaf(s, a.x, b.x), fmaf(s, a.y, b.y)}; }
__device__ __forceinline__ v2f v2_mul(float s, v2f a) { return v2f{s * a.x, s * a.y}; }
// LDS read of one channel pair.  volatile: hipcc would otherwise fuse neighbouring reads into ds_read2_b64, which is
// banked modulo 32 in 16-lane groups (2-way conflicts on this layout) at half the bandwidth of ds_read_b64
typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;
__device__ __forceinline__ v2f lds_pair(const float* p) {
#if S4_DIAG & 8
  return v2f{(float)(size_t)p, 1.f};
#else
  return *(lds_v2f_ptr)(p);
#endif
}

__global__ __launch_bounds__(S4_THREADS, 1) void stem_wino43_kernel(Stem43Args args) {
  // (plain locals: the lambdas below capture by reference, and a by-value kernel argument struct whose address is
  // taken is copied to private memory)
  const float* const a_x = args.x;
  const float* const a_w = args.w;
  float* const a_y = args.y;
  const int a_B = args.B, a_H = args.H, a_W = args.W, a_tiles_x = args.tiles_x, a_tiles_y = args.tiles_y;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // patch buffers A, B and C; C is also the head of the exchange buffer X (12288 floats).  Chunk -> buffer of an
  // item: 0 -> A, 1 -> B, 2 -> A, 3 -> C; the NEXT item's chunks 0, 1 are produced during chunk 3 into A, B and
  // survive the epilogue, which only touches X.
  float* bufA = smem;
  float* bufB = smem + S4_PATCH;
  float* bufC = smem + 2 * S4_PATCH;
  float* img_s = smem + 2 * S4_PATCH + S4_EX;  // [S4_IMR][S4_IMC]
  float* c1_s = img_s + S4_IMR * S4_IMC;       // conv1a w [9][64], b [64], s [64], t [64]
  float* c2_s = c1_s + 768;                    // conv1b bias [64], scale [64], shift [64]
  float* v_s = c2_s + 192;                     // S4_COOP: the fragment buffer

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave % 6, nt = wave / 6;  // scalar: transform row, output-channel tile
  const int phase = wave >> 2;             // waves w, w + 4, w + 8 share a SIMD: they take turns at conv1a
  const int l31 = lane & 31, h = lane >> 5;
  const int ty = l31 >> 3, tx = l31 & 7;

  const int per_img = a_tiles_x * a_tiles_y;
  const int nitems = per_img * a_B;

  // conv1a constants -> LDS once (persistent workgroup); visible after the first barrier below
  for (int i = tid; i < 768; i += S4_THREADS)
    c1_s[i] = i < 576 ? args.w1[i] : i < 640 ? args.b1[i - 576] : i < 704 ? (args.s1 ? args.s1[i - 640] : 1.f)
                                                                       : (args.t1 ? args.t1[i - 704] : 0.f);

  if (tid < 192)
    c2_s[tid] = tid < 64 ? args.bias[tid] : tid < 128 ? (args.scale ? args.scale[tid - 64] : 1.f) : (args.shift ? args.shift[tid - 128] : 0.f);

  // image patch of an item: thread tid < 720 owns pixel (tid / 36, tid % 36) of the 20 x 36 patch
  auto load_img = [&](int it) __attribute__((always_inline)) -> float {
    int t_ = tid;
    asm volatile("" : "+v"(t_));  // (opaque: the patch coordinates are recomputed here, not kept in registers per item)
    const int im_r = (t_ * 1821) >> 16, im_c = t_ - im_r * S4_IMC;  // t / 36 for t < 768
    if (t_ >= S4_IMR * S4_IMC || it >= nitems) return 0.f;
    const int b = it / per_img, t = it - b * per_img;
    const int gy = (t / a_tiles_x) * S4_TH - 2 + im_r, gx = (t % a_tiles_x) * S4_TW - 2 + im_c;
    return (gy >= 0 && gy < a_H && gx >= 0 && gx < a_W) ? a_x[((size_t)b * a_H + gy) * a_W + gx] : 0.f;
  };

  // ---- B stream: this wave's six fragments per half k group, contiguous 3 KB; identical for every item ----
  // (scalar base + 32-bit lane offset: no 64-bit per-lane pointer in registers)
  const float2* wbase = reinterpret_cast<const float2*>(a_w) + (size_t)(xi * 32 + nt * 2) * 6 * 64;
  // half k group hk = 2 kg + hh (16 per item): fragment block index ((xi * 8 + kg) * 2 + nt) * 2 + hh
  auto bfrag = [&](int hk, int nu) __attribute__((always_inline)) -> float2 {
    const float2* blk = wbase + (size_t)((hk >> 1) * 4 + (hk & 1)) * 6 * 64;  // wave-uniform
    return blk[nu * 64 + lane];
  };

  // ---- A fragments: row transform xi of patch rows 4 ty + rho ----
  //   three rows:  xi = 0: 4 d0 - 5 d2 + d4            xi = 5: 4 d1 - 5 d3 + d5          = (c1 dQ + c0 dP) + dR
  //   four rows:   xi = 1: -4 (d1 + d2) + (d3 + d4)    xi = 2: 4 (d1 - d2) + (d4 - d3)
  //                xi = 3:  2 (d3 - d1) + (d4 - d2)    xi = 4: 2 (d1 - d3) + (d4 - d2)   = c0 (dP + c1 dQ) + (dR + c2 dS)
  // rows P, Q, R, S per xi (rho = 4, 5 carry the other swizzle):
  const bool three = xi == 0 || xi == 5;
  const int rP = three ? (xi == 0 ? 0 : 1) : (xi == 3 ? 3 : 1);
  const int rQ = three ? rP + 2 : (xi == 1 || xi == 2 ? 2 : (xi == 3 ? 1 : 3));
  const int rR = three ? rP + 4 : (xi == 1 ? 3 : 4);
  const int rS = xi == 1 ? 4 : (xi == 2 ? 3 : 2);
  const float c0 = three ? 4.f : (xi == 1 ? -4.f : xi == 2 ? 4.f : 2.f);
  const float c1 = three ? -5.f : (xi == 1 ? 1.f : -1.f);
  const float c2 = xi == 1 ? 1.f : -1.f;
  // float offsets inside a plane: pixel (4 ty + rho, de-interleaved column tx + COL[j]) x 8 floats, channel pair
  // 2 h + hh swizzled by ((4 ty + rho) >> 2) & 3 = ty for rho < 4 and (ty + 1) & 3 for rho = 4, 5
  const int pix0 = (4 * ty * S4_PITCH + tx) * 8;
  const int offA = pix0 + 2 * ((2 * h) ^ ty), offB = pix0 + 2 * ((2 * h) ^ ((ty + 1) & 3));  // half hh = 1: ^ 2
  constexpr int COLOFF[6] = {0, 9 * 8, 18 * 8, 27 * 8, 1 * 8, 10 * 8};
  const int ro0 = rP * S4_PITCH * 8, ro1 = rQ * S4_PITCH * 8, ro2 = rR * S4_PITCH * 8, ro3 = rS * S4_PITCH * 8;
  const bool hiR = rR >= 4, hiS = rS >= 4;  // rho = 4, 5: swizzle B

#if S4_DIAG & 1
  unsigned long long dg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long dg_t0 = __builtin_readcyclecounter();
#endif
  // ---- conv1a on the matrix pipe: one unit = 32 patch pixels x 32 channels (two chunks) ----
  //   A[pixel][k]: image taps (k = 9: constant 1 -> bias), B[k][channel]: weights; k = 2 s + h in MFMA step s.
  //   Units 0..17: patch row u, columns 0..31 (row and in-image test scalar, LDS offsets compile-time);
  //   units 18, 19: the 36 pixels of columns 32, 33 (pixel index 32 (u - 18) + m -> row idx >> 1, column 32 + (idx & 1)).
  auto conv1a_unit = [&](int u, int ntile, int y0, int x0, float* dst_lo, float* dst_hi) __attribute__((always_inline)) {
    // (lane index made opaque: everything derived from it below is recomputed here instead of being hoisted out of
    // the item loop, where ~20 loop-invariant per-lane values would sit in registers the k loop needs)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    S4_T(tc0);
    const int l31 = ln & 31, h = ln >> 5;
    const int ch = 32 * ntile + l31;
    const float* cb = c1_s + h * 64 + ch;  // row k = 2 s + h of [9 taps | bias][64]: cb[s * 128]
    const bool edge = u >= 18;
    const int idx_in = min(32 * (u - 18) + l31, 35);
    const float* ip = img_s + (edge ? (idx_in >> 1) * S4_IMC + 32 + (idx_in & 1) : u * S4_IMC + l31);
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    // tap k = 2 s + h at image-patch offset (k / 3) * 36 + k % 3; k = 9 (s = 4, h = 1): constant 1 -> the bias.
    // Both candidates are read at compile-time offsets and selected by h (no per-lane offset registers).
    constexpr int T0[5] = {0, 2, S4_IMC + 1, 2 * S4_IMC, 2 * S4_IMC + 2};
    constexpr int T1[5] = {1, S4_IMC, S4_IMC + 2, 2 * S4_IMC + 1, 2 * S4_IMC + 2};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float a0 = ip[T0[s]], a1 = s == 4 ? 1.f : ip[T1[s]];
      d = mfma32(h ? a1 : a0, cb[s * 128], d);
    }
    const float sc = c1_s[640 + ch], sh = c1_s[704 + ch];
#if S4_DIAG & 1
    float probe_ = fmaxf(d[0], 0.f);  // waits for the last MFMA
    asm volatile("" : "+v"(probe_));
    S4_T(tc1);
    dg[8] += tc1 - tc0;
#endif
    float* dst = (l31 & 16) ? dst_hi : dst_lo;  // channels 16..31 of the unit belong to the second chunk
    const int c16 = l31 & 15;
    const int cofs = (c16 >> 3) * S4_PLANE + (c16 & 1), cpr = (c16 >> 1) & 3;
    if (!edge) {
      // pixel (u, q = 4 h + rr) with rr = (r & 3) + 8 (r >> 2): de-interleaved column (rr & 3) * 9 + (rr >> 2) + h
      const int gy = y0 - 1 + u;
      const bool row_in = gy >= 0 && gy < a_H;
      float* o = dst + cofs + (u * S4_PITCH + h) * 8 + 2 * (cpr ^ ((u >> 2) & 3));
      if (row_in && x0 >= 1 && x0 + 32 < a_W) {
        // interior (scalar test, almost every unit): no per-pixel bounds test
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
#if S4_DIAG & 2  // ablation: no conv1a stores (wrong results)
          if (d[r] == 12345.678f)
#endif
          o[((rr & 3) * 9 + (rr >> 2)) * 8] = fmaxf(d[r], 0.f) * sc + sh;
        }
      } else {
        const int gx0 = x0 - 1 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          // outside the image: conv1b's zero padding (NOT conv1a of a zero image)
          const float v = (row_in && (unsigned)(gx0 + rr) < (unsigned)a_W) ? fmaxf(d[r], 0.f) * sc + sh : 0.f;
          o[((rr & 3) * 9 + (rr >> 2)) * 8] = v;
        }
      }
    } else {
      const int ib = 32 * (u - 18) + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = ib + (r & 3) + 8 * (r >> 2);
        const bool real = idx < 36;  // beyond: the spare pixel slot 35 of patch row 17 (never read)
        const int p = real ? idx >> 1 : 17, cpos = real ? 8 + 9 * (idx & 1) : 35;
        const int gy = y0 - 1 + p, gx = x0 + 31 + (idx & 1);
        const float v = (gy >= 0 && gy < a_H && gx < a_W) ? fmaxf(d[r], 0.f) * sc + sh : 0.f;
        dst[cofs + (p * S4_PITCH + cpos) * 8 + 2 * (cpr ^ ((p >> 2) & 3))] = v;
      }
    }
  };
  // The 20 units of one channel tile over the 12 waves: unit `wave` before half k group `phase` = wave >> 2 of a chunk
  // (the three waves of a SIMD take turns), and waves 0..7 a second one (wave + 12) before the following half.
  // (Measured and dropped: three units for the waves of the three-row transforms, which have a quarter less LDS /
  // VALU work per half -- they then finish the conv1a chunks last: 5.21 -> 5.44 ms.)
  auto conv1a_at_half = [&](int half, int ntile, int y0, int x0, float* dst_lo, float* dst_hi) __attribute__((always_inline)) {
    if (half == phase) conv1a_unit(wave, ntile, y0, x0, dst_lo, dst_hi);
    else if (half == phase + 1 && wave + 12 < 20) conv1a_unit(wave + 12, ntile, y0, x0, dst_lo, dst_hi);
  };
  // ---- prologue: the first item's image patch and chunks 0, 1 ----
  int item = blockIdx.x;
  int y0, x0, b;
  {
    const float v0 = load_img(item);
    if (tid < S4_IMR * S4_IMC) img_s[tid] = v0;
    b = item / per_img;
    const int t_ = item - b * per_img;
    y0 = (t_ / a_tiles_x) * S4_TH; x0 = (t_ % a_tiles_x) * S4_TW;
    __syncthreads();
#pragma unroll 1
    for (int half = 0; half < 4; ++half) conv1a_at_half(half, 0, y0, x0, bufA, bufB);
  }
  float imv = load_img(item + gridDim.x);
  __syncthreads();

  for (; item < nitems; item += gridDim.x) {
    S4_T(t_top);
#if S4_DIAG & 1
    unsigned long long t_prev = t_top;
#endif
    const int nitem = item + gridDim.x;
    const bool have_next = nitem < nitems;
    const int nb_ = nitem / per_img, nt_ = nitem - nb_ * per_img;
    const int ny0 = (nt_ / a_tiles_x) * S4_TH, nx0 = (nt_ % a_tiles_x) * S4_TW;

    f32x16 acc[6];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;

#if S4_COOP
    // ---- cooperative input transform (round 4, second attempt on the duplicated transform) ----
    // The six positions of row xi are needed by the TWO waves (xi, nt = 0 / 1).  Instead of both building the same
    // fragments from the patch, wave (xi, cp = nt) transforms ONE channel pair of the half's two (lane = tile x channel
    // of the pair, scalar reads) and writes its six values per lane to the fragment buffer V[xi][h][tile][14] in LDS;
    // every wave then reads its A fragments (six ds_read_b64 per lane) from V.  Half hg + 1 is transformed while half hg
    // is multiplied: per half  barrier (V complete) | read fragments | barrier (V free) | transform hg + 1 | MFMAs hg.
    // The patch of half hg + 1 is complete when it is read: the conv1a units of an odd chunk run before its halves 0..2
    // only, so the barriers of its half 3 order them before the transform of the next chunk's half 0.
    auto transform_half = [&](int tgt) __attribute__((always_inline)) {
      const int tc = (tgt >> 2) & 3, g = (tgt >> 1) & 1, hh = tgt & 1;
      const float* pst = (tc & 1) == 0 ? bufA : (tc == 1 ? bufB : bufC);
      const float* pl = pst + g * S4_PLANE;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int tl = ln & 31, hc = ln >> 5, tyy = tl >> 3, txx = tl & 7;
      const int px0 = (4 * tyy * S4_PITCH + txx) * 8;
      const float* pA = pl + ((px0 + 2 * ((2 * nt) ^ tyy) + hc) ^ (2 * hh));
      const float* pB = pl + ((px0 + 2 * ((2 * nt) ^ ((tyy + 1) & 3)) + hc) ^ (2 * hh));
      const float* q0 = pA + ro0;
      const float* q1 = pA + ro1;
      const float* q2 = (hiR ? pB : pA) + ro2;
      const float* q3 = (hiS ? pB : pA) + ro3;
      float tr[6];
      if (three) {
#pragma unroll
        for (int j = 0; j < 6; ++j) tr[j] = fmaf(c1, q1[COLOFF[j]], c0 * q0[COLOFF[j]]) + q2[COLOFF[j]];
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j)
          tr[j] = fmaf(c0, fmaf(c1, q1[COLOFF[j]], q0[COLOFF[j]]), fmaf(c2, q3[COLOFF[j]], q2[COLOFF[j]]));
      }
      const float s12 = tr[1] + tr[2], d12 = tr[1] - tr[2], s34 = tr[3] + tr[4], d34 = tr[3] - tr[4];
      const float e31 = tr[3] - tr[1], e42 = tr[4] - tr[2];
      float* vo = v_s + ((xi * 2 + nt) * 32 + tl) * 14 + hc;
      vo[0] = fmaf(-5.f, tr[2], fmaf(4.f, tr[0], tr[4]));
      vo[2] = fmaf(-4.f, s12, s34);
      vo[4] = fmaf(4.f, d12, -d34);
      vo[6] = fmaf(2.f, e31, e42);
      vo[8] = fmaf(-2.f, e31, e42);
      vo[10] = fmaf(-5.f, tr[3], fmaf(4.f, tr[1], tr[5]));
    };
    if (item == (int)blockIdx.x) {  // the first item's half 0 (later items: transformed in half 15 of the item before)
      transform_half(0);
    }
#if S4_DIAG & 1
    unsigned long long cv_cycles = 0;
#endif
#pragma unroll 1
    for (int hg = 0; hg < 16; ++hg) {
      const int c = hg >> 2, half = hg & 3;
      S4_T(te);
      __syncthreads();  // V(hg) complete (and, at a chunk head, the patch buffers of the chunk before free for conv1a)
      v2f v[6];
      {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const float* vi = v_s + ((xi * 2 + (ln >> 5)) * 32 + (ln & 31)) * 14;
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) v[nu] = lds_pair(vi + 2 * nu);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();  // every wave holds its fragments: V may be overwritten
      S4_T(tha);
#if S4_DIAG & 1
      dg[5] += tha - te;
#endif
      if (hg == 8) {
        // the next item's image patch (conv1a of this item finished reading img_s in chunk 1; the next reader is the
        // conv1a in chunk 3), then the request for the item after it
        if (tid < S4_IMR * S4_IMC) img_s[tid] = imv;
        imv = load_img(nitem + gridDim.x);
      }
      if ((c & 1) && (c == 1 || have_next) && !(S4_DIAG & 4)) {
        S4_T(tu);
        conv1a_at_half(half, c == 1 ? 1 : 0, c == 1 ? y0 : ny0, c == 1 ? x0 : nx0, bufA, c == 1 ? bufC : bufB);
#if S4_DIAG & 1
        dg[3] += __builtin_readcyclecounter() - tu;
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      float2 bq[6];
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) bq[nu] = (S4_DIAG & 16) ? make_float2(1.f + nu, 0.5f) : bfrag(hg, nu);
      S4_T(thx);
      if (hg < 15 || have_next) transform_half(hg + 1);
      __builtin_amdgcn_sched_barrier(0);
#if S4_DIAG & 1
      S4_T(thb);
#endif
#pragma unroll
      for (int nu = 0; nu < 6; nu += 2) {
        acc[nu] = mfma32(v[nu].x, bq[nu].x, acc[nu]);
        acc[nu + 1] = mfma32(v[nu + 1].x, bq[nu + 1].x, acc[nu + 1]);
        acc[nu] = mfma32(v[nu].y, bq[nu].y, acc[nu]);
        acc[nu + 1] = mfma32(v[nu + 1].y, bq[nu + 1].y, acc[nu + 1]);
      }
#if S4_DIAG & 1
      {
        S4_T(thc);
        dg[1] += thb - thx; dg[2] += thc - thb; dg[4] += thc - tha;
      }
#endif
    }
    __syncthreads();  // all patch reads of chunk 3 and all conv1a writes are behind: the exchange buffer (= buffer C) is free
#else
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      const float* ps = (c & 1) == 0 ? (c == 0 ? bufA : bufA) : (c == 1 ? bufB : bufC);
      if (c == 2) {
        // the next item's image patch (conv1a of this item finished reading img_s before the barrier of chunk 1;
        // the next reader is the conv1a in chunk 3), then the request for the item after it
        if (tid < S4_IMR * S4_IMC) img_s[tid] = imv;
        imv = load_img(nitem + gridDim.x);
      }
      S4_T(te);
#if S4_DIAG & 1
      unsigned long long cv_cycles = 0;
      dg[0] += te - t_prev;  // item head and chunk seams
#endif
#pragma unroll 1
      for (int half = 0; half < 4; ++half) {  // (not unrolled: one copy of the conv1a code and of the half k group)
        const int g = half >> 1, hh = half & 1;
        // conv1a, staggered over the three waves of a SIMD (one of them transforms pixels while the other two keep
        // the matrix pipe busy): chunk 1 -> this item's chunks 2, 3 (buffers A, C); chunk 3 -> the next item's
        // chunks 0, 1 (buffers A, B)
        if ((c & 1) && (c == 1 || have_next) && !(S4_DIAG & 4)) {
          S4_T(tu);
          conv1a_at_half(half, c == 1 ? 1 : 0, c == 1 ? y0 : ny0, c == 1 ? x0 : nx0, bufA, c == 1 ? bufC : bufB);
#if S4_DIAG & 1
          cv_cycles += __builtin_readcyclecounter() - tu;
#endif
        }
        // (scheduling fences: hipcc otherwise issues all 36 / 48 reads of a half -- or of several halves -- up front,
        // and with 108 of the 168 registers holding accumulators and filter fragments that spills)
        __builtin_amdgcn_sched_barrier(0);
        S4_T(tha);
        // this half's filter fragments: requested here, consumed after the transforms below (~1.5 k cycles later);
        // they are live neither during conv1a above nor during the epilogue (register budget)
        float2 bq[6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) bq[nu] = (S4_DIAG & 16) ? make_float2(1.f + nu, 0.5f) : bfrag(4 * c + half, nu);
        const float* pl = ps + g * S4_PLANE;
        const float* pA = pl + (offA ^ (2 * hh));
        const float* pB = pl + (offB ^ (2 * hh));
        const float* q0 = pA + ro0;
        const float* q1 = pA + ro1;
        const float* q2 = (hiR ? pB : pA) + ro2;
        const float* q3 = (hiS ? pB : pA) + ro3;
        // row transform of column j (two columns' reads in flight), then the six column transforms:
        //   v0 = 4 T0 - 5 T2 + T4   v1 = -4 T1 - 4 T2 + T3 + T4   v2 = 4 T1 - 4 T2 - T3 + T4
        //   v3 = -2 T1 - T2 + 2 T3 + T4   v4 = 2 T1 - T2 - 2 T3 + T4   v5 = 4 T1 - 5 T3 + T5
        v2f tr[6];
        if (three) {
          v2f d[2][3];
#pragma unroll
          for (int j = 0; j < 2; ++j) { d[j][0] = lds_pair(q0 + COLOFF[j]); d[j][1] = lds_pair(q1 + COLOFF[j]); d[j][2] = lds_pair(q2 + COLOFF[j]); }
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            tr[j] = v2_fma(c1, d[j & 1][1], v2_mul(c0, d[j & 1][0])) + d[j & 1][2];
            if (j + 2 < 6) { d[j & 1][0] = lds_pair(q0 + COLOFF[j + 2]); d[j & 1][1] = lds_pair(q1 + COLOFF[j + 2]); d[j & 1][2] = lds_pair(q2 + COLOFF[j + 2]); }
          }
        } else {
          v2f d[2][4];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            d[j][0] = lds_pair(q0 + COLOFF[j]); d[j][1] = lds_pair(q1 + COLOFF[j]);
            d[j][2] = lds_pair(q2 + COLOFF[j]); d[j][3] = lds_pair(q3 + COLOFF[j]);
          }
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            tr[j] = v2_fma(c0, v2_fma(c1, d[j & 1][1], d[j & 1][0]), v2_fma(c2, d[j & 1][3], d[j & 1][2]));
            if (j + 2 < 6) {
              d[j & 1][0] = lds_pair(q0 + COLOFF[j + 2]); d[j & 1][1] = lds_pair(q1 + COLOFF[j + 2]);
              d[j & 1][2] = lds_pair(q2 + COLOFF[j + 2]); d[j & 1][3] = lds_pair(q3 + COLOFF[j + 2]);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        v2f v[6];
        {
          const v2f s12 = tr[1] + tr[2], d12 = tr[1] - tr[2], s34 = tr[3] + tr[4], d34 = tr[3] - tr[4];
          const v2f e31 = tr[3] - tr[1], e42 = tr[4] - tr[2];
          v[0] = v2_fma(-5.f, tr[2], v2_fma(4.f, tr[0], tr[4]));
          v[1] = v2_fma(-4.f, s12, s34);
          v[2] = v2_fma(4.f, d12, -d34);
          v[3] = v2_fma(2.f, e31, e42);
          v[4] = v2_fma(-2.f, e31, e42);
          v[5] = v2_fma(-5.f, tr[3], v2_fma(4.f, tr[1], tr[5]));
        }
#if S4_DIAG & 1
        asm volatile("" : "+v"(v[0]), "+v"(v[5]));
        S4_T(thb);
#endif
#pragma unroll
        for (int nu = 0; nu < 6; nu += 2) {
          acc[nu] = mfma32(v[nu].x, bq[nu].x, acc[nu]);
          acc[nu + 1] = mfma32(v[nu + 1].x, bq[nu + 1].x, acc[nu + 1]);
          acc[nu] = mfma32(v[nu].y, bq[nu].y, acc[nu]);
          acc[nu + 1] = mfma32(v[nu + 1].y, bq[nu + 1].y, acc[nu + 1]);
        }
#if S4_DIAG & 1
        {
          S4_T(thc);
          dg[1] += thb - tha; dg[2] += thc - thb;  // transform phase | MFMA issue phase of the half
        }
#endif
      }
      S4_T(tg);
      __syncthreads();
      S4_T(th);
#if S4_DIAG & 1
      dg[3] += cv_cycles; dg[4] += tg - te - cv_cycles; dg[5] += th - tg;
      t_prev = th;
#endif
    }
#endif
    S4_T(ti);

    // ---- output transform.  Column direction (nu -> j) lane-local:
    //   z0 = m0 + m1 + m2 + m3 + m4   z1 = (m1 - m2) + 2 (m3 - m4)   z2 = (m1 + m2) + 4 (m3 + m4)
    //   z3 = (m1 - m2) + 8 (m3 - m4) + m5
    // then four exchange passes (output-channel tile x column pair) through X = [xi 6][jj 2][tile 32][cout 32]; the
    // row direction (xi -> i) is the same combination across the six waves
    float* ex = bufC;
    int tq = tid;
    asm volatile("" : "+v"(tq));  // (opaque: keeps the reader's index arithmetic out of the item loop's live ranges)
    const int co_q = (tq & 7) * 4, ip2 = (tq >> 3) & 1, tile = tq >> 4;  // reader role (tid < 512)
    const int Ho = a_H >> 1, Wo = a_W >> 1;
#if S4_DIAG & 32  // ablation: no epilogue; the accumulators stay alive
    {
      float sum_ = 0.f;
#pragma unroll
      for (int nu = 0; nu < 6; ++nu)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum_ += acc[nu][r];
      if (sum_ == 12345.678f) a_y[tid] = sum_;
    }
#endif
#pragma unroll 1
    for (int pnt = 0; pnt < ((S4_DIAG & 32) ? 0 : 2); ++pnt)  // output-channel tile
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {   // pooled column inside the Winograd tile = output columns 2 jp, 2 jp + 1
      if (nt == pnt) {
        int ln = lane;
        asm volatile("" : "+v"(ln));  // (opaque, as in conv1a_unit)
        float* ob = ex + xi * 2048 + 4 * (ln >> 5) * 32 + (ln & 31);  // + tile row of register r: a compile-time offset
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          float* o = ob + ((r & 3) + 8 * (r >> 2)) * 32;
          if (jp == 0) {
            o[0] = (acc[0][r] + s12) + s34;
            o[1024] = fmaf(2.f, d34, d12);
          } else {
            o[0] = fmaf(4.f, s34, s12);
            o[1024] = fmaf(8.f, d34, d12) + acc[5][r];
          }
        }
      }
      __syncthreads();
      if (tid < 512) {
        // thread -> (tile, channel quad, pooled row ip2 of the tile): output rows 2 ip2, 2 ip2 + 1 of both columns
        //   ip2 = 0: y0 = z0 + (z1 + z2) + (z3 + z4), y1 = (z1 - z2) + 2 (z3 - z4)
        //   ip2 = 1: y2 = (z1 + z2) + 4 (z3 + z4),    y3 = (z1 - z2) + 8 (z3 - z4) + z5
        const int co = pnt * 32 + co_q;
        const float kd = ip2 ? 8.f : 2.f;
        float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          __builtin_amdgcn_sched_barrier(0);  // a few reads at a time: the other cout tile's accumulators are still live
          const float* zp = ex + jj * 1024 + tile * 32 + co_q;
          const float4 z1 = *reinterpret_cast<const float4*>(zp + 2048);
          const float4 z2 = *reinterpret_cast<const float4*>(zp + 2 * 2048);
          const float4 s12 = f4_addv(z1, z2), d12 = f4_subv(z1, z2);
          __builtin_amdgcn_sched_barrier(0);
          const float4 z3 = *reinterpret_cast<const float4*>(zp + 3 * 2048);
          const float4 z4 = *reinterpret_cast<const float4*>(zp + 4 * 2048);
          const float4 s34 = f4_addv(z3, z4), d34 = f4_subv(z3, z4);
          __builtin_amdgcn_sched_barrier(0);
          const float4 ze = *reinterpret_cast<const float4*>(zp + (ip2 ? 5 * 2048 : 0));  // z5 or z0
          float4 ya = ip2 ? f4_fma(4.f, s34, s12) : f4_addv(f4_addv(ze, s12), s34);
          float4 yb = f4_fma(kd, d34, d12);
          if (ip2) yb = f4_addv(yb, ze);
          const float4 bi = *reinterpret_cast<const float4*>(c2_s + co);
          const float4 sc = *reinterpret_cast<const float4*>(c2_s + 64 + co);
          const float4 sh = *reinterpret_cast<const float4*>(c2_s + 128 + co);
          float4 t = f4_addv(ya, bi);
          t = make_float4(fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f));
          ya = make_float4(t.x * sc.x + sh.x, t.y * sc.y + sh.y, t.z * sc.z + sh.z, t.w * sc.w + sh.w);
          t = f4_addv(yb, bi);
          t = make_float4(fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f));
          yb = make_float4(t.x * sc.x + sh.x, t.y * sc.y + sh.y, t.z * sc.z + sh.z, t.w * sc.w + sh.w);
          const float4 m = f4_max(ya, yb);
          pool = jj == 0 ? m : f4_max(pool, m);
        }
        const int oy = (y0 >> 1) + 2 * (tile >> 3) + ip2, ox = (x0 >> 1) + 2 * (tile & 7) + jp;
        if (ox < Wo && oy < Ho) *reinterpret_cast<float4*>(a_y + (((size_t)b * Ho + oy) * Wo + ox) * 64 + co) = pool;
      }
      __syncthreads();  // X is free again (next pass / chunk 3 of the next item)
    }
#if S4_DIAG & 1
    {
      S4_T(tj);
      dg[6] += tj - ti; dg[7] += 1;
    }
#endif
    y0 = ny0; x0 = nx0; b = nb_;
  }
#if S4_DIAG & 1
  if (args.diag && lane == 0) {
    unsigned long long* o = args.diag + ((size_t)blockIdx.x * 12 + wave) * 16;
    for (int i = 0; i < 8; ++i) o[i] = dg[i];
    o[8] = __builtin_readcyclecounter() - dg_t0;
    o[9] = dg[8];
  }
#endif
}

// conv1a (1 -> 64) + conv1b (64 -> 64, Winograd F(4x4,3x3)) + 2x2 max-pool in one launch: image [B,H,W] -> [B,H/2,W/2,64]
extern "C" int gfc_sp_stem_wino43(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                                  const float* w2_wino43, const float* b2, const float* s2, const float* t2, float* y,
                                  int B, int H, int W, void* stream) {
  if (!image || !w1 || !b1 || !w2_wino43 || !b2 || !y || B <= 0 || H < 2 || W < 2) return GFC_ERR_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return GFC_ERR_INVALID;
  Stem43Args a = {};
  a.x = image; a.w = w2_wino43; a.bias = b2; a.scale = s2; a.shift = t2; a.y = y;
  a.B = B; a.H = H; a.W = W;
  a.w1 = w1; a.b1 = b1; a.s1 = s1; a.t1 = t1;
  a.tiles_x = (W + S4_TW - 1) / S4_TW;
  a.tiles_y = (H + S4_TH - 1) / S4_TH;
  const long long nitems = (long long)a.tiles_x * a.tiles_y * B;
  if (nitems >= (1ll << 31)) return GFC_ERR_UNSUPPORTED;
  constexpr size_t lds = (size_t)(2 * S4_PATCH + S4_EX + S4_IMR * S4_IMC + 768 + 192 + (S4_COOP ? S4_VFLOATS : 0)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  static_assert(S4_EX >= S4_PATCH, "patch buffer C is the head of the exchange buffer");
  static std::atomic<unsigned long long> lds_ok{0};
  gfc_allow_dynamic_lds((const void*)stem_wino43_kernel, lds, lds_ok);
  long long grid = gfc_device_cus();  // one persistent workgroup per CU
  if (grid > nitems) grid = nitems;
#if S4_DIAG & 1
  a.diag = g_s4_diag;
#endif
  hipLaunchKernelGGL(stem_wino43_kernel, dim3((unsigned)grid), dim3(S4_THREADS), lds, (hipStream_t)stream, a);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
