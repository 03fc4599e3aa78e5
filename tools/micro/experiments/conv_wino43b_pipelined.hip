// SuperPoint stem (conv1a 1->64 + conv1b 64->64 as Winograd F(4x4,3x3) + 2x2 max-pool), second mapping: ALL 36
// transform positions of a (16 Winograd tiles x 16 output channels) block live in ONE wave.
//
// Same function and arithmetic as csrc/conv_wino43.hip (reference gluefactory/models/extractors/superpoint_open.py:
// 61-77,100-108; gluefactory_nonfree/superpoint.py:214-218); what changes is who holds what:
//   * v_mfma_f32_16x16x4_f32 (M = 16 Winograd tiles, N = 16 output channels, K = 4 input channels; 4 accumulator
//     registers): 36 positions x 4 = 144 accumulator registers per wave, 8 waves per CU (2 per SIMD, 256 registers):
//     wave (th, cq) = tile half th (16 of the item's 32 tiles) x output-channel quarter cq.
//   * the OUTPUT transform A^T M A is lane-local (a lane holds all 36 positions of its 4 (tile, channel) pairs): no
//     exchange through LDS, no barrier in the epilogue -- the first mapping needs four exchange passes because xi is
//     spread over waves.
//   * the INPUT transform B^T d B is computed ONCE per (tile, input channel) -- one lane, one 6x6 patch, 36 values
//     written to LDS as the A operands of every output-channel quarter (the first mapping computes it in both nt waves)
//     -- and it is lane-local as well: no row / column split over waves, no latency chain of partial reads.
//   * operands of the MFMAs come from LDS as ds_read_b128 (4 positions of one (tile | channel, k) per lane): the
//     transformed filters of one 4-channel group (36 KB) are brought by global_load_lds DMA, double-buffered.
// Phases per 8-channel half chunk (all 8 waves, 3 barriers): P2 transform (waves 0-3, one patch per lane) | P3a 36 MFMAs on
// channel group 0 | P3b 36 MFMAs on channel group 1; conv1a of the NEXT half chunk (one output channel per wave, VALU, the
// same fmaf chain as every other stem) is interleaved with the MFMAs, which do not touch the patch buffer.
#include "common.h"

#define B4_THREADS 512
#ifndef B4_DIAG
#define B4_DIAG 0  // ablations with wrong results (tools/ab_build.sh WORKTREE b4 "-DB4_DIAG=..."): 1 no transform, 2 no conv1a between the MFMAs, 4 no MFMAs, 8 no epilogue, 16 no filter DMA
#endif
#define B4_PC 34                  // patch columns (18 rows)
#define B4_NPIX 612
#define B4_PLANE 641              // floats per channel plane of the patch: 612 + pad, = 1 (mod 32)
#define B4_IMC 36                 // image patch 20 x 36
#define B4_VK 160                 // V: stride between the 4 input channels of a group (128 + 32: b128 writes of two k conflict-free)
#define B4_VPG (4 * B4_VK)        // floats per (channel group, position group)
#define B4_VCG (9 * B4_VPG)
#define B4_BCG (9 * 4 * 64 * 4)   // filter floats per 4-channel group: [pg 9][k 4][cout 64][4 positions]
#define B4_OFF_C1 720
#define B4_OFF_C2 (B4_OFF_C1 + 768)
#define B4_OFF_PATCH (B4_OFF_C2 + 192)
#define B4_OFF_V (B4_OFF_PATCH + 8 * B4_PLANE)
#define B4_OFF_B (B4_OFF_V + 2 * B4_VCG)
#define B4_LDS_FLOATS (B4_OFF_B + 2 * B4_BCG)

#if B4_DIAG & 32
static unsigned long long* g_b4_diag = nullptr;
extern "C" void gfc_diag_set_stem43b_stamps(void* p) { g_b4_diag = (unsigned long long*)p; }
// (stamps live in VGPRs: with them in SGPRs the register allocator runs out of scalars and the DMA's inline assembly
// fails verification)
#define B4_T(v_) const unsigned long long v_ = __builtin_readcyclecounter() + b4_vz
#else
#define B4_T(v_) do {} while (0)
#endif
typedef __attribute__((address_space(3))) void* b4_lds_ptr_t;
typedef float v2f __attribute__((ext_vector_type(2)));

struct StemBArgs {
  const float* x;      // image [B,H,W]
  const float* w;      // conv1b filters packed by gfc_pack_conv3x3_wino43b
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
  unsigned long long* diag;  // B4_DIAG & 32: 8 words per wave
};

// U = G g G^T (6x6) in float64, rounded once -> out[cg = cin/4][pg = pos/4][k = cin%4][cout][j = pos%4], pos = 6 xi + nu
__global__ void pack_conv3x3_wino43b_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin) {
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
  const int cg = ci >> 2, k = ci & 3;
  for (int xi = 0; xi < 6; ++xi)
    for (int nu = 0; nu < 6; ++nu) {
      const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
      const int pos = xi * 6 + nu;
      out[((((size_t)cg * 9 + (pos >> 2)) * 4 + k) * cout + co) * 4 + (pos & 3)] = (float)u;
    }
}

extern "C" int gfc_pack_conv3x3_wino43b(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
  if (!w_oihw || !w_packed || cout != 64 || cin != 64) return GFC_ERR_INVALID;  // the stem's conv1b only
  const int total = cout * cin;
  hipLaunchKernelGGL(pack_conv3x3_wino43b_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_packed, cout, cin);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// one line of B^T d (F(4x4,3x3) input transform)
__device__ __forceinline__ void b4_bt6(const float d0, const float d1, const float d2, const float d3, const float d4,
                                       const float d5, float (&o)[6]) {
  o[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
  o[1] = a + b;
  o[2] = a - b;
  const float c = d4 - d2, e = 2.f * (d3 - d1);
  o[3] = c + e;
  o[4] = c - e;
  o[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}
// one line of A^T m (output transform): 6 -> 4
__device__ __forceinline__ void b4_at4(const float m0, const float m1, const float m2, const float m3, const float m4,
                                       const float m5, float (&o)[4]) {
  const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  o[0] = (m0 + s12) + s34;
  o[1] = fmaf(2.f, d34, d12);
  o[2] = fmaf(4.f, s34, s12);
  o[3] = fmaf(8.f, d34, d12) + m5;
}

__global__ __launch_bounds__(B4_THREADS, 1) void stem_wino43b_kernel(StemBArgs args) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img_s = smem;
  float* c1_s = smem + B4_OFF_C1;
  float* c2_s = smem + B4_OFF_C2;
  float* patch_s = smem + B4_OFF_PATCH;  // [2][4 channels][B4_PLANE]
  float* V_s = smem + B4_OFF_V;          // [2][B4_VCG]
  float* B_s = smem + B4_OFF_B;          // [2][B4_BCG]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a_H = args.H, a_W = args.W;
  // MFMA role
  const int th = wave >> 2, cq = wave & 3;
  const int fm = lane & 15, fk = lane >> 4;
  const int a_frag = fk * B4_VK + (th * 16 + fm) * 4;  // + buffer * B4_VCG + pg * B4_VPG
  const int b_frag = (fk * 64 + cq * 16 + fm) * 4;     // + buffer * B4_BCG + pg * 1024

  for (int i = tid; i < 768; i += B4_THREADS)
    c1_s[i] = i < 576 ? args.w1[i] : i < 640 ? args.b1[i - 576] : i < 704 ? (args.s1 ? args.s1[i - 640] : 1.f)
                                                                           : (args.t1 ? args.t1[i - 704] : 0.f);
  for (int i = tid; i < 192; i += B4_THREADS)
    c2_s[i] = i < 64 ? args.bias[i] : i < 128 ? (args.scale ? args.scale[i - 64] : 1.f) : (args.shift ? args.shift[i - 128] : 0.f);

  // filters of channel group cg -> B_s[cg & 1]: 36 pieces of 1 KB, wave w brings pieces w, w + 8, ...
  auto dma_filters = [&](int cg) __attribute__((always_inline)) {
#if !(B4_DIAG & 16)
    const float* src = args.w + (size_t)cg * B4_BCG + lane * 4;
    float* dst = B_s + (cg & 1) * B4_BCG;
    for (int piece = wave; piece < 36; piece += 8) {
      unsigned keep;
      const unsigned la = (unsigned)(size_t)(b4_lds_ptr_t)(dst + piece * 256);
      const float* gp = src + piece * 256;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gp), "s"(__builtin_amdgcn_readfirstlane(la))
                   : "memory");
    }
#endif
  };

#if B4_DIAG & 32
  unsigned b4_vz32;
  asm volatile("v_mov_b32 %0, 0" : "=v"(b4_vz32));
  const unsigned long long b4_vz = b4_vz32;
  unsigned long long dg[8] = {b4_vz, b4_vz, b4_vz, b4_vz, b4_vz, b4_vz, b4_vz, b4_vz};
  const unsigned long long dg_t0 = __builtin_readcyclecounter() + b4_vz;
#endif
  const int ntiles = args.tiles_x * args.tiles_y;
  const int nitems = ntiles * args.B;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / ntiles;
    const int t_in = item - b * ntiles;
    const int y0 = (t_in / args.tiles_x) * 16, x0 = (t_in % args.tiles_x) * 32;
    const float* img = args.x + (size_t)b * a_H * a_W;

    // conv1a of one wave batch: channel c (wave-uniform), 64 pixel PAIRS (p, p + 306: same patch column, nine rows apart)
    // evaluated with packed fp32 FMAs (each half is the fmaf chain of every other stem) -> patch_s[(c >> 2) & 1][c & 3]
    auto conv1a_batch = [&](int c, int it) __attribute__((always_inline)) {
      float cw[12];
#pragma unroll
      for (int t = 0; t < 9; ++t) cw[t] = c1_s[t * 64 + c];
      cw[9] = c1_s[576 + c];
      cw[10] = c1_s[640 + c];
      cw[11] = c1_s[704 + c];
      const int p = it * 64 + lane;
      const int pc = min(p, 305);
      const int row = pc / B4_PC, col = pc - row * B4_PC;
      const float* ip = img_s + row * B4_IMC + col;
      v2f a = v2f{0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const v2f iv = v2f{ip[(t / 3) * B4_IMC + (t % 3)], ip[(t / 3 + 9) * B4_IMC + (t % 3)]};
        a = v2f{fmaf(iv.x, cw[t], a.x), fmaf(iv.y, cw[t], a.y)};
      }
      a = v2f{a.x + cw[9], a.y + cw[9]};
      const v2f v = v2f{fmaxf(a.x, 0.f) * cw[10] + cw[11], fmaxf(a.y, 0.f) * cw[10] + cw[11]};
      if (p < 306) {
        const int gy = y0 - 1 + row, gx = x0 - 1 + col;
        const bool cin_ = (unsigned)gx < (unsigned)a_W;
        float* dst = patch_s + (((c >> 2) & 1) * 4 + (c & 3)) * B4_PLANE + p;
        dst[0] = (cin_ && (unsigned)gy < (unsigned)a_H) ? v.x : 0.f;
        dst[306] = (cin_ && (unsigned)(gy + 9) < (unsigned)a_H) ? v.y : 0.f;
      }
    };
    // input transform of channel group cg by two waves (w2 = 0 / 1: tile rows 2 w2, 2 w2 + 1), one (tile, channel) per lane
    auto transform = [&](int cg, int w2) __attribute__((always_inline)) {
#if !(B4_DIAG & 1)
      const int tx = lane & 7, k = (lane >> 3) & 3, ty = 2 * w2 + (lane >> 5);
      const float* pp = patch_s + ((cg & 1) * 4 + k) * B4_PLANE + (4 * ty) * B4_PC + 4 * tx;
      float tmp[6][6];  // B^T d
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float o[6];
        b4_bt6(pp[j], pp[B4_PC + j], pp[2 * B4_PC + j], pp[3 * B4_PC + j], pp[4 * B4_PC + j], pp[5 * B4_PC + j], o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
      }
      float* vp = V_s + (cg & 1) * B4_VCG + k * B4_VK + (ty * 8 + tx) * 4;
      // two rows of B^T d B at a time = positions 12 i2 .. 12 i2 + 11 = three complete position groups
#pragma unroll
      for (int i2 = 0; i2 < 3; ++i2) {
        float o0[6], o1[6];
        b4_bt6(tmp[2 * i2][0], tmp[2 * i2][1], tmp[2 * i2][2], tmp[2 * i2][3], tmp[2 * i2][4], tmp[2 * i2][5], o0);
        b4_bt6(tmp[2 * i2 + 1][0], tmp[2 * i2 + 1][1], tmp[2 * i2 + 1][2], tmp[2 * i2 + 1][3], tmp[2 * i2 + 1][4],
               tmp[2 * i2 + 1][5], o1);
        *reinterpret_cast<float4*>(vp + (3 * i2) * B4_VPG) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<float4*>(vp + (3 * i2 + 1) * B4_VPG) = make_float4(o0[4], o0[5], o1[0], o1[1]);
        *reinterpret_cast<float4*>(vp + (3 * i2 + 2) * B4_VPG) = make_float4(o1[2], o1[3], o1[4], o1[5]);
      }
#endif
    };

    B4_T(t_item);
    __syncthreads();  // the previous item is through with every buffer (and the constants are in place)
    for (int i = tid; i < 20 * B4_IMC; i += B4_THREADS) {
      const int gy = y0 - 2 + i / B4_IMC, gx = x0 - 2 + i % B4_IMC;
      img_s[i] = (gy >= 0 && gy < a_H && gx >= 0 && gx < a_W) ? img[(size_t)gy * a_W + gx] : 0.f;
    }
    dma_filters(0);
    dma_filters(1);
    __syncthreads();
    // pipeline fill: conv1a of channel groups 0 and 1 (8 channels x 5 batches over 8 waves), transform of group 0
#pragma unroll 1
    for (int it = 0; it < 5; ++it) conv1a_batch(wave, it);
    __syncthreads();
    if (wave < 2) transform(0, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    B4_T(t_loop);
#if B4_DIAG & 32
    dg[0] += t_loop - t_item;
#endif

    f32x4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int s16 = 0; s16 < 16; ++s16) {
      B4_T(t_s);
      // ---- the two halves of a step, in opposite order on the two waves of a SIMD (waves w and w + 4):
      //   mfma:  36 MFMAs of channel group s16 (V_s / B_s buffer s16 & 1)
      //   valu:  input transform of group s16 + 1 (waves 2j, 2j + 1, j = s16 & 3) or this wave's share of conv1a of
      //          group s16 + 2 (the other six waves; 4 channels x 5 batches = 20 batches) ----
      auto mfma_half = [&]() __attribute__((always_inline)) {
        const float* ap = V_s + (s16 & 1) * B4_VCG + a_frag;
        const float* bp = B_s + (s16 & 1) * B4_BCG + b_frag;
        float4 a_cur = *reinterpret_cast<const float4*>(ap);
        float4 w_cur = *reinterpret_cast<const float4*>(bp);
#pragma unroll
        for (int pg = 0; pg < 9; ++pg) {
          float4 a_nxt = a_cur, w_nxt = w_cur;
          if (pg < 8) {
            a_nxt = *reinterpret_cast<const float4*>(ap + (pg + 1) * B4_VPG);
            w_nxt = *reinterpret_cast<const float4*>(bp + (pg + 1) * 1024);
          }
#if B4_DIAG & 4
          acc[4 * pg][0] += a_cur.x * w_cur.x; acc[4 * pg + 1][0] += a_cur.y * w_cur.y; acc[4 * pg + 2][0] += a_cur.z * w_cur.z; acc[4 * pg + 3][0] += a_cur.w * w_cur.w;
#else
          acc[4 * pg] = mfma16(a_cur.x, w_cur.x, acc[4 * pg]);
          acc[4 * pg + 1] = mfma16(a_cur.y, w_cur.y, acc[4 * pg + 1]);
          acc[4 * pg + 2] = mfma16(a_cur.z, w_cur.z, acc[4 * pg + 2]);
          acc[4 * pg + 3] = mfma16(a_cur.w, w_cur.w, acc[4 * pg + 3]);
#endif
          a_cur = a_nxt;
          w_cur = w_nxt;
        }
      };
      auto valu_half = [&]() __attribute__((always_inline)) {
        const int j2 = 2 * (s16 & 3);
        if (wave == j2 || wave == j2 + 1) {
          if (s16 < 15) transform(s16 + 1, wave - j2);
        } else if (s16 < 14 && !(B4_DIAG & 2)) {
          const int r = wave < j2 ? wave : wave - 2;  // rank among the six conv1a waves
#pragma unroll 1
          for (int ii = r; ii < 20; ii += 6) conv1a_batch(4 * (s16 + 2) + ii / 5, ii % 5);
        }
      };
      // (ONE copy of the MFMA half: with it in both arms of a branch the 144 accumulator registers go to scratch)
      if (wave < 4) valu_half();
      mfma_half();
      if (wave >= 4) valu_half();
      B4_T(t_w);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's filter pieces of group s16 + 1 (requested a step ago)
      __syncthreads();  // V_s / patch_s / B_s of the next step are complete, this step's readers are done
      if (s16 < 14) dma_filters(s16 + 2);
#if B4_DIAG & 32
      const unsigned long long t_e = __builtin_readcyclecounter() + b4_vz;
      dg[1] += t_w - t_s;
      dg[2] += t_e - t_w;
#endif
    }
    B4_T(t_epi);

    // ---- epilogue: lane-local output transform, bias, ReLU, BN affine, 2x2 max-pool, NHWC store ----
    const int n = cq * 16 + fm;
    const float bias = c2_s[n], sc = c2_s[64 + n], sh = c2_s[128 + n];
    const int Ho = a_H >> 1, Wo = a_W >> 1;
    float* yb = args.y + (size_t)b * Ho * Wo * 64 + n;
#if B4_DIAG & 8
    {
      f32x4 sum = acc[0];
#pragma unroll
      for (int p = 1; p < 36; ++p) sum += acc[p];
      if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) yb[0] = 1.f;
    }
#else
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = th * 16 + fk * 4 + r;
      const int ty = t >> 3, tx = t & 7;
      float z[6][4];  // M A (columns transformed)
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        float o[4];
        b4_at4(acc[xi * 6][r], acc[xi * 6 + 1][r], acc[xi * 6 + 2][r], acc[xi * 6 + 3][r], acc[xi * 6 + 4][r], acc[xi * 6 + 5][r], o);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[xi][j] = o[j];
      }
      float yv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float o[4];
        b4_at4(z[0][j], z[1][j], z[2][j], z[3][j], z[4][j], z[5][j], o);
#pragma unroll
        for (int i = 0; i < 4; ++i) yv[i][j] = fmaxf(o[i] + bias, 0.f) * sc + sh;
      }
#pragma unroll
      for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          const float m = fmaxf(fmaxf(yv[2 * pi][2 * pj], yv[2 * pi][2 * pj + 1]), fmaxf(yv[2 * pi + 1][2 * pj], yv[2 * pi + 1][2 * pj + 1]));
          const int yo = (y0 >> 1) + 2 * ty + pi, xo = (x0 >> 1) + 2 * tx + pj;
          if (yo < Ho && xo < Wo) yb[((size_t)yo * Wo + xo) * 64] = m;
        }
    }
#endif
#if B4_DIAG & 32
    dg[5] += __builtin_readcyclecounter() + b4_vz - t_epi;
    dg[6] += 1;
#endif
  }
#if B4_DIAG & 32
  if (args.diag && lane == 0) {
    unsigned long long* o = args.diag + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 7; ++i) o[i] = dg[i];
    o[7] = __builtin_readcyclecounter() + b4_vz - dg_t0;
  }
#endif
}

// conv1a (1 -> 64) + conv1b (64 -> 64, Winograd F(4x4,3x3)) + 2x2 max-pool in one launch: image [B,H,W] -> [B,H/2,W/2,64]
extern "C" int gfc_sp_stem_wino43b(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                                   const float* w2_wino43b, const float* b2, const float* s2, const float* t2, float* y,
                                   int B, int H, int W, void* stream) {
  if (!image || !w1 || !b1 || !w2_wino43b || !b2 || !y || B <= 0 || H < 2 || W < 2) return GFC_ERR_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return GFC_ERR_INVALID;
  StemBArgs a = {};
  a.x = image; a.w = w2_wino43b; a.bias = b2; a.scale = s2; a.shift = t2; a.y = y;
  a.B = B; a.H = H; a.W = W;
  a.w1 = w1; a.b1 = b1; a.s1 = s1; a.t1 = t1;
  a.tiles_x = (W + 31) / 32;
  a.tiles_y = (H + 15) / 16;
  const long long nitems = (long long)a.tiles_x * a.tiles_y * B;
  if (nitems >= (1ll << 31)) return GFC_ERR_UNSUPPORTED;
  constexpr size_t lds = (size_t)B4_LDS_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};
  gfc_allow_dynamic_lds((const void*)stem_wino43b_kernel, lds, lds_ok);
  long long grid = gfc_device_cus();  // one persistent workgroup per CU
  if (grid > nitems) grid = nitems;
#if B4_DIAG & 32
  a.diag = g_b4_diag;
#endif
  hipLaunchKernelGGL(stem_wino43b_kernel, dim3((unsigned)grid), dim3(B4_THREADS), lds, (hipStream_t)stream, a);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
