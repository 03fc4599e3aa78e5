// The DISK network (config 5): a thin U-Net of 5x5 convolutions, as fp32-MFMA implicit GEMM on NHWC activations.
//
// Replaces the call `kornia.feature.DISK(...)` makes into its network (reference
// gluefactory/models/extractors/disk_kornia.py:24-47: `self.model.heatmap_and_dense_descriptors`).  kornia's source is
// absent offline; the architecture below is restated from kornia's published code (kornia/feature/disk/disk.py,
// kornia/feature/disk/_unets/{unet,blocks}.py) -- NETWORK PARITY UNPINNED, the checker is oracle/disk_unet.py:
//
//   Unet(in_features=3, size=5, down=[16,32,64,64,64], up=[64,64,64,desc_dim+1]), "thin" blocks:
//     down block i:  [avg_pool2d(2)  unless first] -> Conv
//     up block:      bilinear x2 (align_corners=False) of the bottom path, cat [bottom_big | horizontal] -> Conv
//     Conv:          [InstanceNorm2d(eps 1e-5, no affine) -> PReLU(per channel)  unless first] -> Conv2d 5x5, pad 2, bias
//
// Kernels
//   disk_conv5x5_kernel<NT>  one workgroup (4 waves) = 16x16 pixels x NT*32 output channels.  The 20x20 halo patch of a
//                            16-channel chunk sits in LDS, normalised + gated WHILE it is staged (zero padding applied
//                            after the gate, as Conv2d pads its own input); the A operand of tap (dy,dx) is the same
//                            patch read at a shifted pixel offset (no im2col); K walks (chunk) x (25 taps) with the
//                            [NT*32 x 16] weight slice double-buffered in LDS.  Pitch 20 floats: 16 consecutive pixels
//                            (or output channels) cover the 16 bank quads once -> conflict-free ds_read_b128.
//                            Output with a channel stride, so that a layer writes straight into its slice of the
//                            concatenated tensor the matching up block reads (no torch.cat copy).
//   disk_instnorm_partial / finalize   per (image, channel) mean and 1/sqrt(var + eps) over H x W, float64 sums,
//                            two stages in a fixed order (deterministic).
//   disk_avgpool2_kernel, disk_upsample2_kernel, disk_nchw_to_nhwc4_kernel   the glue between the levels.
#include "common.h"

#define DT 16              // output tile edge
#define DH (DT + 4)        // halo edge
#define DKC 16             // channels per chunk
#define DLD 20             // LDS pitch in floats
#define DSTAT_T 480        // threads of the statistics kernel: divisible by C/4 for C = 16, 32, 64, 80, 96, 128

// OIHW [cout][cin][5][5] -> [chunk][tap][co_pad][16] (cin padded to a multiple of 16, cout to a multiple of 32, zeros)
__global__ void disk_pack_conv5x5_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int co_pad,
                                         int chunks) {
  const long long total = (long long)chunks * 25 * co_pad * DKC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % DKC);
    long long t = i / DKC;
    const int co = (int)(t % co_pad);
    t /= co_pad;
    const int tap = (int)(t % 25), chunk = (int)(t / 25);
    const int ci = chunk * DKC + c;
    out[i] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * 25 + tap] : 0.f;
  }
}

struct Disk5Args {
  const float* x;      // [B,H,W,cin] contiguous
  const float* mean;   // [B,cin] or null (no normalisation)
  const float* rstd;   // [B,cin]
  const float* prelu;  // [cin] or null (no gate)
  const float* w;      // packed, already offset to the first output channel of this launch
  const float* bias;   // already offset
  float* y;            // already offset to the first output channel of this launch
  int ldy;             // output channel stride per pixel
  int B, H, W, cin, chunks, co_pad, cout;  // cout = valid output channels of this launch
  int tiles_x, tiles_y;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void disk_conv5x5_kernel(Disk5Args a) {
  constexpr int NB = NT * 32;
  __shared__ __attribute__((aligned(16))) float in_s[DH * DH * DLD];
  __shared__ __attribute__((aligned(16))) float w_s[2][NB * DLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int t = blockIdx.x;
  const int x0 = (t % a.tiles_x) * DT;
  t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * DT;
  const int b = t / a.tiles_y;
  const int co0 = blockIdx.y * NB;
  const float* xin = a.x + (size_t)b * a.H * a.W * a.cin;
  const int c4 = (tid & 3) * 4;  // this thread's channel quad inside a chunk (patch and weight staging alike)

  auto stage_patch = [&](int chunk) __attribute__((always_inline)) {
    const int c = chunk * DKC + c4;
    const bool cvalid = c < a.cin;  // cin is a multiple of 4; channels >= cin of the last chunk are zeros
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f), sl = make_float4(1.f, 1.f, 1.f, 1.f);
    if (cvalid && a.mean) {
      mu = *reinterpret_cast<const float4*>(a.mean + (size_t)b * a.cin + c);
      rs = *reinterpret_cast<const float4*>(a.rstd + (size_t)b * a.cin + c);
    }
    if (cvalid && a.prelu) sl = *reinterpret_cast<const float4*>(a.prelu + c);
#pragma unroll
    for (int i = 0; i < (DH * DH * 4 + 255) / 256; ++i) {
      const int idx = tid + 256 * i;
      const int p = idx >> 2;
      if (p >= DH * DH) break;
      const int gy = y0 - 2 + p / DH, gx = x0 - 2 + p % DH;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cvalid && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        v = *reinterpret_cast<const float4*>(xin + ((size_t)gy * a.W + gx) * a.cin + c);
        v.x = (v.x - mu.x) * rs.x; v.y = (v.y - mu.y) * rs.y; v.z = (v.z - mu.z) * rs.z; v.w = (v.w - mu.w) * rs.w;
        v.x = v.x >= 0.f ? v.x : sl.x * v.x; v.y = v.y >= 0.f ? v.y : sl.y * v.y;
        v.z = v.z >= 0.f ? v.z : sl.z * v.z; v.w = v.w >= 0.f ? v.w : sl.w * v.w;
      }
      *reinterpret_cast<float4*>(in_s + p * DLD + c4) = v;
    }
  };
  // weight slice of one step: NB rows x 16 floats, contiguous in the packed array
  const int wrow = tid >> 2;
  const bool wactive = wrow < NB;
  auto load_w = [&](int step) __attribute__((always_inline)) -> float4 {
    if (!wactive) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(a.w + ((size_t)step * a.co_pad + co0 + wrow) * DKC + c4);
  };
  auto store_w = [&](int buf, float4 v) __attribute__((always_inline)) {
    if (wactive) *reinterpret_cast<float4*>(&w_s[buf][wrow * DLD + c4]) = v;
  };

  // MFMA tile (wave, mt) = image rows 2*wave + mt and that + 8 of the 16x16 tile, 16 pixels each
  int a_off[2], b_off[NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int py = 2 * wave + mt + 8 * (l31 >> 4), px = l31 & 15;
    a_off[mt] = (py * DH + px) * DLD + 4 * h;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_off[nt] = (nt * 32 + l31) * DLD + 4 * h;

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nsteps = a.chunks * 25;
  stage_patch(0);
  store_w(0, load_w(0));
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / 25, tap = step - chunk * 25;
    const bool has_next = step + 1 < nsteps;
    float4 wnext = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_next) wnext = load_w(step + 1);
    const int dy = tap / 5, dx = tap - dy * 5;
    const float* ap = in_s + (dy * DH + dx) * DLD;
    const float* bp = w_s[step & 1];
#pragma unroll
    for (int g = 0; g < DKC / 8; ++g) {
      float4 af[2], bf[NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const float4*>(ap + a_off[mt] + 8 * g);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const float4*>(bp + b_off[nt] + 8 * g);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = mfma32(af[mt].x, bf[nt].x, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].y, bf[nt].y, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].z, bf[nt].z, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].w, bf[nt].w, acc[mt][nt]);
        }
    }
    if (has_next && tap == 24) {  // chunk boundary: every wave is through with the patch before it is replaced
      __syncthreads();
      stage_patch(chunk + 1);
    }
    if (has_next) store_w((step + 1) & 1, wnext);
    __syncthreads();
  }

  // ---- epilogue: + bias, store (lanes over output channels: 128 B per pixel and tile) ----
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = co0 + nt * 32 + l31;
    if (co >= a.cout) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, h);  // pixel of the MFMA tile: row (m >> 4), column (m & 15)
        const int gy = y0 + 2 * wave + mt + 8 * (m >> 4), gx = x0 + (m & 15);
        if (gy < a.H && gx < a.W) a.y[((size_t)((size_t)b * a.H + gy) * a.W + gx) * a.ldy + co] = acc[mt][nt][r] + bv;
      }
  }
}

// One output channel (DISK's heat-map: channel 128 of the last layer).  A matrix-VECTOR product per pixel: on the MFMA
// kernel it costs a 32-wide block for one column (0.33 ms of the 2.2 ms network per VGA image); here one thread owns
// one pixel of the 16x16 tile, the patch is staged exactly as above (normalised + gated, pitch 20: conflict-free
// ds_read_b128 across 16 consecutive pixels) and the 25 x 16 weights of a chunk are wave-uniform scalar loads.
__global__ __launch_bounds__(256) void disk_conv5x5_c1_kernel(Disk5Args a) {
  __shared__ __attribute__((aligned(16))) float in_s[DH * DH * DLD];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int x0 = (t % a.tiles_x) * DT;
  t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * DT;
  const int b = t / a.tiles_y;
  const float* xin = a.x + (size_t)b * a.H * a.W * a.cin;
  const int c4 = (tid & 3) * 4;
  const int py = tid >> 4, px = tid & 15;
  float acc = 0.f;
  for (int chunk = 0; chunk < a.chunks; ++chunk) {
    const int c = chunk * DKC + c4;
    const bool cvalid = c < a.cin;
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f), sl = make_float4(1.f, 1.f, 1.f, 1.f);
    if (cvalid && a.mean) {
      mu = *reinterpret_cast<const float4*>(a.mean + (size_t)b * a.cin + c);
      rs = *reinterpret_cast<const float4*>(a.rstd + (size_t)b * a.cin + c);
    }
    if (cvalid && a.prelu) sl = *reinterpret_cast<const float4*>(a.prelu + c);
    if (chunk) __syncthreads();  // every thread is through with the previous chunk's patch
#pragma unroll
    for (int i = 0; i < (DH * DH * 4 + 255) / 256; ++i) {
      const int idx = tid + 256 * i;
      const int p = idx >> 2;
      if (p >= DH * DH) break;
      const int gy = y0 - 2 + p / DH, gx = x0 - 2 + p % DH;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cvalid && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        v = *reinterpret_cast<const float4*>(xin + ((size_t)gy * a.W + gx) * a.cin + c);
        v.x = (v.x - mu.x) * rs.x; v.y = (v.y - mu.y) * rs.y; v.z = (v.z - mu.z) * rs.z; v.w = (v.w - mu.w) * rs.w;
        v.x = v.x >= 0.f ? v.x : sl.x * v.x; v.y = v.y >= 0.f ? v.y : sl.y * v.y;
        v.z = v.z >= 0.f ? v.z : sl.z * v.z; v.w = v.w >= 0.f ? v.w : sl.w * v.w;
      }
      *reinterpret_cast<float4*>(in_s + p * DLD + c4) = v;
    }
    __syncthreads();
    const float* wc = a.w + (size_t)chunk * 25 * a.co_pad * DKC;  // this channel's row of every tap slice (uniform)
#pragma unroll 5
    for (int tap = 0; tap < 25; ++tap) {
      const int dy = tap / 5, dx = tap - dy * 5;
      const float* ap = in_s + ((py + dy) * DH + px + dx) * DLD;
      const float* wt = wc + (size_t)tap * a.co_pad * DKC;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(ap + 4 * q);
        acc = fmaf(v.x, wt[4 * q], acc);
        acc = fmaf(v.y, wt[4 * q + 1], acc);
        acc = fmaf(v.z, wt[4 * q + 2], acc);
        acc = fmaf(v.w, wt[4 * q + 3], acc);
      }
    }
  }
  const int gy = y0 + py, gx = x0 + px;
  if (gy < a.H && gx < a.W) a.y[((size_t)((size_t)b * a.H + gy) * a.W + gx) * a.ldy] = acc + (a.bias ? a.bias[0] : 0.f);
}

extern "C" size_t gfc_disk_conv5x5_packed_floats(int cout, int cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t chunks = (size_t)(cin + DKC - 1) / DKC, co_pad = (size_t)(cout + 31) / 32 * 32;
  return chunks * 25 * co_pad * DKC;
}

extern "C" int gfc_disk_pack_conv5x5(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
  if (!w_oihw || !w_packed || cout <= 0 || cin <= 0) return GFC_ERR_INVALID;
  const int chunks = (cin + DKC - 1) / DKC, co_pad = (cout + 31) / 32 * 32;
  const long long total = (long long)chunks * 25 * co_pad * DKC;
  hipLaunchKernelGGL(disk_pack_conv5x5_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, w_oihw, w_packed, cout, cin, co_pad, chunks);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_disk_conv5x5(const float* x, const float* mean, const float* rstd, const float* prelu,
                                const float* w_packed, const float* bias, float* y, int ldy, int B, int H, int W, int cin,
                                int cout_total, int co_first, int co_count, void* stream) {
  if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout_total <= 0) return GFC_ERR_INVALID;
  if (co_first < 0 || co_count <= 0 || co_first + co_count > cout_total || co_first % 32 || ldy < co_count) return GFC_ERR_INVALID;
  if (cin % 4 || (mean != nullptr) != (rstd != nullptr)) return GFC_ERR_INVALID;
  Disk5Args a;
  a.x = x; a.mean = mean; a.rstd = rstd; a.prelu = prelu; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.cin = cin;
  a.chunks = (cin + DKC - 1) / DKC;
  a.co_pad = (cout_total + 31) / 32 * 32;
  a.tiles_x = (W + DT - 1) / DT;
  a.tiles_y = (H + DT - 1) / DT;
  const long long ntiles = (long long)a.tiles_x * a.tiles_y * B;
  if (ntiles > 0x7FFFFFFFll) return GFC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // output channels in blocks of 64; a remainder of <= 32 channels goes through the one-tile variant
  // (a single left-over channel -- 129 = 64 + 64 + 1, DISK's heat-map -- goes to the per-pixel VALU kernel).  Every block stays inside the
  // co_pad rows of a packed slice: co_first is a multiple of 32 and a 64-wide block is only used for > 32 channels.
  const int full = co_count / 64, rem = co_count - full * 64;
  int done = 0;
  if (full > 0 || rem > 32) {
    const int blocks = rem > 32 ? full + 1 : full;
    a.w = w_packed + (size_t)co_first * DKC;  // row offset inside every [co_pad][16] slice
    a.bias = bias ? bias + co_first : nullptr;
    a.y = y;
    a.cout = rem > 32 ? co_count : full * 64;
    hipLaunchKernelGGL((disk_conv5x5_kernel<2>), dim3((unsigned)ntiles, blocks), dim3(256), 0, st, a);
    done = a.cout;
  }
  if (done < co_count) {
    a.w = w_packed + (size_t)(co_first + done) * DKC;
    a.bias = bias ? bias + co_first + done : nullptr;
    a.y = y + done;
    a.cout = co_count - done;
    if (a.cout == 1) hipLaunchKernelGGL(disk_conv5x5_c1_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((disk_conv5x5_kernel<1>), dim3((unsigned)ntiles, 1), dim3(256), 0, st, a);
  }
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// InstanceNorm2d statistics: x [B, HW, C] contiguous -> mean[B,C], rstd[B,C] = 1/sqrt(biased var + eps).
// Stage 1: grid (slabs, B); thread = (pixel lane, channel quad), float64 partial sums -> part[B][slab][C][2].
// Stage 2: one thread per (image, channel) adds the slabs in order.
// ------------------------------------------------------------------------------------------------------------------
#define DSTAT_SLABS 64
__global__ __launch_bounds__(DSTAT_T) void disk_instnorm_partial_kernel(const float* __restrict__ x, long long HW, int C,
                                                                        double* __restrict__ part) {
  __shared__ double red[DSTAT_T * 8];
  const int q = C >> 2;            // channel quads; DSTAT_T % q == 0 (checked by the host)
  const int tid = threadIdx.x, cq = tid % q, pl = tid / q, npl = DSTAT_T / q;
  const int slab = blockIdx.x, b = blockIdx.y;
  const long long per = (HW + DSTAT_SLABS - 1) / DSTAT_SLABS;
  const long long p0 = (long long)slab * per, p1 = p0 + per < HW ? p0 + per : HW;
  const float* xb = x + (size_t)b * HW * C + cq * 4;
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  for (long long p = p0 + pl; p < p1; p += npl) {
    const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * C);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s[j]; red[tid * 8 + 4 + j] = ss[j]; }
  __syncthreads();
  if (tid < q) {  // thread cq adds its pixel lanes in order
    double ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < npl; ++l)
#pragma unroll
      for (int j = 0; j < 8; ++j) ts[j] += red[(l * q + tid) * 8 + j];
    double* o = part + (((size_t)b * DSTAT_SLABS + slab) * C + tid * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = ts[j]; o[2 * j + 1] = ts[4 + j]; }
  }
}

__global__ void disk_instnorm_finalize_kernel(const double* __restrict__ part, long long HW, int C, int B, float eps,
                                              float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  double s = 0, ss = 0;
  for (int k = 0; k < DSTAT_SLABS; ++k) {
    const double* p = part + (((size_t)b * DSTAT_SLABS + k) * C + c) * 2;
    s += p[0];
    ss += p[1];
  }
  const double m = s / (double)HW;
  double var = ss / (double)HW - m * m;
  if (var < 0) var = 0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

extern "C" size_t gfc_disk_instnorm_workspace_bytes(int B, int C) {
  if (B <= 0 || C <= 0) return 0;
  return gfc_align((size_t)B * DSTAT_SLABS * C * 2 * sizeof(double));
}

extern "C" int gfc_disk_instnorm_stats(const float* x, int B, int H, int W, int C, float eps, float* mean, float* rstd,
                                       void* ws, size_t ws_bytes, void* stream) {
  if (!x || !mean || !rstd || !ws || B <= 0 || H <= 0 || W <= 0 || C <= 0) return GFC_ERR_INVALID;
  if (C % 4 || DSTAT_T % (C / 4)) return GFC_ERR_UNSUPPORTED;
  if (ws_bytes < gfc_disk_instnorm_workspace_bytes(B, C)) return GFC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(disk_instnorm_partial_kernel, dim3(DSTAT_SLABS, B), dim3(DSTAT_T), 0, st, x, HW, C, (double*)ws);
  hipLaunchKernelGGL(disk_instnorm_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, (const double*)ws, HW, C, B,
                     eps, mean, rstd);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// glue between the levels (NHWC, float4 over channels)
// ------------------------------------------------------------------------------------------------------------------
// F.avg_pool2d(x, 2): x [B,H,W,C] with channel stride ldx -> y [B,H/2,W/2,C] contiguous
__global__ __launch_bounds__(256) void disk_avgpool2_kernel(const float* __restrict__ x, int ldx, int B, int H, int W, int C,
                                                            float* __restrict__ y) {
  const int q = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)B * Ho * Wo * q;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % q);
    long long t = i / q;
    const int xo = (int)(t % Wo);
    t /= Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    const float* p = x + (((size_t)b * H + 2 * yo) * W + 2 * xo) * ldx + cq * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(p), v01 = *reinterpret_cast<const float4*>(p + ldx);
    const float4 v10 = *reinterpret_cast<const float4*>(p + (size_t)W * ldx);
    const float4 v11 = *reinterpret_cast<const float4*>(p + (size_t)W * ldx + ldx);
    float4 o;  // ATen's order: the window is summed row by row, then divided by 4
    o.x = (((v00.x + v01.x) + v10.x) + v11.x) * 0.25f;
    o.y = (((v00.y + v01.y) + v10.y) + v11.y) * 0.25f;
    o.z = (((v00.z + v01.z) + v10.z) + v11.z) * 0.25f;
    o.w = (((v00.w + v01.w) + v10.w) + v11.w) * 0.25f;
    *reinterpret_cast<float4*>(y + (size_t)i * 4) = o;
  }
}

// F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False): x [B,h,w,C] contiguous ->
// y [B,2h,2w,*] with channel stride ldy (the first C channels of a concatenated tensor)
__global__ __launch_bounds__(256) void disk_upsample2_kernel(const float* __restrict__ x, int B, int h, int w, int C,
                                                             float* __restrict__ y, int ldy) {
  const int q = C >> 2, Ho = 2 * h, Wo = 2 * w;
  const long long total = (long long)B * Ho * Wo * q;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % q);
    long long t = i / q;
    const int xo = (int)(t % Wo);
    t /= Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    // source coordinate (dst + 0.5) / 2 - 0.5, clamped at 0 (ATen area_pixel_compute_source_index)
    float sy = (yo + 0.5f) * 0.5f - 0.5f, sx = (xo + 0.5f) * 0.5f - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* xb = x + (size_t)b * h * w * C + cq * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * w + x0) * C);
    const float4 v01 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * w + x1) * C);
    const float4 v10 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * w + x0) * C);
    const float4 v11 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * w + x1) * C);
    float4 o;  // ATen: hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *reinterpret_cast<float4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * ldy + cq * 4) = o;
  }
}

// image [B,3,H,W] -> [B,H,W,4] (fourth channel zero): the input of the first convolution
__global__ __launch_bounds__(256) void disk_nchw_to_nhwc4_kernel(const float* __restrict__ img, int B, long long HW,
                                                                 float* __restrict__ y) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / HW, p = i - b * HW;
    const float* s = img + (size_t)b * 3 * HW + p;
    *reinterpret_cast<float4*>(y + (size_t)i * 4) = make_float4(s[0], s[HW], s[2 * HW], 0.f);
  }
}

static inline unsigned disk_grid(long long total) {
  long long g = (total + 255) / 256;
  return (unsigned)(g > 65535 * 4 ? 65535 * 4 : (g < 1 ? 1 : g));
}

extern "C" int gfc_disk_avgpool2(const float* x, int ldx, int B, int H, int W, int C, float* y, void* stream) {
  if (!x || !y || B <= 0 || H < 2 || W < 2 || C <= 0 || C % 4 || ldx < C || ldx % 4 || H % 2 || W % 2) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(disk_avgpool2_kernel, dim3(disk_grid((long long)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, B, H, W, C, y);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_disk_upsample2(const float* x, int B, int h, int w, int C, float* y, int ldy, void* stream) {
  if (!x || !y || B <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 4 || ldy < C || ldy % 4) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(disk_upsample2_kernel, dim3(disk_grid((long long)B * 4 * h * w * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, B, h, w, C, y, ldy);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_disk_nchw3_to_nhwc4(const float* image, int B, int H, int W, float* y, void* stream) {
  if (!image || !y || B <= 0 || H <= 0 || W <= 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(disk_nchw_to_nhwc4_kernel, dim3(disk_grid((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                     image, B, (long long)H * W, y);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
