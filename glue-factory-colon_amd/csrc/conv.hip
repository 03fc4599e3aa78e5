// 3x3 convolutions of the SuperPoint encoder as fp32-MFMA implicit GEMM (NHWC).
//
// Replaces: VGGBlock conv -> ReLU -> BatchNorm(eval) [+ MaxPool2d(2,2)]
//           (reference gluefactory/models/extractors/superpoint_open.py:61-77,100-108)
//           and conv -> ReLU [+ pool] of gluefactory_nonfree/superpoint.py:214-224.
//
// One workgroup (4 waves) computes a 16x16 pixel tile x 64 output channels.  The input
// halo tile (18x18 pixels x 32 channels) sits in LDS; the A operand of tap (dy,dx) is the
// same LDS image read at a shifted pixel offset, so no im2col buffer exists.  K is walked
// as (channel chunk of 32) x (9 taps); per step the 64x32 weight slice is double-buffered
// in LDS and prefetched through registers while the MFMAs of the current step run.
// Each lane reads 4 consecutive k with one ds_read_b128; MFMA s of a group pairs k = 4h+s of
// both lane halves (any k order is a valid dot product as long as A and B agree).
#include <stdlib.h>

#include "common.h"

#define CT 16              // output tile edge (pixels)
#define CH (CT + 2)        // halo tile edge
#define CKC 32             // channel granularity of the API (every variant's chunk divides it)
#define CNB 64             // output channels per workgroup

__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int total = cout * cin * 9;
  if (i >= total) return;
  int ci = i % cin;
  int co = (i / cin) % cout;
  int tap = i / (cin * cout);
  out[i] = w[((size_t)co * cin + ci) * 9 + tap];
}

// (image * [0.299, 0.587, 0.114]).sum(1): superpoint_open.py:128-130
__global__ void rgb_to_gray_kernel(const float* __restrict__ img, float* __restrict__ out, int B, long long hw) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * hw) return;
  long long b = i / hw, p = i % hw;
  const float* s = img + b * 3 * hw + p;
  float v = s[0] * 0.299f;
  v += s[hw] * 0.587f;
  v += s[2 * hw] * 0.114f;
  out[i] = v;
}

// First layer, cin = 1: direct VALU convolution (0.7 % of the FLOPs, bound by the 64-channel write).
// 4 threads per pixel, 16 channels each; w [9][64].
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y, int B,
                                                         int H, int W, int relu) {
  __shared__ float ws[9 * 64 + 3 * 64];
  for (int i = threadIdx.x; i < 9 * 64; i += 256) ws[i] = w[i];
  if (threadIdx.x < 64) {
    ws[576 + threadIdx.x] = bias[threadIdx.x];
    ws[640 + threadIdx.x] = scale ? scale[threadIdx.x] : 1.f;
    ws[704 + threadIdx.x] = shift ? shift[threadIdx.x] : 0.f;
  }
  __syncthreads();
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  long long pix = t >> 2;
  int cg = (int)(t & 3) * 16;
  if (pix >= (long long)B * H * W) return;
  int px = (int)(pix % W);
  int py = (int)((pix / W) % H);
  const float* xb = x + (pix - (long long)py * W - px);
  float v[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      int yy = py + dy - 1, xx = px + dx - 1;
      v[dy * 3 + dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xb[(long long)yy * W + xx] : 0.f;
    }
  float* yo = y + pix * 64 + cg;
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = cg + c4 * 4 + j;
      float a = 0.f;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) a = fmaf(v[tp], ws[tp * 64 + c], a);
      a += ws[576 + c];
      if (relu) a = fmaxf(a, 0.f);
      o[j] = a * ws[640 + c] + ws[704 + c];
    }
    *reinterpret_cast<float4*>(yo + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;
  int B, H, W, cin, cout, relu;
  int tiles_x, tiles_y;
  // STEM variant only: the cin = 1 layer in front (conv1a), recomputed on the halo tile in LDS
  const float* w1;   // [9][64]
  const float* b1;
  const float* s1;   // nullable (no BN)
  const float* t1;
};

// STEM = true: `a.x` is the 1-channel image [B,H,W]; the first layer (1 -> 64, conv + ReLU + BN) is
// evaluated on the fly for the 18x18 halo pixels from a 20x20 image patch in LDS instead of being
// read from HBM (superpoint_open.py:100-103: backbone.0.0 feeding backbone.0.1).  This removes the
// largest activation of the network (B*H*W*64 floats written and read back) and one launch.
#define CIM (CT + 4)
#ifndef CONV_LOAD_TAP
#define CONV_LOAD_TAP 2   // tap at which the next input chunk's global loads are issued (stored after tap 8)
#endif
// KC = channels per chunk: 32 (65 KB LDS, 2 workgroups / CU) or 16 (36 KB, 3 workgroups / CU)
// PERSIST = workgroups walk several work items with the next item's operands prefetched (see below); without it the
// grid has one workgroup per item and the hand-over code (and its registers) is compiled out.
template <bool POOL, bool STEM, int KC, bool PERSIST>
__global__ __launch_bounds__(256, (KC == 16 && !PERSIST) ? 3 : 2) void conv3x3_mfma_kernel(ConvArgs a) {
  constexpr int CLD = KC + 4;                        // LDS row stride in floats (bank-conflict padding, 16B aligned)
  constexpr int C4 = KC / 4;                         // float4 per staged row
  constexpr int NI = (CH * CH * C4 + 255) / 256;     // input float4 per thread per chunk
  constexpr int NWR = CNB * C4 / 256;                // weight float4 per thread per step (1 or 2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_s = smem;                   // [CH*CH][CLD]
  float* w_s = smem + CH * CH * CLD;    // [2][CNB][CLD]
  float* img_s = w_s + 2 * CNB * CLD;   // [CIM*CIM] (STEM only)
  float* c1_s = img_s + CIM * CIM;      // conv1a constants (STEM only): w1 [9][64], b1 [64], s1 [64], t1 [64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int h = lane >> 5;

  // Persistent workgroups: work item w = tile + ntiles * (output-channel block); a workgroup walks w = blockIdx.x,
  // blockIdx.x + gridDim.x, ...  While the last K step of one item runs, the halo tile and first weight slice of
  // the next item are already on their way (global -> registers), so neither the load latency of the prologue nor the
  // store drain of the epilogue leaves the MFMA pipe idle (in-kernel stamps: they were 16 % + 13 % of a workgroup's
  // life at 64 -> 64 channels).
  const int cin = a.cin;
  const int nchunks = cin / KC;
  const int nsteps = nchunks * 9;
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  const int nitems = ntiles * (a.cout / CNB);
  int item = blockIdx.x;
  // current item (x0, y0, b, nb, xin, wbase) and the next one (n*)
  int x0, y0, b, nb, nx0 = 0, ny0 = 0, nb_b = 0, nnb = 0;
  const float* xin;
  const float* wbase;
  const float* nxin = nullptr;
  const float* nwbase = nullptr;
#define CONV_DECODE(w_, x0_, y0_, b_, nb_, xin_, wbase_)                                    \
  do {                                                                                      \
    int t_ = (w_) % ntiles;                                                                 \
    nb_ = (w_) / ntiles;                                                                    \
    x0_ = (t_ % a.tiles_x) * CT;                                                            \
    t_ /= a.tiles_x;                                                                        \
    y0_ = (t_ % a.tiles_y) * CT;                                                            \
    b_ = t_ / a.tiles_y;                                                                    \
    xin_ = a.x + (size_t)b_ * a.H * a.W * (STEM ? 1 : cin);                                 \
    wbase_ = a.w + (size_t)nb_ * CNB * cin; /* + tap*cout*cin + co*cin + chunk*KC */         \
  } while (0)
  CONV_DECODE(item, x0, y0, b, nb, xin, wbase);

  // ---- staging (register prefetch).  Macros, not lambdas: the prefetch registers must stay in
  // VGPRs (a by-reference lambda capture demoted them to a private-memory array). ----
  float4 wreg0, wreg1;
  float4 ireg[NI];
  // weight rows st_co (, st_co + 32).  16-channel chunks: a ds_write_b128 lane group covers two rows; at a pitch of 20
  // floats adjacent rows overlap by 4 of the 32 banks, rows four apart do not -> permute rows inside blocks of 8
  const int stq_ = tid / C4;
  const int st_co = C4 == 4 ? ((stq_ & ~7) | ((stq_ & 1) << 2) | ((stq_ >> 1) & 3)) : stq_;
  const int st_c4 = (tid % C4) * 4;
  wreg1 = make_float4(0.f, 0.f, 0.f, 0.f);
#define CONV_LOAD_W(base_, step_)                                                               \
  do {                                                                                      \
    const int ch_ = (step_) / 9, tp_ = (step_) - ch_ * 9;                                   \
    const float* src_ = (base_) + (size_t)tp_ * a.cout * cin + ch_ * KC + st_c4;            \
    wreg0 = *reinterpret_cast<const float4*>(src_ + (size_t)st_co * cin);                   \
    if constexpr (NWR == 2) wreg1 = *reinterpret_cast<const float4*>(src_ + (size_t)(st_co + 32) * cin); \
  } while (0)
#define CONV_STORE_W(buf_)                                                                  \
  do {                                                                                      \
    float* dst_ = w_s + (buf_) * CNB * CLD + st_co * CLD + st_c4;                           \
    *reinterpret_cast<float4*>(dst_) = wreg0;                                               \
    if constexpr (NWR == 2) *reinterpret_cast<float4*>(dst_ + 32 * CLD) = wreg1;            \
  } while (0)
#define CONV_LOAD_IN(chunk_, xin_, y0_, x0_)                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                       \
    const int idx_ = tid + 256 * i_;                                                        \
    const int p_ = idx_ / C4;                                                               \
    const int gy_ = (y0_) - 1 + p_ / CH, gx_ = (x0_) - 1 + p_ % CH;                         \
    float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                            \
    if (idx_ < CH * CH * C4 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W)              \
      v_ = *reinterpret_cast<const float4*>((xin_) + ((size_t)gy_ * a.W + gx_) * cin + (chunk_) * KC + st_c4); \
    ireg[i_] = v_;                                                                          \
  }
#define CONV_FILL_IN(chunk_, y0_, x0_)                                                      \
  {                                                                                         \
    const int c0_ = (chunk_) * KC + st_c4;                                                  \
    /* conv1a constants from LDS (staged once per workgroup): a global load here costs its L2 latency in front  \
       of every chunk's evaluation (stem +1 %) */                                           \
    float4 wv_[9];                                                                          \
    _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_)                                        \
        wv_[t_] = *reinterpret_cast<const float4*>(c1_s + t_ * 64 + c0_);                   \
    const float4 b1_ = *reinterpret_cast<const float4*>(c1_s + 576 + c0_);                  \
    const float4 s1_ = *reinterpret_cast<const float4*>(c1_s + 640 + c0_);                  \
    const float4 t1_ = *reinterpret_cast<const float4*>(c1_s + 704 + c0_);                  \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                     \
      const int idx_ = tid + 256 * i_;                                                      \
      const int p_ = idx_ / C4;                                                             \
      const int py_ = p_ / CH, px_ = p_ % CH;                                               \
      const int gy_ = (y0_) - 1 + py_, gx_ = (x0_) - 1 + px_;                               \
      float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                          \
      if (idx_ < CH * CH * C4 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W) {          \
        _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) {                                  \
          const float f_ = img_s[(py_ + t_ / 3) * CIM + px_ + t_ % 3];                      \
          v_.x = fmaf(f_, wv_[t_].x, v_.x);                                                 \
          v_.y = fmaf(f_, wv_[t_].y, v_.y);                                                 \
          v_.z = fmaf(f_, wv_[t_].z, v_.z);                                                 \
          v_.w = fmaf(f_, wv_[t_].w, v_.w);                                                 \
        }                                                                                   \
        v_.x = fmaxf(v_.x + b1_.x, 0.f) * s1_.x + t1_.x;                                    \
        v_.y = fmaxf(v_.y + b1_.y, 0.f) * s1_.y + t1_.y;                                    \
        v_.z = fmaxf(v_.z + b1_.z, 0.f) * s1_.z + t1_.z;                                    \
        v_.w = fmaxf(v_.w + b1_.w, 0.f) * s1_.w + t1_.w;                                    \
      }                                                                                     \
      ireg[i_] = v_;                                                                        \
    }                                                                                       \
  }
#define CONV_STORE_IN()                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                       \
    const int idx_ = tid + 256 * i_;                                                        \
    if (idx_ < CH * CH * C4) *reinterpret_cast<float4*>(in_s + (idx_ / C4) * CLD + st_c4) = ireg[i_]; \
  }

  // this lane's two pixel rows in the halo image (tap (0,0) corner), and weight rows
  int a_off[2], b_off[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    // MFMA tile (wave, mt) = image rows 2*wave + mt and that + 8, 16 pixels each.  Rows EIGHT apart (not adjacent):
    // ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} on 64 banks, and with an
    // 18-pixel halo pitch two adjacent rows put lanes 12,13 and 26,27 of a group on the same banks (2-way conflict
    // on every A read: PMC showed 26-39 % of the LDS cycles were conflict cycles); 8 rows = 144 pixels = 0 mod 16.
    int py = 2 * wave + mt + 8 * (l31 >> 4), px = l31 & 15;
    a_off[mt] = (py * CH + px) * CLD + 4 * h;
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) b_off[nt] = (nt * 32 + l31) * CLD + 4 * h;

  f32x16 acc[2][2];
  // STEM: 20x20 image patch around the tile (zero outside the image = conv1a's own zero padding), 2 floats / thread
  float ip0 = 0.f, ip1 = 0.f;
#define CONV_LOAD_IMG(img_, y0_, x0_)                                                       \
  do {                                                                                      \
    const int i0_ = tid, i1_ = tid + 256;                                                   \
    const int gy0_ = (y0_) - 2 + i0_ / CIM, gx0_ = (x0_) - 2 + i0_ % CIM;                   \
    const int gy1_ = (y0_) - 2 + i1_ / CIM, gx1_ = (x0_) - 2 + i1_ % CIM;                   \
    ip0 = (gy0_ >= 0 && gy0_ < a.H && gx0_ >= 0 && gx0_ < a.W) ? (img_)[(size_t)gy0_ * a.W + gx0_] : 0.f; \
    ip1 = (i1_ < CIM * CIM && gy1_ >= 0 && gy1_ < a.H && gx1_ >= 0 && gx1_ < a.W)          \
              ? (img_)[(size_t)gy1_ * a.W + gx1_] : 0.f;                                    \
  } while (0)
#define CONV_STORE_IMG()                                                                    \
  do {                                                                                      \
    img_s[tid] = ip0;                                                                       \
    if (tid + 256 < CIM * CIM) img_s[tid + 256] = ip1;                                      \
  } while (0)

  if constexpr (STEM) {
    CONV_LOAD_IMG(xin, y0, x0);
    CONV_STORE_IMG();
    for (int i = tid; i < 768; i += 256)
      c1_s[i] = i < 576 ? a.w1[i] : i < 640 ? a.b1[i - 576] : i < 704 ? (a.s1 ? a.s1[i - 640] : 1.f)
                                                                         : (a.t1 ? a.t1[i - 704] : 0.f);
    __syncthreads();
    CONV_FILL_IN(0, y0, x0);
  } else {
    CONV_LOAD_IN(0, xin, y0, x0);
  }
  CONV_LOAD_W(wbase, 0);

  while (true) {
  CONV_STORE_IN();
  CONV_STORE_W(0);
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const int next_item = item + gridDim.x;
  const bool more = PERSIST && next_item < nitems;
  if (more) CONV_DECODE(next_item, nx0, ny0, nb_b, nnb, nxin, nwbase);
  if constexpr (STEM) {
    if (more) CONV_LOAD_IMG(nxin, ny0, nx0);  // 2 registers; stored once this item's last FILL_IN has run
  }

  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / 9, tap = step - chunk * 9;
    const bool has_next = step + 1 < nsteps;
    const bool new_chunk = has_next && tap == 8;
    // weight slice of the next step (or slice 0 of the next item): requested now, written to the other LDS buffer
    // after this step's MFMAs.  (Writing it at the top of the following step instead -- "write after the barrier",
    // +1.4 % on the bare loop shape of tools/micro/mfma_feed.hip -- measured -4 % here: vmcnt retires in order, so
    // the early write also waits for the older halo-tile loads and the previous item's output stores.)
    if (has_next) CONV_LOAD_W(wbase, step + 1);
    else if (more) CONV_LOAD_W(nwbase, 0);
    if constexpr (STEM) {
      if (new_chunk) CONV_FILL_IN(chunk + 1, y0, x0);
      if (more && step == nsteps - 1) CONV_FILL_IN(0, ny0, nx0);
    } else {
      // the next input chunk (of this item, or chunk 0 of the next item) is requested LOAD_TAP steps before the
      // chunk boundary that stores it: first-touch HBM latency is longer than one step under load
      if (tap == CONV_LOAD_TAP) {
        if (chunk + 1 < nchunks) { CONV_LOAD_IN(chunk + 1, xin, y0, x0); }
        else if (more) { CONV_LOAD_IN(0, nxin, ny0, nx0); }
      }
    }

    const int dy = tap / 3, dx = tap - dy * 3;
    const float* ap = in_s + (dy * CH + dx) * CLD;
    const float* bp = w_s + (step & 1) * CNB * CLD;
#pragma unroll
    for (int g = 0; g < KC / 8; ++g) {
      float4 af[2], bf[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const float4*>(ap + a_off[mt] + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bf[nt] = *reinterpret_cast<const float4*>(bp + b_off[nt] + 8 * g);
      // k-major order: consecutive MFMAs go to different accumulators (same per-accumulator order of the 4 k pairs)
#define CONV_MFMA4(c_)                                                   \
  acc[0][0] = mfma32(af[0].c_, bf[0].c_, acc[0][0]);                     \
  acc[0][1] = mfma32(af[0].c_, bf[1].c_, acc[0][1]);                     \
  acc[1][0] = mfma32(af[1].c_, bf[0].c_, acc[1][0]);                     \
  acc[1][1] = mfma32(af[1].c_, bf[1].c_, acc[1][1]);
      CONV_MFMA4(x) CONV_MFMA4(y) CONV_MFMA4(z) CONV_MFMA4(w)
#undef CONV_MFMA4
    }
    if (has_next) CONV_STORE_W((step + 1) & 1);
    if (new_chunk) {
      __syncthreads();  // every wave is done reading the old input chunk
      CONV_STORE_IN();
    }
    if constexpr (STEM) {
      if (more && step == nsteps - 9) CONV_STORE_IMG();  // visible after this step's barrier, read 8 steps later
    }
    __syncthreads();
  }

  // ---- epilogue: bias, ReLU, per-channel affine (BN), optional 2x2 max-pool ----
  if (!POOL) {
    // Un-pooled outputs are 4x the pooled volume: transpose each 32-pixel x 64-channel block through a
    // per-wave LDS patch (the halo tile is dead by now) and store whole float4 channel groups --
    // 16 store instructions per wave instead of 64, each covering 4 pixels x 256 contiguous bytes.
    constexpr int ELD = 68;
    float* patch = in_s + wave * 32 * ELD;
    const int er = lane >> 4, ec = (lane & 15) * 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int co = nb * CNB + nt * 32 + l31;
        float bi = a.bias[co];
        float sc = a.scale ? a.scale[co] : 1.f;
        float sh = a.shift ? a.shift[co] : 0.f;
        asm volatile("" : "+v"(bi), "+v"(sc), "+v"(sh));
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
        const int p = er + 4 * i;  // pixel inside the 2x16 block
        const int gy = y0 + 2 * wave + mt + 8 * (p >> 4), gx = x0 + (p & 15);
        const float4 v = *reinterpret_cast<const float4*>(patch + p * ELD + ec);
        if (gy < a.H && gx < a.W)
          *reinterpret_cast<float4*>(a.y + (((size_t)b * a.H + gy) * a.W + gx) * a.cout + nb * CNB + ec) = v;
      }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int co = nb * CNB + nt * 32 + l31;
      float bi = a.bias[co];
      float sc = a.scale ? a.scale[co] : 1.f;
      float sh = a.shift ? a.shift[co] : 0.f;
      // waited for once, here: otherwise hipcc re-waits (vmcnt(0)) inside every predicated store block below, and each
      // of those waits also waits for the previous store (see gemm.hip: gemm_epilogue)
      asm volatile("" : "+v"(bi), "+v"(sc), "+v"(sh));
      // rows 2*wave (mt 0) and 2*wave + 1 (mt 1) are vertical neighbours, register r and r+1 horizontal ones;
      // registers 8..15 are the same for the rows 8 further down
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
        const int pxl = (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;  // even column inside the 16-wide tile
        const int oy = (y0 >> 1) + wave + 4 * (r >> 3), ox = (x0 >> 1) + (pxl >> 1);
        if (oy < Ho && ox < Wo) a.y[(((size_t)b * Ho + oy) * Wo + ox) * a.cout + co] = m;
      }
    }
  }
  if (!more) break;
  item = next_item; x0 = nx0; y0 = ny0; b = nb_b; nb = nnb; xin = nxin; wbase = nwbase;
  if constexpr (!POOL) __syncthreads();  // the patch reads above are complete before the next halo tile lands in LDS
  }  // persistent loop over work items
}

extern "C" int gfc_pack_conv3x3(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
  if (!w_oihw || !w_packed || cout <= 0 || cin <= 0) return GFC_ERR_INVALID;
  int total = cout * cin * 9;
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_packed, cout, cin);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

int gfc_rgb_to_gray(const float* img, float* out, int B, int H, int W, hipStream_t stream) {
  long long hw = (long long)H * W;
  long long total = hw * B;
  hipLaunchKernelGGL(rgb_to_gray_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, out, B, hw);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

static int launch_conv(ConvArgs a, bool pool, bool stem, hipStream_t st);

extern "C" int gfc_conv3x3(const float* x, const float* w_packed, const float* bias, const float* scale,
                           const float* shift, float* y, int B, int H, int W, int cin, int cout, int relu, int pool,
                           void* stream) {
  if (!x || !w_packed || !bias || !y || B <= 0 || H <= 0 || W <= 0) return GFC_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return GFC_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (cin == 1) {
    if (cout != 64 || pool) return GFC_ERR_UNSUPPORTED;
    long long threads = (long long)B * H * W * 4;
    hipLaunchKernelGGL(conv3x3_c1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x, w_packed, bias,
                       scale, shift, y, B, H, W, relu);
    GFC_LAUNCH_CHECK();
    return GFC_OK;
  }
  if (cin % CKC != 0 || cout % CNB != 0) return GFC_ERR_UNSUPPORTED;
  ConvArgs a = {};
  a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y;
  a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.w1 = a.b1 = a.s1 = a.t1 = nullptr;
  return launch_conv(a, pool != 0, false, st);
}

template <bool POOL, bool STEM, int KC, bool PERSIST>
static int launch_conv_t(const ConvArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)(CH * CH * (KC + 4) + 2 * CNB * (KC + 4) + (STEM ? CIM * CIM + 768 : 0)) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};  // per (instantiation, device): runtime.h
  if (lds > 64 * 1024) gfc_allow_dynamic_lds((const void*)conv3x3_mfma_kernel<POOL, STEM, KC, PERSIST>, lds, lds_ok);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<POOL, STEM, KC, PERSIST>), grid, dim3(256), lds, st, a);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

static int launch_conv(ConvArgs a, bool pool, bool stem, hipStream_t st) {
  a.tiles_x = (a.W + CT - 1) / CT;
  a.tiles_y = (a.H + CT - 1) / CT;
  const int ncu = gfc_device_cus();
  // Variant per layer kind (same-box A/B, tools/ab_build.sh + tools/bench_kernels.py --only conv, 32 VGA images):
  //  * pooled layers and the stem: 16-channel LDS chunks (36 KB: three workgroups per CU), one workgroup per item
  //    (persistent workgroups cost the third resident workgroup in registers: stem -1.5 %, conv2b -1 %);
  //  * un-pooled layers: 32-channel chunks, persistent workgroups (+1.5..2 %; small grids such as conv4a at
  //    60x80 +16 %, because two resident workgroups per CU then share the items evenly).
  // GFC_CONV_KC=32|16 and GFC_CONV_PERSIST=0|1 force a variant.
  const int forced_kc = gfc_knobs().conv_kc, forced_p = gfc_knobs().conv_persist;
  const int kc = forced_kc ? forced_kc : (pool ? 16 : 32);
  const bool persist = forced_p >= 0 ? forced_p != 0 : !pool;
  const long long nitems = (long long)a.tiles_x * a.tiles_y * a.B * (a.cout / CNB);
  const long long resident = (long long)ncu * 2;
  dim3 grid((unsigned)(persist && nitems > resident ? resident : nitems));
#define CONV_DISPATCH(KC_, P_)                                                      \
  do {                                                                              \
    if (stem) return launch_conv_t<true, true, KC_, P_>(a, grid, st);               \
    if (pool) return launch_conv_t<true, false, KC_, P_>(a, grid, st);              \
    return launch_conv_t<false, false, KC_, P_>(a, grid, st);                       \
  } while (0)
  if (kc == 16) {
    if (persist) CONV_DISPATCH(16, true);
    CONV_DISPATCH(16, false);
  }
  if (persist) CONV_DISPATCH(32, true);
  CONV_DISPATCH(32, false);
#undef CONV_DISPATCH
}

// conv1a (1 -> 64) + conv1b (64 -> 64) + 2x2 max-pool in one launch: gray image [B,H,W] -> [B,H/2,W/2,64].
extern "C" int gfc_sp_stem(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                           const float* w2_packed, const float* b2, const float* s2, const float* t2, float* y, int B,
                           int H, int W, void* stream) {
  if (!image || !w1 || !b1 || !w2_packed || !b2 || !y || B <= 0 || H < 2 || W < 2) return GFC_ERR_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return GFC_ERR_INVALID;
  ConvArgs a = {};
  a.x = image; a.w = w2_packed; a.bias = b2; a.scale = s2; a.shift = t2; a.y = y;
  a.B = B; a.H = H; a.W = W; a.cin = 64; a.cout = 64; a.relu = 1;
  a.w1 = w1; a.b1 = b1; a.s1 = s1; a.t1 = t1;
  return launch_conv(a, true, true, (hipStream_t)stream);
}
