// fp32-MFMA "NT" GEMM  Y[M,N] = [A0|A1][M,K] * W[N,K]^T  with fused epilogues.
//
// Replaces every nn.Linear on the LightGlue path (reference
// gluefactory/models/matchers/lightglue.py:139-148,158,163-164,181-189,276-288: F.linear -> addmm),
// the 1x1 convolutions of the SuperPoint heads (superpoint_open.py:112-118, NHWC makes them
// plain GEMMs) and, in batched form, einsum("bmd,bnd->bmn") (lightglue.py:285).
//
// Workgroup = 4 waves = 128x128 output tile, K stepped by 32 through LDS (row stride 36 floats:
// conflict-free ds_read_b128 of 4 consecutive k per lane).  Each wave owns a 64x64 sub-tile =
// 2x2 MFMA 32x32 tiles; per 8-deep k group it issues 4 LDS reads and 16 v_mfma_f32_32x32x2_f32.
// The next K tile is prefetched global->registers while the current one is multiplied.
#include "common.h"

#define GBM 128
#define GBN 128
#define GBK 32
#define GLD (GBK + 4)

struct GemmArgs {
  const float* A0;
  const float* A1;
  const float* W;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  const float* rot_cos;
  const float* rot_sin;
  float* Y;
  long long strideA, strideW, strideY;  // batch strides (blockIdx.z)
  int lda0, lda1, ldw, ldy;
  int K0, K1, M, N, rot_cols;
  float alpha;
};

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[GBM * GLD];
  __shared__ __attribute__((aligned(16))) float Bs[GBN * GLD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const long long z = blockIdx.z;
  const float* A0 = g.A0 + z * g.strideA;
  const float* A1 = g.A1 ? g.A1 + z * g.strideA : nullptr;
  const float* W = g.W + z * g.strideW;
  float* Y = g.Y + z * g.strideY;
  const int K = g.K0 + g.K1;
  const int ktiles = K / GBK;

  // global -> register staging: 4 float4 of A and 4 of W per thread per K tile
  // (macros, not lambdas: the prefetch registers must stay in VGPRs, not in a private-memory array)
  const int s_c4 = (tid & 7) * 4;
  const int s_r0 = tid >> 3;  // rows s_r0 + 32*i
  const float* a_src[4];
  const float* a1_src[4];
  const float* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = min(m0 + s_r0 + 32 * i, g.M - 1);
    const int n = min(n0 + s_r0 + 32 * i, g.N - 1);
    a_src[i] = A0 + (size_t)r * g.lda0 + s_c4;
    a1_src[i] = A1 ? A1 + (size_t)r * g.lda1 + s_c4 : nullptr;
    w_src[i] = W + (size_t)n * g.ldw + s_c4;
  }
  float4 areg0, areg1, areg2, areg3, wreg0, wreg1, wreg2, wreg3;
#define GEMM_LOAD_TILE(kt)                                                                      \
  do {                                                                                          \
    const int k0_ = (kt) * GBK;                                                                 \
    if (k0_ < g.K0) {                                                                           \
      areg0 = *reinterpret_cast<const float4*>(a_src[0] + k0_);                                 \
      areg1 = *reinterpret_cast<const float4*>(a_src[1] + k0_);                                 \
      areg2 = *reinterpret_cast<const float4*>(a_src[2] + k0_);                                 \
      areg3 = *reinterpret_cast<const float4*>(a_src[3] + k0_);                                 \
    } else {                                                                                    \
      areg0 = *reinterpret_cast<const float4*>(a1_src[0] + (k0_ - g.K0));                       \
      areg1 = *reinterpret_cast<const float4*>(a1_src[1] + (k0_ - g.K0));                       \
      areg2 = *reinterpret_cast<const float4*>(a1_src[2] + (k0_ - g.K0));                       \
      areg3 = *reinterpret_cast<const float4*>(a1_src[3] + (k0_ - g.K0));                       \
    }                                                                                           \
    wreg0 = *reinterpret_cast<const float4*>(w_src[0] + k0_);                                   \
    wreg1 = *reinterpret_cast<const float4*>(w_src[1] + k0_);                                   \
    wreg2 = *reinterpret_cast<const float4*>(w_src[2] + k0_);                                   \
    wreg3 = *reinterpret_cast<const float4*>(w_src[3] + k0_);                                   \
  } while (0)
#define GEMM_STORE_TILE()                                                                       \
  do {                                                                                          \
    float* as_ = As + s_r0 * GLD + s_c4;                                                        \
    float* bs_ = Bs + s_r0 * GLD + s_c4;                                                        \
    *reinterpret_cast<float4*>(as_) = areg0;                                                    \
    *reinterpret_cast<float4*>(as_ + 32 * GLD) = areg1;                                         \
    *reinterpret_cast<float4*>(as_ + 64 * GLD) = areg2;                                         \
    *reinterpret_cast<float4*>(as_ + 96 * GLD) = areg3;                                         \
    *reinterpret_cast<float4*>(bs_) = wreg0;                                                    \
    *reinterpret_cast<float4*>(bs_ + 32 * GLD) = wreg1;                                         \
    *reinterpret_cast<float4*>(bs_ + 64 * GLD) = wreg2;                                         \
    *reinterpret_cast<float4*>(bs_ + 96 * GLD) = wreg3;                                         \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const float* ap = As + (wm * 64 + l31) * GLD + 4 * h;
  const float* bp = Bs + (wn * 64 + l31) * GLD + 4 * h;

  GEMM_LOAD_TILE(0);
  GEMM_STORE_TILE();
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const bool has_next = kt + 1 < ktiles;
    if (has_next) GEMM_LOAD_TILE(kt + 1);
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      float4 af[2], bf[2];
      af[0] = *reinterpret_cast<const float4*>(ap + 8 * gk);
      af[1] = *reinterpret_cast<const float4*>(ap + 32 * GLD + 8 * gk);
      bf[0] = *reinterpret_cast<const float4*>(bp + 8 * gk);
      bf[1] = *reinterpret_cast<const float4*>(bp + 32 * GLD + 8 * gk);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = mfma32(af[mt].x, bf[nt].x, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].y, bf[nt].y, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].z, bf[nt].z, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].w, bf[nt].w, acc[mt][nt]);
        }
    }
    __syncthreads();
    if (has_next) {
      GEMM_STORE_TILE();
      __syncthreads();
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = n0 + wn * 64 + nt * 32 + l31;
    const bool col_ok = col < g.N;
    const int cc = col_ok ? col : g.N - 1;
    const float bi = g.bias ? g.bias[cc] : 0.f;
    const float sc = g.scale ? g.scale[cc] : 1.f;
    const float sh = g.shift ? g.shift[cc] : 0.f;
    const bool rot = g.rot_cos != nullptr && col < g.rot_cols;  // wave-uniform per 32-col tile (rot_cols % 64 == 0)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mt * 32 + acc_row(r, h);
        const bool row_ok = row < g.M;
        float v = acc[mt][nt][r] + bi;
        if (g.rot_cos != nullptr) {
          // rotary: out[d] = t[d]*cos[d] + rot(t)[d]*sin[d], rot(t)[2i] = -t[2i+1], rot(t)[2i+1] = t[2i]
          float other = __shfl_xor(v, 1);
          if (rot) {
            const int rr = row_ok ? row : g.M - 1;
            const int d = col & 63;
            float c = g.rot_cos[(size_t)rr * 64 + d], s = g.rot_sin[(size_t)rr * 64 + d];
            float rv = (col & 1) ? other : -other;
            v = v * c + rv * s;
          }
        }
        v = v * sc + sh;
        v *= g.alpha;
        if (row_ok && col_ok) {
          size_t o = (size_t)row * g.ldy + col;
          if (g.residual) v = g.residual[o] + v;
          Y[o] = v;
        }
      }
    }
  }
}

static int launch_gemm(const GemmArgs& g, int batch, hipStream_t st) {
  dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, batch);
  hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, st, g);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_linear(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1, const float* W, int ldw,
                          const float* bias, const float* scale, const float* shift, float alpha,
                          const float* residual, const float* rot_cos, const float* rot_sin, int rot_cols, float* Y,
                          int ldy, int M, int N, void* stream) {
  if (!A0 || !W || !Y || M <= 0 || N <= 0 || K0 <= 0 || K0 % GBK || K1 % GBK || K1 < 0) return GFC_ERR_INVALID;
  if ((K1 > 0) != (A1 != nullptr)) return GFC_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return GFC_ERR_INVALID;
  if ((rot_cos == nullptr) != (rot_sin == nullptr)) return GFC_ERR_INVALID;
  if (rot_cos && (rot_cols % 64 != 0)) return GFC_ERR_INVALID;
  if (lda0 % 4 || (A1 && lda1 % 4) || ldw % 4) return GFC_ERR_INVALID;  // 16-byte vector loads
  GemmArgs g;
  g.A0 = A0; g.A1 = A1; g.W = W; g.bias = bias; g.scale = scale; g.shift = shift; g.residual = residual;
  g.rot_cos = rot_cos; g.rot_sin = rot_sin; g.Y = Y;
  g.strideA = g.strideW = g.strideY = 0;
  g.lda0 = lda0; g.lda1 = lda1; g.ldw = ldw; g.ldy = ldy;
  g.K0 = K0; g.K1 = K1; g.M = M; g.N = N; g.rot_cols = rot_cols; g.alpha = alpha;
  return launch_gemm(g, 1, (hipStream_t)stream);
}

extern "C" int gfc_batched_nt(const float* A, int lda, long long strideA, const float* Bm, int ldb, long long strideB,
                              float* Y, int ldy, long long strideY, int M, int N, int K, int batch, void* stream) {
  if (!A || !Bm || !Y || M <= 0 || N <= 0 || K <= 0 || K % GBK || batch <= 0) return GFC_ERR_INVALID;
  if (lda % 4 || ldb % 4 || strideA % 4 || strideB % 4) return GFC_ERR_INVALID;
  GemmArgs g;
  g.A0 = A; g.A1 = nullptr; g.W = Bm; g.bias = nullptr; g.scale = nullptr; g.shift = nullptr; g.residual = nullptr;
  g.rot_cos = nullptr; g.rot_sin = nullptr; g.Y = Y;
  g.strideA = strideA; g.strideW = strideB; g.strideY = strideY;
  g.lda0 = lda; g.lda1 = 0; g.ldw = ldb; g.ldy = ldy;
  g.K0 = K; g.K1 = 0; g.M = M; g.N = N; g.rot_cols = 0; g.alpha = 1.f;
  return launch_gemm(g, batch, (hipStream_t)stream);
}
