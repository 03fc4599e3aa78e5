// ARCHIVED (round 4): the bf16x3-split GEMM that lived in csrc/gemm.hip between launch_gemm_dma and launch_gemm_t.
// Not compiled into libgfc_amd.so any more; it needs gemm.hip's GemmArgs / gemm_epilogue to build (paste it back there).
// See tools/micro/experiments/README.md for the last measured numbers.

// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENTAL, opt-in: the same GEMM on the bf16 matrix pipe at fp32 accuracy (see conv_split.hip for the
// arithmetic: three bf16 planes per operand, six bf16 MFMA products per fp32 product, fp32 accumulation).
// 128x128 tile, 16-deep K tile = one v_mfma_f32_32x32x16_bf16 k block.  W is split once at load time
// (gfc_pack_linear_split: [N][K/16][3 planes][16] bf16), A is split while it is staged into LDS.
// LDS row = 3 planes x 16 bf16 (96 B) + 16 B pad (pitch / 16 odd: conflict-free ds_read_b128).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x4 __attribute__((ext_vector_type(4)));
#define GS_ROW 112

__device__ __forceinline__ void gsplit3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

__global__ void pack_linear_split_kernel(const float* __restrict__ w, int ldw, __bf16* __restrict__ out, int N, int K) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * K) return;
  const int k = (int)(i % K), n = (int)(i / K);
  __bf16 hi, mid, lo;
  gsplit3(w[(size_t)n * ldw + k], hi, mid, lo);
  __bf16* o = out + (((size_t)n * (K / 16) + k / 16) * 3) * 16 + (k % 16);
  o[0] = hi; o[16] = mid; o[32] = lo;
}

__global__ __launch_bounds__(256, 2) void gemm_nt_split_kernel(GemmArgs g) {
  constexpr int NW = 2, MT = 2, BM = 128, BN = 128, BK = 16;
  constexpr int TILEB = (BM + BN) * GS_ROW;  // bytes per buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* sm = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / NW, wn = wave % NW;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const long long z = blockIdx.z;
  const float* A0 = g.A0 + z * g.strideA;
  const float* A1 = g.A1 ? g.A1 + z * g.strideA : nullptr;
  const char* Ws = reinterpret_cast<const char*>(g.W);  // split weights: row n = (K/16) blocks of 96 B
  float* Y = g.Y + z * g.strideY;
  const int K = g.K0 + g.K1, ktiles = K / BK;
  const size_t wrow = (size_t)(K / 16) * 96;

  // A staging: 128 rows x 4 float4; thread -> rows r, r + 64, channels c4..c4+3
  const int s_c4 = (tid & 3) * 4, s_r = tid >> 2;
  const size_t ar0 = (size_t)min(m0 + s_r, g.M - 1), ar1 = (size_t)min(m0 + s_r + 64, g.M - 1);
  float4 areg0, areg1, wreg0, wreg1, wreg2;
  // W staging: 128 rows x 6 pieces of 16 B = 768 pieces, 3 per thread
  const int q0 = tid, q1 = tid + 256, q2 = tid + 512;
  const char* wp0 = Ws + (size_t)min(n0 + q0 / 6, g.N - 1) * wrow + (q0 % 6) * 16;
  const char* wp1 = Ws + (size_t)min(n0 + q1 / 6, g.N - 1) * wrow + (q1 % 6) * 16;
  const char* wp2 = Ws + (size_t)min(n0 + q2 / 6, g.N - 1) * wrow + (q2 % 6) * 16;
#define GS_LOAD(kt)                                                                                   \
  do {                                                                                                \
    const int k0_ = (kt) * BK;                                                                        \
    const bool first_ = k0_ < g.K0;                                                                   \
    const float* ab_ = (first_ ? A0 : A1) + (first_ ? k0_ : k0_ - g.K0) + s_c4;                       \
    const size_t ld_ = first_ ? g.lda0 : g.lda1;                                                      \
    areg0 = *reinterpret_cast<const float4*>(ab_ + ar0 * ld_);                                        \
    areg1 = *reinterpret_cast<const float4*>(ab_ + ar1 * ld_);                                        \
    wreg0 = *reinterpret_cast<const float4*>(wp0 + (size_t)(kt) * 96);                                \
    wreg1 = *reinterpret_cast<const float4*>(wp1 + (size_t)(kt) * 96);                                \
    wreg2 = *reinterpret_cast<const float4*>(wp2 + (size_t)(kt) * 96);                                \
  } while (0)
#define GS_SPLIT_STORE(v_, dst_)                                                                      \
  do {                                                                                                \
    __bf16 h0_, m0_, l0_, h1_, m1_, l1_, h2_, m2_, l2_, h3_, m3_, l3_;                                \
    gsplit3((v_).x, h0_, m0_, l0_); gsplit3((v_).y, h1_, m1_, l1_);                                   \
    gsplit3((v_).z, h2_, m2_, l2_); gsplit3((v_).w, h3_, m3_, l3_);                                   \
    const gbf16x4 hi_ = {h0_, h1_, h2_, h3_}, mid_ = {m0_, m1_, m2_, m3_}, lo_ = {l0_, l1_, l2_, l3_}; \
    *reinterpret_cast<gbf16x4*>(dst_) = hi_;                                                          \
    *reinterpret_cast<gbf16x4*>((dst_) + 32) = mid_;                                                  \
    *reinterpret_cast<gbf16x4*>((dst_) + 64) = lo_;                                                   \
  } while (0)
#define GS_STORE(buf_)                                                                                \
  do {                                                                                                \
    char* as_ = sm + (buf_) * TILEB;                                                                  \
    char* bs_ = as_ + BM * GS_ROW;                                                                    \
    GS_SPLIT_STORE(areg0, as_ + s_r * GS_ROW + s_c4 * 2);                                             \
    GS_SPLIT_STORE(areg1, as_ + (s_r + 64) * GS_ROW + s_c4 * 2);                                      \
    *reinterpret_cast<float4*>(bs_ + (q0 / 6) * GS_ROW + (q0 % 6) * 16) = wreg0;                      \
    *reinterpret_cast<float4*>(bs_ + (q1 / 6) * GS_ROW + (q1 % 6) * 16) = wreg1;                      \
    *reinterpret_cast<float4*>(bs_ + (q2 / 6) * GS_ROW + (q2 % 6) * 16) = wreg2;                      \
  } while (0)

  f32x16 acc[MT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < MT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const int a_off = (wm * 64 + l31) * GS_ROW + h * 16;
  const int b_off = BM * GS_ROW + (wn * 64 + l31) * GS_ROW + h * 16;

  GS_LOAD(0);
  GS_STORE(0);
  if (ktiles > 1) GS_LOAD(1);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + 1 < ktiles) {
      GS_STORE((kt + 1) & 1);
      if (kt + 2 < ktiles) GS_LOAD(kt + 2);
    }
    const char* ap = sm + (kt & 1) * TILEB + a_off;
    const char* bp = sm + (kt & 1) * TILEB + b_off;
    gbf16x8 af[MT][3], bf[MT][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        af[mt][pl] = *reinterpret_cast<const gbf16x8*>(ap + mt * 32 * GS_ROW + 32 * pl);
        bf[mt][pl] = *reinterpret_cast<const gbf16x8*>(bp + mt * 32 * GS_ROW + 32 * pl);
      }
#define GS_MM(pa_, pb_)                                                                               \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                   \
  _Pragma("unroll") for (int nt = 0; nt < MT; ++nt)                                                   \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][pa_], bf[nt][pb_], acc[mt][nt], 0, 0, 0);
    GS_MM(2, 0) GS_MM(0, 2) GS_MM(1, 1) GS_MM(1, 0) GS_MM(0, 1) GS_MM(0, 0)
#undef GS_MM
    __syncthreads();
  }
#undef GS_LOAD
#undef GS_STORE
#undef GS_SPLIT_STORE
  gemm_epilogue<NW, MT>(g, acc, smem, Y, m0, n0, wm, wn, lane, wave);
}

extern "C" int gfc_pack_linear_split(const float* W, int ldw, void* w_split, int N, int K, void* stream) {
  if (!W || !w_split || N <= 0 || K <= 0 || K % 16 || ldw < K) return GFC_ERR_INVALID;
  const long long total = (long long)N * K;
  hipLaunchKernelGGL(pack_linear_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     W, ldw, (__bf16*)w_split, N, K);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_linear_split(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1, const void* w_split,
                                const float* bias, const float* scale, const float* shift, float alpha,
                                const float* residual, const float* rot_cos, const float* rot_sin, int rot_cols,
                                float* Y, int ldy, int M, int N, void* stream) {
  if (!A0 || !w_split || !Y || M <= 0 || N <= 0 || K0 <= 0 || K0 % GBK || K1 % GBK || K1 < 0) return GFC_ERR_INVALID;
  if ((K1 > 0) != (A1 != nullptr)) return GFC_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return GFC_ERR_INVALID;
  if ((rot_cos == nullptr) != (rot_sin == nullptr)) return GFC_ERR_INVALID;
  if (rot_cos && (rot_cols % 64 != 0)) return GFC_ERR_INVALID;
  if (lda0 % 4 || (A1 && lda1 % 4)) return GFC_ERR_INVALID;
  GemmArgs g = {};
  g.A0 = A0; g.A1 = A1; g.W = (const float*)w_split; g.bias = bias; g.scale = scale; g.shift = shift;
  g.residual = residual; g.rot_cos = rot_cos; g.rot_sin = rot_sin; g.Y = Y;
  g.lda0 = lda0; g.lda1 = lda1; g.ldw = 0; g.ldy = ldy;
  g.K0 = K0; g.K1 = K1; g.M = M; g.N = N; g.rot_cols = rot_cols; g.alpha = alpha;
  constexpr size_t kloop = (size_t)2 * 256 * GS_ROW, patches = (size_t)2 * 2 * 32 * (64 + 4) * sizeof(float);
  const size_t lds = kloop > patches ? kloop : patches;
  dim3 grid((N + 127) / 128, (M + 127) / 128, 1);
  hipLaunchKernelGGL(gemm_nt_split_kernel, grid, dim3(256), lds, (hipStream_t)stream, g);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

