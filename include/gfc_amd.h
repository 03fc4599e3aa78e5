/* gfc_amd.h -- C ABI of libgfc_amd.so: SuperPoint + LightGlue hot path on MI355X (gfx950).
 *
 * The reference (ipastore/glue-factory-colon) is pure Python on PyTorch: it has no FFI
 * for this path.  The boundary it offers is the model registry
 * (gluefactory/models/__init__.py:7-30 `get_model`) and the module contract
 * `BaseModel.forward(data: dict) -> dict` (gluefactory/models/base_model.py:101-113).
 * Each entry point below replaces the ATen op sequence of one reference function; the
 * reference file:line it replaces is cited on every declaration.  The Python binding a
 * maintainer adds is shown in INTEGRATION.md (ctypes, no torch types cross this line).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc'ed or torch-allocated) unless marked
 *     "host"; fp32 everywhere, int32 for counts/tables, int64 for match indices
 *     (the reference returns torch.long, lightglue.py:294-319);
 *   - no allocation and no synchronisation inside: the caller passes a workspace of at
 *     least gfc_*_workspace_bytes() and a hipStream_t (as void*); every kernel is
 *     enqueued on that stream and the call returns immediately;
 *   - no MUTABLE global state: calls are re-entrant per stream and from any host thread.
 *     What the library keeps process-wide is read-only after its first use and never
 *     changes a result: the tuning knobs (GFC_GEMM_TILE, GFC_ATTN_CFG, ... read ONCE from
 *     the environment by the first call, csrc/runtime.h -- set them before that call;
 *     they select between kernel variants held to the same parity tests) and per-device
 *     facts (CU count, the dynamic-LDS attribute of each kernel instantiation);
 *   - return value: 0 = ok, otherwise a gfc_status.  Nothing is written on error.
 *   - activations inside the extractor are NHWC ("channels last").
 */
#ifndef GFC_AMD_H
#define GFC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GFC_OK = 0,
  GFC_ERR_INVALID = 1,     /* bad shape / null pointer / unsupported combination of arguments */
  GFC_ERR_WORKSPACE = 2,   /* workspace smaller than gfc_*_workspace_bytes() */
  GFC_ERR_UNSUPPORTED = 3, /* valid for the reference, not built here (e.g. nms_radius > 4) */
  GFC_ERR_LAUNCH = 4       /* hipGetLastError() != hipSuccess after a launch */
} gfc_status;

/* Library / device identification ("gfx950" builds only). */
const char* gfc_version(void);

/* ------------------------------------------------------------------------------------
 * Primitives (also exported so that parity tests can pin each kernel in isolation)
 * ---------------------------------------------------------------------------------- */

/* Re-pack a conv weight from the state-dict layout [Cout][Cin][3][3] to the kernel
 * layout [tap=ky*3+kx][Cout][Cin].  (nn.Conv2d weights, superpoint_open.py:64-66) */
int gfc_pack_conv3x3(const float* w_oihw, float* w_packed, int cout, int cin, void* stream);

/* 3x3 convolution, stride 1, zero padding 1, NHWC, fused epilogue
 *   y = conv(x) + bias;  if relu: y = max(y,0);  if scale: y = y*scale + shift;
 *   if pool: 2x2/2 max-pool of y (floor semantics).
 * Replaces VGGBlock (superpoint_open.py:61-77: conv -> ReLU -> BatchNorm(eval)) followed by
 * nn.MaxPool2d(2,2) (superpoint_open.py:104-106), and conv+ReLU(+pool) of
 * gluefactory_nonfree/superpoint.py:214-224.  cin % 32 == 0 (or cin == 1), cout % 64 == 0.
 * x [B,H,W,cin], w packed [9][cout][cin] (cin == 1: [9][cout]), y [B,Ho,Wo,cout]. */
int gfc_conv3x3(const float* x, const float* w_packed, const float* bias, const float* scale,
                const float* shift, float* y, int B, int H, int W, int cin, int cout, int relu,
                int pool, void* stream);

/* The same layer as gfc_conv3x3 (relu / BN affine / pool epilogue, NHWC) as Winograd F(2x2,3x3) on fp32 MFMA:
 * 16 instead of 36 multiplications per 2x2 output block, fp32 products and accumulation, filter transform done
 * once in float64 by gfc_pack_conv3x3_wino (w_packed: 16*cout*cin floats in MFMA-fragment order).
 * cin % 16 == 0, cout % 64 == 0.  gfc_sp_stem_wino = gfc_sp_stem with conv1b in this form. */
int gfc_pack_conv3x3_wino(const float* w_oihw, float* w_packed, int cout, int cin, void* stream);
int gfc_conv3x3_wino(const float* x, const float* w_wino, const float* bias, const float* scale, const float* shift,
                     float* y, int B, int H, int W, int cin, int cout, int relu, int pool, void* stream);
int gfc_sp_stem_wino(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                     const float* w2_wino, const float* b2, const float* s2, const float* t2, float* y, int B, int H,
                     int W, void* stream);

/* The stem with conv1b as Winograd F(4x4,3x3) (36 instead of 64 products per 4x4 output block and input channel)
 * and conv1a evaluated on the matrix pipe; same function and arguments as gfc_sp_stem_wino except for the filter
 * packing (gfc_pack_conv3x3_wino43: [64][64][3][3] -> 36*64*64 floats in MFMA-fragment order, transform in float64).
 * superpoint_open.py:61-77,100-108; superpoint.py:214-218. */
int gfc_pack_conv3x3_wino43(const float* w_oihw, float* w_packed, int cout, int cin, void* stream);
int gfc_sp_stem_wino43(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                       const float* w2_wino43, const float* b2, const float* s2, const float* t2, float* y, int B, int H,
                       int W, void* stream);

/* Extractor stem: conv1a (1 -> 64) + conv1b (64 -> 64) + 2x2 max-pool in ONE launch.  The first layer is
 * recomputed per workgroup on the 18x18 halo of its 16x16 tile from a 20x20 image patch in LDS, so its
 * [B,H,W,64] output (the largest activation of the network) never exists in HBM.
 * image [B,H,W] (one channel), w1 [9][64], w2 packed [9][64][64], y [B,H/2,W/2,64].
 * Replaces backbone.0 of superpoint_open.py:100-106 / conv1a, conv1b, pool of superpoint.py:214-216. */
int gfc_sp_stem(const float* image, const float* w1, const float* b1, const float* s1, const float* t1,
                const float* w2_packed, const float* b2, const float* s2, const float* t2, float* y, int B, int H,
                int W, void* stream);

/* y[M,N] = epilogue( [A0 | A1][M,K0+K1] * W[N,K0+K1]^T + bias ).  Row-major, leading
 * dimensions in floats.  Replaces nn.Linear / 1x1 conv (F.linear -> addmm) at
 * lightglue.py:139-148,158,163-164,181-189 and superpoint_open.py:112-118.
 *   A1 may be NULL (K1 = 0): the two-source form implements torch.cat([x, msg], -1) of
 *   lightglue.py:164,221-222 without materialising the concatenation.
 *   scale/shift (per column, nullable): y = y*scale + shift  (BatchNorm of the 1x1 heads)
 *   alpha: y *= alpha (final_proj / d^(1/4), lightglue.py:281-284)
 *   residual (nullable, ld = ldy): y += residual (lightglue.py:164)
 *   rot_cos/rot_sin (nullable, [M,64]): rotary embedding applied to columns < rot_cols,
 *     pairing adjacent columns (lightglue.py:43-50,160-161); columns are head-major, 64 wide.
 * K0, K1 multiples of 32. */
int gfc_linear(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1, const float* W,
               int ldw, const float* bias, const float* scale, const float* shift, float alpha,
               const float* residual, const float* rot_cos, const float* rot_sin, int rot_cols,
               float* Y, int ldy, int M, int N, void* stream);

/* Batched similarity: for z < batch: Y_z[M,N] = A_z[M,K] * B_z[N,K]^T (einsum "bmd,bnd->bmn",
 * lightglue.py:285).  Strides between batch entries in floats. */
int gfc_batched_nt(const float* A, int lda, long long strideA, const float* Bm, int ldb,
                   long long strideB, float* Y, int ldy, long long strideY, int M, int N, int K,
                   int batch, void* stream);

/* Multi-head attention over packed rows, head_dim 64, fp32 MFMA, never materialising the
 * score matrix.  problems[p] = {q_row0, n_q, kv_row0, n_kv} (int32 x4, device).  For every
 * problem and head h:  O[q_row0+i, 64h:64h+64] = softmax_j(Q_i . K_j * scale) V_j.
 * Replaces F.scaled_dot_product_attention (lightglue.py:119-122) and, called with the two
 * directions as two problems, the bidirectional cross attention of lightglue.py:207-217. */
int gfc_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                  int ldo, const int32_t* problems, int n_problems, int max_nq, int heads, float scale,
                  void* ws, size_t ws_bytes, void* stream);
/* Optional scratch (ws may be NULL): small problem sets (batch 1..2) share each query block's keys out over
 * several workgroups and merge the partial soft-maxes; 0 when the set is large enough not to need it. */
size_t gfc_attention_workspace_bytes(int n_problems, int max_nq, int heads);

/* In-place LayerNorm(eps 1e-5, affine) + exact (erf) GELU over rows of width 512.
 * Replaces ffn[1], ffn[2] (lightglue.py:143-148). */
int gfc_layernorm_gelu(float* x, int ld, int rows, int width, const float* gamma, const float* beta,
                       void* stream);

/* The WHOLE LightGlue FFN with its residual in one kernel (lightglue.py:143-148, called at :162-164,219-222):
 *   Y[M,256] = residual + ( GELU_erf( LayerNorm_512( [A0 | A1] * W0[512,K0+K1]^T + b0 ; gamma, beta, eps 1e-5 ) )
 *                           * W3[256,512]^T + b3 ).
 * Row-owning 128-row workgroup tiles: the row statistics, the [M,512] pre-activation AND the activated hidden tile stay
 * on chip (the hidden tile goes from the first GEMM's accumulators through LDS into the second GEMM).  residual
 * (nullable) and Y share the row stride ldy and may alias (in-place residual update); b3 nullable.  Bit-identical to
 * gfc_linear_layernorm_gelu followed by gfc_linear(..., residual). */
int gfc_ffn_fused(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1, const float* W0, int ldw0,
                  const float* b0, const float* gamma, const float* beta, const float* W3, int ldw3, const float* b3,
                  const float* residual, float* Y, int ldy, int M, void* stream);

/* The first two stages of the LightGlue FFN in one kernel (lightglue.py:143-148, called at :164,221-222):
 *   Y[M,512] = GELU_erf( LayerNorm_512( [A0 | A1] * W[512,K0+K1]^T + bias ; gamma, beta, eps 1e-5 ) ).
 * Row-owning workgroup tiles (128 rows x all 512 columns): the row statistics stay on chip and the
 * pre-activation never reaches HBM.  N must be 512; same operand conventions as gfc_linear. */
int gfc_linear_layernorm_gelu(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1, const float* W,
                              int ldw, const float* bias, const float* gamma, const float* beta, float* Y, int ldy,
                              int M, int N, void* stream);

/* ------------------------------------------------------------------------------------
 * SuperPoint extractor
 * ---------------------------------------------------------------------------------- */
typedef struct {
  /* encoder conv1a..conv4b.  w[0]: [9][64] (cin = 1); w[1..7]: packed [9][cout][cin].
   * scale/shift: folded eval-mode BatchNorm (alpha = gamma/sqrt(var+eps), beta - mean*alpha),
   * NULL for the official (no-BN) variant. */
  const float* w[8];
  const float* bias[8];
  const float* scale[8];
  const float* shift[8];
  /* both 3x3 heads merged: 128 -> 512 = [detector hidden | descriptor hidden] */
  const float* wh;
  const float* bias_h;
  const float* scale_h;
  const float* shift_h;
  /* 1x1 detector 256 -> 65 ([65][256]) and 1x1 descriptor 256 -> desc_dim ([desc_dim][256]) */
  const float* wp;
  const float* bias_p;
  const float* scale_p;
  const float* shift_p;
  const float* wd;
  const float* bias_d;
  const float* scale_d;
  const float* shift_d;
  int desc_dim;
  /* conv_mode = 0: direct implicit GEMM on fp32 MFMA (gfc_conv3x3 / gfc_sp_stem).  (1 was the round-1 bf16x3-split
   * arithmetic, retired from the library in round 4: gfc_sp_dense rejects it.) */
  int conv_mode;
  /* conv_mode = 2: Winograd F(2x2,3x3) on the fp32 matrix pipe (gfc_conv3x3_wino / gfc_sp_stem_wino): w_wino[1..7] /
   * wh_wino are the filters transformed (in float64) and packed by gfc_pack_conv3x3_wino; w[0] (conv1a, cin = 1)
   * is still used.  Every product and accumulation is fp32; 2.25x fewer multiplications than conv_mode 0. */
  const float* w_wino[8];
  const float* wh_wino;
  /* conv_mode = 2, optional: conv1b's filters transformed for Winograd F(4x4,3x3) and packed by
   * gfc_pack_conv3x3_wino43.  When set, the stem (conv1a + conv1b + pool) runs as gfc_sp_stem_wino43; NULL (or
   * $GFC_STEM_F43=0): gfc_sp_stem_wino (F(2x2,3x3), w_wino[1]).  Only the stem: for deeper layers F(4x4,3x3) does
   * not hold the 1e-5 heat-map bar (tools/micro/winograd_f43_numerics.py). */
  const float* w_stem_wino43;
} gfc_sp_params;

typedef enum { GFC_SAMPLE_OPEN = 0, GFC_SAMPLE_LEGACY = 1, GFC_SAMPLE_FIXED = 2 } gfc_sample_mode;

/* Optional per-launch timing of a hot kernel (host struct owned by the caller, no global state): the entry point
 * that takes a gfc_trace brackets every launch of ITS traced kernel with hipEventRecord(start[count]) /
 * hipEventRecord(stop[count]) on the call's stream and increments count (while count < capacity):
 *   gfc_sp_dense           -> the stem kernel (conv1a + conv1b + pool, 44 % of the extractor's FLOPs),
 *   gfc_lg_forward_packed  -> the attention launches (2 per layer: self, cross), the largest share of the step.
 * Events come from gfc_event_create().  Used by bench.py for the live roofline figures; NULL in production. */
typedef struct {
  void** start;
  void** stop;
  int capacity;
  int count;
} gfc_trace;

int gfc_event_create(void** event);
int gfc_event_destroy(void* event);
/* milliseconds between two recorded events (both must have completed) */
int gfc_event_elapsed_ms(void* start, void* stop, float* ms);

/* bench-only: what the matrix pipe of THIS device sustains.  Every SIMD of every CU issues `mfmas_per_wave`
 * back-to-back v_mfma_f32_32x32x2_f32 (registers only, 4 accumulators); returns the fp32-MFMA rate over the whole
 * launch and the shader clock it ran at (s_memtime ticks, = shader cycles, per 100 MHz s_memrealtime tick).  The
 * data-sheet peak (157.3 TFLOP/s) assumes 2.4 GHz; under full matrix load the part clocks lower. */
int gfc_probe_mfma_peak(int mfmas_per_wave, float* tflops, float* shader_clock_ghz, void* stream);

size_t gfc_sp_workspace_bytes(int B, int C, int H, int W);

/* image [B,C,H,W] (C = 1 or 3, RGB -> grey fused) -> heat-map [B, 8*(H/8), 8*(W/8)] (softmax over
 * 65 logits, dustbin dropped, depth-to-space) and raw descriptor map [B,H/8,W/8,desc_dim]
 * (NHWC, before L2 normalisation).  superpoint_open.py:128-144; superpoint.py:208-241. */
int gfc_sp_dense(const gfc_sp_params* p, const float* image, int B, int C, int H, int W, float* heatmap,
                 float* desc_raw, void* ws, size_t ws_bytes, gfc_trace* trace, void* stream);

/* Detector head in one launch: 1x1 convolution 256 -> 65 (+ eval-BN affine when scale / shift are given), softmax over the
 * 65 logits of every cell, dustbin dropped, 8x8 depth-to-space:  heat[b, 8y+i, 8x+j] = softmax_c(W . hidden[b,y,x,:] + b)[8i+j].
 * hidden [B*h8*w8][lda] NHWC rows (the detector's 256 hidden channels first, lda >= 256, a multiple of 4); w [65][256];
 * hidden and w 16-byte aligned (GFC_ERR_INVALID otherwise: both are read with 128-bit loads);
 * bias / scale / shift [65] (scale and shift nullable together); heat [B][8 h8][8 w8].  The logits never reach HBM.
 * superpoint_open.py:111-114,138-144; superpoint.py:193-194,229-235 (part of gfc_sp_dense). */
int gfc_sp_detector_head(const float* hidden, int lda, const float* w, const float* bias, const float* scale,
                         const float* shift, int B, int h8, int w8, float* heat, void* stream);

/* Max-pool NMS with two recovery rounds, then outer `border` rows/cols := -1.
 * valid_wh (nullable, int32 [B,2] = (w,h)): right/bottom border measured from the true image
 * extent.  superpoint_open.py:36-51,148-154; superpoint.py:63-83,249-260.  radius <= 4. */
int gfc_sp_nms(const float* heatmap, int B, int H, int W, int radius, int border, const int32_t* valid_wh,
               float* out, void* stream);

size_t gfc_sp_select_workspace_bytes(int B, int H, int W);

/* Per image: candidates = pixels with score > threshold in row-major order; if more than k
 * (k < 0: unlimited) the k best by descending score (ties: lower linear index first), else all
 * of them in row-major order.  kpts [B,cap,2] (x,y as floats), scores [B,cap], counts [B].
 * cap = row capacity of the output arrays (>= k when k >= 0; H*W when unlimited).
 * superpoint_open.py:156-192,54-58; superpoint.py:262-300,86-90. */
int gfc_sp_select(const float* scores, int B, int H, int W, float threshold, int k, int cap, float* kpts,
                  float* kscores, int32_t* counts, void* ws, size_t ws_bytes, void* stream);

/* gfc_sp_nms + gfc_sp_select in one pass over the heat-map: the NMS kernel appends every pixel above the
 * threshold to the per-image key list (unordered; keys are unique, so the result is still deterministic) and the
 * selection kernel skips its scan.  nms_out (nullable) receives the suppressed map.
 * 1 <= k <= 8192, 1 <= radius <= 4 (radius 0 keeps ~all pixels: use the two separate stages). */
size_t gfc_sp_nms_select_workspace_bytes(int B, int H, int W);
int gfc_sp_nms_select(const float* heatmap, int B, int H, int W, int radius, int border, const int32_t* valid_wh,
                      float threshold, int k, int cap, float* nms_out, float* kpts, float* kscores, int32_t* counts,
                      void* ws, size_t ws_bytes, void* stream);

/* Bilinear sampling of L2-normalised dense descriptors at keypoints + L2 normalisation.
 * desc_raw [B,h,w,D] un-normalised (normalisation over D is applied per corner on the fly),
 * kpts [B,cap,2] integer-valued pixel coordinates, n_kpts [B] (nullable = cap for all),
 * out [B,cap,D].  If kpts_out != NULL it receives kpts + 0.5 (may alias kpts).
 * superpoint_open.py:22-33,133-135,221; superpoint.py:120-152. */
int gfc_sp_sample(const float* desc_raw, int B, int h, int w, int D, const float* kpts, const int32_t* n_kpts,
                  int cap, int mode, float* out, float* kpts_out, void* stream);

/* pad_and_stack(mode="random_c") for key points + zeros for their scores, in place, one launch, no host sync
 * (gluefactory/models/utils/misc.py:19-62,103-113; force_num_keypoints at superpoint_open.py:193-219,
 * superpoint.py:330-365, disk_kornia.py:109-124): slots in [counts[b], k) get per-coordinate uniform samples in
 * [min, max] of the image's own key points ([low, high] when it has none; high = min over `sizes` if given, else
 * high_fallback) from a counter-based generator keyed by (seed, image, slot) -- random padding, as in the reference. */
int gfc_sp_pad_keypoints(float* kpts, float* kscores, const int32_t* counts, int B, int cap, int k, float low,
                         const float* sizes, int n_sizes, float high_fallback, unsigned int seed, void* stream);

/* L2-normalise rows in place (dense_outputs: F.normalize over channels, superpoint_open.py:133-135). */
int gfc_l2norm_rows(float* x, long long rows, int width, void* stream);

/* ------------------------------------------------------------------------------------
 * DISK extractor: the stages behind the network (disk_kornia.py:42-47,129-137; kornia's
 * heatmap_to_keypoints / nms / merge_with_descriptors restated: parity for them is unpinned, kornia is absent)
 * ---------------------------------------------------------------------------------- */
/* heat-map [B,H,W] -> key points.  A pixel survives iff it is the first maximum (row-major window scan, as
 * F.max_pool2d(return_indices=True) reports it) of the window x window neighbourhood centred on it and > cutoff.
 * n >= 0: keep the scores strictly above the (n+1)-th largest survivor (min(n+1, count)-th: with count <= n the
 * minimum is dropped), first n in row-major order; n < 0: all survivors.  kpts [B,cap,2] (x,y as floats, integer
 * valued, NOT sorted by score), kscores [B,cap], counts [B].  cap >= n (n >= 0) or >= H*W.  window odd, <= 9. */
size_t gfc_disk_select_workspace_bytes(int B, int H, int W);
int gfc_disk_nms_select(const float* heatmap, int B, int H, int W, int window, float cutoff, int n, int cap,
                        float* kpts, float* kscores, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
/* dense descriptors [B,D,H,W] (NCHW) read at the integer key-point pixels and L2-normalised over D
 * (F.normalize, eps 1e-12) -> out [B,cap,D]; slots >= counts[b] (counts nullable) are zero-filled. */
int gfc_disk_gather_descriptors(const float* dense_nchw, int B, int D, int H, int W, const float* kpts,
                                const int32_t* counts, int cap, float* out, void* stream);
/* the same from an NHWC array [B,H,W,D] (what gfc_disk_conv5x5 writes: one contiguous row per key point) */
int gfc_disk_gather_descriptors_nhwc(const float* dense_nhwc, int B, int D, int H, int W, const float* kpts,
                                     const int32_t* counts, int cap, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * DISK network (disk_kornia.py:24-47 -> kornia.feature.DISK.heatmap_and_dense_descriptors): the thin U-Net
 * Unet(in_features=3, size=5, down=[16,32,64,64,64], up=[64,64,64,desc_dim+1]) of kornia/feature/disk/_unets,
 * restated from kornia's published source (absent offline: NETWORK PARITY UNPINNED).  Activations are NHWC fp32.
 * One "Conv" of that network = [InstanceNorm2d -> PReLU ->] Conv2d(5x5, padding 2, bias): the statistics come from
 * gfc_disk_instnorm_stats, normalisation and gate are applied while gfc_disk_conv5x5 stages its input.
 * ---------------------------------------------------------------------------------- */
/* Conv2d weights OIHW [cout][cin][5][5] -> the layout gfc_disk_conv5x5 reads (cin padded to 16, cout to 32). */
size_t gfc_disk_conv5x5_packed_floats(int cout, int cin);
int gfc_disk_pack_conv5x5(const float* w_oihw, float* w_packed, int cout, int cin, void* stream);
/* y[b,y,x, 0:co_count] (channel stride ldy >= co_count: a slice of a wider, concatenated tensor) =
 *   conv5x5(gate(norm(x)))[b,y,x, co_first : co_first + co_count] + bias  of a layer with cout_total output channels
 * (co_first % 32 == 0; the whole layer: co_first 0, co_count cout_total -- DISK's last layer is run as descriptors
 * [0,128) and heat-map [128,129) into two arrays);  x [B,H,W,cin] contiguous, cin % 4 == 0 (channels beyond cin of the
 * last 16-chunk count as zero); mean/rstd [B,cin] (both or neither; InstanceNorm2d without affine), prelu [cin] or
 * NULL (PReLU slope per channel); zero padding is applied AFTER normalisation and gate, as Conv2d pads its own input.
 * w_packed / bias are the layer's whole arrays (not offset by the caller). */
int gfc_disk_conv5x5(const float* x, const float* mean, const float* rstd, const float* prelu, const float* w_packed,
                     const float* bias, float* y, int ldy, int B, int H, int W, int cin, int cout_total, int co_first,
                     int co_count, void* stream);
/* per (image, channel) mean and 1 / sqrt(biased variance + eps) over H x W of x [B,H,W,C] contiguous
 * (float64 sums, fixed summation order); C in {16, 32, 64, 80, 96, 128} or any C with 480 % (C/4) == 0. */
size_t gfc_disk_instnorm_workspace_bytes(int B, int C);
int gfc_disk_instnorm_stats(const float* x, int B, int H, int W, int C, float eps, float* mean, float* rstd, void* ws,
                            size_t ws_bytes, void* stream);
/* F.avg_pool2d(x, 2): x [B,H,W,C] with channel stride ldx (H, W even) -> y [B,H/2,W/2,C] contiguous */
int gfc_disk_avgpool2(const float* x, int ldx, int B, int H, int W, int C, float* y, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False): x [B,h,w,C] contiguous -> the first C
 * channels of y [B,2h,2w,ldy] */
int gfc_disk_upsample2(const float* x, int B, int h, int w, int C, float* y, int ldy, void* stream);
/* image [B,3,H,W] -> [B,H,W,4] (fourth channel zero): the NHWC input of the first convolution */
int gfc_disk_nchw3_to_nhwc4(const float* image, int B, int H, int W, float* y, void* stream);

/* ------------------------------------------------------------------------------------
 * LightGlue matcher
 * ---------------------------------------------------------------------------------- */
#define GFC_LG_MAX_LAYERS 16
typedef struct {
  int n_layers;    /* 9 */
  int input_dim;   /* 256, or 128 with input_proj */
  const float* input_proj_w; /* [256][input_dim] or NULL */
  const float* input_proj_b;
  const float* posenc_wr;    /* [32][posenc_dim] */
  int posenc_dim;            /* 2, or 4 with add_scale_ori (0 is read as 2) */
  /* self block.  wqkv rows re-ordered to [q(256) | k(256) | v(256)], each head-major:
   * new row s*256 + h*64 + d  <-  state-dict row h*192 + d*3 + s   (lightglue.py:157-159) */
  const float* wqkv[GFC_LG_MAX_LAYERS];
  const float* bqkv[GFC_LG_MAX_LAYERS];
  /* out_proj.  NULL = folded into s_ffn0_w at load time (no GEMM of its own):
   *   ffn0(cat[x, out_proj(ctx)]) = [x | ctx] . [W0a | W0b.Wo]^T + (b0 + W0b.bo)   (lightglue.py:163-164) */
  const float* s_out_w[GFC_LG_MAX_LAYERS];
  const float* s_out_b[GFC_LG_MAX_LAYERS];
  const float* s_ffn0_w[GFC_LG_MAX_LAYERS]; /* [512][512] */
  const float* s_ffn0_b[GFC_LG_MAX_LAYERS];
  const float* s_ln_g[GFC_LG_MAX_LAYERS];
  const float* s_ln_b[GFC_LG_MAX_LAYERS];
  const float* s_ffn3_w[GFC_LG_MAX_LAYERS]; /* [256][512] */
  const float* s_ffn3_b[GFC_LG_MAX_LAYERS];
  /* cross block.  c_qkv_w = [to_qk ; to_v] stacked to [512][256] */
  const float* c_qkv_w[GFC_LG_MAX_LAYERS];
  const float* c_qkv_b[GFC_LG_MAX_LAYERS];
  const float* c_out_w[GFC_LG_MAX_LAYERS]; /* to_out; NULL = folded into c_ffn0_w (lightglue.py:219-222) */
  const float* c_out_b[GFC_LG_MAX_LAYERS];
  const float* c_ffn0_w[GFC_LG_MAX_LAYERS];
  const float* c_ffn0_b[GFC_LG_MAX_LAYERS];
  const float* c_ln_g[GFC_LG_MAX_LAYERS];
  const float* c_ln_b[GFC_LG_MAX_LAYERS];
  const float* c_ffn3_w[GFC_LG_MAX_LAYERS];
  const float* c_ffn3_b[GFC_LG_MAX_LAYERS];
  /* assignment heads log_assignment.{i} (lightglue.py:272-291).  gfc_lg_forward uses layer n_layers-1;
   * the adaptive path (early stop) may read any layer's head, and every layer's matchability for pruning */
  const float* final_proj_w[GFC_LG_MAX_LAYERS]; /* [256][256] */
  const float* final_proj_b[GFC_LG_MAX_LAYERS];
  const float* matchability_w[GFC_LG_MAX_LAYERS]; /* [256] */
  const float* matchability_b[GFC_LG_MAX_LAYERS]; /* [1] */
  /* token_confidence.{i}.token.0 (lightglue.py:69-80), i < n_layers-1 */
  const float* token_w[GFC_LG_MAX_LAYERS]; /* [256] */
  const float* token_b[GFC_LG_MAX_LAYERS]; /* [1] */
} gfc_lg_params;

size_t gfc_lg_workspace_bytes(int B, int M, int N);

/* Rotary tables: kpts [rows,2] pixel coords, per image i: rows [row0[i], row0[i]+n[i]) are
 * normalised with size[i] = (w,h): (k - size/2) / (max(size)/2), projected by Wr [32][dim];
 * cos/sin [rows,64] with each value repeated twice.  lightglue.py:28-40,53-66.
 * dim = 4 (conf.add_scale_ori, lightglue.py:359,436-453): scale_ori [rows,2] = (scale, orientation) is appended to
 * the normalised key point before the projection; dim = 2: scale_ori must be NULL. */
int gfc_lg_posenc(const float* kpts, const float* scale_ori, const float* sizes, const int32_t* row0, const int32_t* n,
                  int n_images, int max_n, const float* wr, int dim, float* cos_out, float* sin_out, void* stream);

/* log assignment [B,M+1,N+1] from sim [B,M,N] (ld = N), z0 [B,M], z1 [B,N]:
 * sigmoid_log_double_softmax, lightglue.py:257-269.  ws: 2*B*(M+N) floats. */
int gfc_lg_log_assignment(const float* sim, const float* z0, const float* z1, int B, int M, int N, float* out,
                          void* ws, size_t ws_bytes, void* stream);

/* Mutual arg-max matches from a log assignment [B,M+1,N+1]: filter_matches, lightglue.py:294-319.
 * m0 [B,M] int64, m1 [B,N] int64, ms0 [B,M], ms1 [B,N].  Ties: first index.
 * ws: B*(M+N)*(4+4) bytes. */
int gfc_lg_filter_matches(const float* scores, int B, int M, int N, float threshold, int64_t* m0, int64_t* m1,
                          float* ms0, float* ms1, void* ws, size_t ws_bytes, void* stream);

/* One transformer layer (self block on every image + bidirectional cross block, lightglue.py:231-245) on the
 * packed descriptor rows x [rows,256] (updated in place), with rotary tables cos/sin [rows,64] and attention
 * problem tables {q_row0, n_q, kv_row0, n_kv} (int32 x4 per entry, n_problems entries each; device).
 * Building block of gfc_lg_forward; driven layer by layer by the host for adaptive depth / width
 * (early stop and point pruning, lightglue.py:500-521), where rows are re-packed between layers. */
size_t gfc_lg_layer_workspace_bytes(int rows);
int gfc_lg_layer(const gfc_lg_params* p, int layer, float* x, const float* cos_tab, const float* sin_tab, int rows,
                 const int32_t* self_problems, const int32_t* cross_problems, int n_problems, int max_n, void* ws,
                 size_t ws_bytes, void* stream);

/* out[row] = x[row,:256] . w + b, optionally through a sigmoid: TokenConfidence (lightglue.py:69-80) and
 * MatchAssignment.get_matchability (lightglue.py:290-291). */
int gfc_lg_rowdot(const float* x, int ld, int rows, const float* w, const float* b, int apply_sigmoid, float* out,
                  void* stream);

/* MatchAssignment of layer `layer` + filter_matches on x0 [B*M,256], x1 [B*N,256]
 * (lightglue.py:279-288,294-319); outputs as gfc_lg_forward. */
size_t gfc_lg_assign_workspace_bytes(int B, int M, int N);
int gfc_lg_assign(const gfc_lg_params* p, int layer, const float* x0, const float* x1, int B, int M, int N,
                  float threshold, int64_t* m0, int64_t* m1, float* ms0, float* ms1, float* log_assignment, void* ws,
                  size_t ws_bytes, void* stream);

/* Whole matcher: LightGlue.forward (lightglue.py:422-553) with early stop / pruning disabled.
 * kpts0 [B,M,2], kpts1 [B,N,2] (pixel coords), desc0 [B,M,Din], desc1 [B,N,Din],
 * size0/size1 [B,2] = (w,h) floats; scale_ori0 [B,M,2] / scale_ori1 [B,N,2] = (scales, oris) of the key points when
 * p->posenc_dim == 4 (add_scale_ori), NULL otherwise.  Outputs as filter_matches + log_assignment [B,M+1,N+1]
 * + ref_desc0 [B,M,256], ref_desc1 [B,N,256] (last-layer descriptors, lightglue.py:495-498). */
int gfc_lg_forward(const gfc_lg_params* p, const float* kpts0, const float* kpts1, const float* desc0,
                   const float* desc1, const float* size0, const float* size1, const float* scale_ori0,
                   const float* scale_ori1, int B, int M, int N,
                   float threshold, int64_t* m0, int64_t* m1, float* ms0, float* ms1, float* log_assignment,
                   float* ref_desc0, float* ref_desc1, void* ws, size_t ws_bytes, void* stream);

/* The same matcher on PACKED rows, without a single device-to-device copy: kpts [B*M + B*N, 2] and desc
 * [B*M + B*N, Din] hold the rows of side 0 (pair-major) followed by the rows of side 1 -- the layout one extractor
 * call over both views' images produces; scale_ori [B*M + B*N, 2] or NULL likewise.  `rows` [B*M + B*N, 256] is the
 * caller's row buffer: layer 0 reads the descriptors where they are (desc is never written), every later layer
 * works in place on `rows`, which ends up holding the last layer's descriptors = ref_descriptors0 (first B*M rows)
 * and ref_descriptors1.  attention_trace: optional event pairs around the attention launches (see gfc_trace).
 * Replaces the same reference lines as gfc_lg_forward (lightglue.py:422-553). */
size_t gfc_lg_packed_workspace_bytes(int B, int M, int N);
int gfc_lg_forward_packed(const gfc_lg_params* p, const float* kpts, const float* desc, const float* size0,
                          const float* size1, const float* scale_ori, int B, int M, int N, float threshold, int64_t* m0,
                          int64_t* m1, float* ms0, float* ms1, float* log_assignment, float* rows, void* ws,
                          size_t ws_bytes, gfc_trace* attention_trace, void* stream);

/* The same matcher over B pairs with THEIR OWN key-point counts (m[i], n[i] > 0, host arrays, B <=
 * GFC_LG_MAX_RAGGED_PAIRS): the regime of the reference's evaluation loop, which calls the matcher once per pair only
 * because the IMAGES of an HPatches-style list differ in size (gluefactory/utils/export_predictions.py:36-45,
 * datasets/hpatches.py:60) -- LightGlue itself is image-size independent once the key points exist.  One launch
 * sequence for all pairs: the layers run over all rows at once (the attention kernel takes a per-problem table
 * anyway), the assignment head once per GROUP = maximal run of consecutive pairs with equal (m, n).
 * Row layout of kpts [R,2] / desc [R,Din] / scale_ori [R,2] / rows [R,256], R = sum(m) + sum(n): group after group,
 * inside a group the side-0 rows of its pairs (pair-major) followed by their side-1 rows -- i.e. every group is laid out
 * as gfc_lg_forward_packed lays out a uniform batch (B equal pairs = one group = exactly that function's layout).
 * Outputs are flat, in pair order: m0 / ms0 [sum m], m1 / ms1 [sum n], log_assignment [sum (m+1)(n+1)] (pair i's
 * [m+1, n+1] matrix contiguous).  size0 / size1 [B,2] (device).  No allocation, no synchronisation; the counts travel
 * to the device as kernel arguments.  Same arithmetic per pair as gfc_lg_forward_packed (lightglue.py:422-553). */
#define GFC_LG_MAX_RAGGED_PAIRS 128
size_t gfc_lg_ragged_workspace_bytes(int B, const int32_t* m, const int32_t* n);
int gfc_lg_forward_ragged(const gfc_lg_params* p, const float* kpts, const float* desc, const float* size0,
                          const float* size1, const float* scale_ori, int B, const int32_t* m, const int32_t* n,
                          float threshold, int64_t* m0, int64_t* m1, float* ms0, float* ms1, float* log_assignment,
                          float* rows, void* ws, size_t ws_bytes, gfc_trace* attention_trace, void* stream);

/* Nearest-neighbour matcher ("next" row; the matcher of the reference's superpoint+NN configurations):
 * sim = desc0 . desc1^T, top-2 per row / column, ratio test d1 <= ratio^2 d2 and distance test d1 <= th^2 on
 * d = 2(1 - sim) (thresholds <= 0 disable a test), optional mutual check; matching scores are 0/1;
 * log_assignment (nullable) [B,M+1,N+1] = log_softmax rows + log_softmax columns, zero border.
 * Replaces NearestNeighborMatcher._forward (gluefactory/models/matchers/nearest_neighbor_matcher.py:15-79). */
size_t gfc_nn_workspace_bytes(int B, int M, int N);
int gfc_nn_match(const float* desc0, const float* desc1, int B, int M, int N, int D, float ratio_thresh,
                 float distance_thresh, int mutual, int64_t* m0, int64_t* m1, float* ms0, float* ms1, float* sim,
                 float* log_assignment, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Evaluation ("next" row: the caller right after the matcher)
 * ---------------------------------------------------------------------------------- */

/* Match metrics against a ground-truth homography, one workgroup per pair, M x N distance matrix never
 * materialised.  kp0 [B,M,2], kp1 [B,N,2] (pixels), m0 [B,M] int64 (-1 = unmatched), H / Hinv [B,3,3]
 * (H_0to1 and its inverse, row-major).  out [B,6] = prec@1px, prec@3px, num_matches, num_keypoints,
 * gt_match_recall@pos_th, gt_match_precision@pos_th;  gt_m0_out (nullable) [B,M] int64 ground-truth
 * matches (-1 unmatched, -2 ignore).  Replaces eval_matches_homography (gluefactory/eval/utils.py:141-185),
 * sym_homography_error (geometry/homography.py:314-323), gt_matches_from_homography
 * (geometry/gt_generation.py:730-801; the evaluation calls it with pos_th = neg_th = 3). */
int gfc_eval_matches_homography(const float* kp0, const float* kp1, const int64_t* m0, const float* H,
                                const float* Hinv, int B, int M, int N, float pos_th, float neg_th, float* out,
                                int64_t* gt_m0_out, void* stream);

/* Soft-argmax refinement of selected key points (official variant, `refinement_radius` > 0): kpts [B,cap,2] (x, y,
 * integer valued, the first counts[b] rows of image b; counts nullable = all cap) move by the score-weighted mean
 * offset inside the (2*radius+1)^2 window of the dense heat-map [B,H,W] (zero outside).  Replaces
 * soft_argmax_refinement (gluefactory_nonfree/superpoint.py:100-116, called at :302-305). */
int gfc_sp_refine_keypoints(const float* heatmap, int B, int H, int W, float* kpts, const int32_t* counts, int cap,
                            int radius, void* stream);

/* Specular-mask filtering of key points (reference gluefactory/models/extractors/utils.py:4-42; mask [B,Hm,Wm] bytes,
 * non-zero = keep; image_wh nullable [B,2] int32 = (w, h): the mask is cropped to it, key points beyond are dropped).
 * gfc_sp_mask_scores: the open variant's order (superpoint_open.py:177-188, filter BEFORE top-k): every pixel of the
 * suppressed score map [B,H,W] outside the mask becomes -inf, then gfc_sp_select runs on it.
 * gfc_sp_filter_keypoints: the official variant's order (gluefactory_nonfree/superpoint.py:310-328, filter AFTER
 * top-k): stable in-place compaction of kpts [B,cap,2] (x, y before the +0.5 offset) / kscores [B,cap]; counts [B]
 * is read and updated.  A key point survives when floor/ceil(kp - keypoint_offset) are inside and the mask is set on
 * all four pixels (the reference's callers pass keypoint_offset = 0). */
int gfc_sp_mask_scores(float* scores, int B, int H, int W, const uint8_t* mask, int Hm, int Wm, const int32_t* image_wh,
                       void* stream);
int gfc_sp_filter_keypoints(float* kpts, float* kscores, int32_t* counts, int B, int cap, const uint8_t* mask, int Hm,
                            int Wm, const int32_t* image_wh, float keypoint_offset, void* stream);

/* Weighted DLT homography from the predicted matches and its corner error ("next" row rank 3).  kp0 [B,M,2],
 * kp1 [B,N,2], m0 [B,M] int64 (-1 = unmatched), scores0 [B,M] (the weights), H_gt [B,9] row-major, image_size0 [B,2]
 * = (w, h) of view 0.  H_out [B,9]: normalised-DLT estimate divided by (H[2][2] + 1e-8), all +inf when a pair has
 * fewer than 4 matches or the estimate is not finite; err_out [B]: mean distance of the 4 warped image corners to
 * their ground-truth positions (+inf likewise).  Replaces eval_homography_dlt (gluefactory/eval/utils.py:276-302):
 * kornia's find_homography_dlt (third-party, unpinned) + homography_corner_error (geometry/homography.py:336-342). */
int gfc_eval_homography_dlt(const float* kp0, const float* kp1, const int64_t* m0, const float* scores0,
                            const float* H_gt, const float* image_size0, int B, int M, int N, float* H_out,
                            float* err_out, void* stream);

/* Image preprocessing ("next" row rank 1): [uint8 -> float /255 ->] antialiased bilinear resize, fused.
 * src: src_is_u8_hwc != 0: B interleaved HxWxC byte images (bgr != 0 reverses the channel order, as read_image does
 * after cv2.imread, utils/image.py:135-145), converted like numpy_image_to_torch (image.py:148-156); else B planar
 * CxHxW fp32 images.  dst [B,C,OH,OW] fp32.  antialias != 0 and a down-scaling axis: Gaussian blur with
 * sigma = max((in/out - 1)/2, 0.001) per axis, kernel size int(max(4 sigma, 3)) made odd, reflect border, then
 * bilinear interpolation with torch's align_corners=False/True source-index rule.  Replaces the
 * kornia.geometry.transform.resize call of ImagePreprocessor.__call__ (gluefactory/utils/image.py:33-47; kornia is
 * third-party and unpinned).  GFC_ERR_UNSUPPORTED for blur kernels wider than 63 taps (down-scaling > ~30x). */
int gfc_preprocess_resize(const void* src, int src_is_u8_hwc, int bgr, int B, int C, int H, int W, float* dst, int OH,
                          int OW, int align_corners, int antialias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GFC_AMD_H */
