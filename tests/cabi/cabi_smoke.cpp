// Pure C-ABI consumer of libgfc_amd.so: no Python, no torch -- HIP runtime + include/gfc_amd.h only.
// Runs the extractor stages on a synthetic image with weights read from a flat binary blob written by the
// test (tests/test_gpu_cabi.py) and prints the key points, so that the test can compare them with the Python
// boundary module.  Build: hipcc --offload-arch=gfx950 cabi_smoke.cpp -I include -L <pkg> -lgfc_amd
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gfc_amd.h"

#define HIP_OK(x)                                                     \
  do {                                                                \
    if ((x) != hipSuccess) { fprintf(stderr, "HIP error line %d\n", __LINE__); return 2; } \
  } while (0)
#define GFC_OK_(x)                                                        \
  do {                                                                    \
    int s_ = (x);                                                         \
    if (s_ != GFC_OK) { fprintf(stderr, "gfc status %d line %d\n", s_, __LINE__); return 3; } \
  } while (0)

static float* upload(const std::vector<float>& v) {
  float* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: cabi_smoke blob.bin K\n"); return 1; }
  const int K = atoi(argv[2]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  // blob: int32 H, W; then float tensors in a fixed order (see the test): image, then for each of 8 encoder layers
  // w_oihw, bias, scale, shift; heads 3x3 (merged 512): w, b, scale, shift; pb: w[65*256], b, scale, shift; db: ...
  int hw[2];
  if (fread(hw, 4, 2, f) != 2) return 1;
  const int H = hw[0], W = hw[1];
  auto rd = [&](size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, f) != n) v.clear(); return v; };
  std::vector<float> image = rd((size_t)H * W);
  const int cin[8] = {1, 64, 64, 64, 64, 128, 128, 128}, cout[8] = {64, 64, 64, 64, 128, 128, 128, 128};
  gfc_sp_params p = {};
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  for (int i = 0; i < 8; ++i) {
    std::vector<float> w = rd((size_t)cout[i] * cin[i] * 9), b = rd(cout[i]), sc = rd(cout[i]), sh = rd(cout[i]);
    if (w.empty() || sh.empty()) return 1;
    float* dw = upload(w);
    float* dp = nullptr;
    HIP_OK(hipMalloc(&dp, w.size() * 4));
    GFC_OK_(gfc_pack_conv3x3(dw, dp, cout[i], cin[i], st));
    p.w[i] = dp; p.bias[i] = upload(b); p.scale[i] = upload(sc); p.shift[i] = upload(sh);
    if (i >= 1) {  // the default arithmetic of the boundary modules: Winograd F(2x2,3x3) filters, packed by the library
      float* dq = nullptr;
      HIP_OK(hipMalloc(&dq, (size_t)16 * cout[i] * cin[i] * 4));
      GFC_OK_(gfc_pack_conv3x3_wino(dw, dq, cout[i], cin[i], st));
      p.w_wino[i] = dq;
    }
    if (i == 1) {  // the stem's conv1b additionally as Winograd F(4x4,3x3): what the boundary modules run by default
      float* d43 = nullptr;
      HIP_OK(hipMalloc(&d43, (size_t)36 * 64 * 64 * 4));
      GFC_OK_(gfc_pack_conv3x3_wino43(dw, d43, 64, 64, st));
      p.w_stem_wino43 = d43;
    }
  }
  {
    std::vector<float> w = rd((size_t)512 * 128 * 9), b = rd(512), sc = rd(512), sh = rd(512);
    float* dw = upload(w);
    float* dp = nullptr;
    HIP_OK(hipMalloc(&dp, w.size() * 4));
    GFC_OK_(gfc_pack_conv3x3(dw, dp, 512, 128, st));
    p.wh = dp; p.bias_h = upload(b); p.scale_h = upload(sc); p.shift_h = upload(sh);
    float* dq = nullptr;
    HIP_OK(hipMalloc(&dq, (size_t)16 * 512 * 128 * 4));
    GFC_OK_(gfc_pack_conv3x3_wino(dw, dq, 512, 128, st));
    p.wh_wino = dq;
  }
  { auto w = rd(65 * 256), b = rd(65), sc = rd(65), sh = rd(65);
    p.wp = upload(w); p.bias_p = upload(b); p.scale_p = upload(sc); p.shift_p = upload(sh); }
  { auto w = rd(256 * 256), b = rd(256), sc = rd(256), sh = rd(256);
    if (sh.empty()) return 1;
    p.wd = upload(w); p.bias_d = upload(b); p.scale_d = upload(sc); p.shift_d = upload(sh); }
  p.desc_dim = 256;
  p.conv_mode = 2;  // Winograd on fp32 MFMA, the modules' default (0 = direct implicit GEMM)
  fclose(f);

  const int B = 1, h8 = H / 8, w8 = W / 8;
  float* d_img = upload(image);
  float *heat, *desc_raw, *kpts, *ksc, *desc, *kout;
  int32_t* counts;
  void *ws, *ws2;
  const size_t wsb = gfc_sp_workspace_bytes(B, 1, H, W), wsb2 = gfc_sp_nms_select_workspace_bytes(B, h8 * 8, w8 * 8);
  HIP_OK(hipMalloc(&heat, (size_t)h8 * 8 * w8 * 8 * 4));
  HIP_OK(hipMalloc(&desc_raw, (size_t)h8 * w8 * 256 * 4));
  HIP_OK(hipMalloc(&kpts, (size_t)K * 2 * 4));
  HIP_OK(hipMalloc(&kout, (size_t)K * 2 * 4));
  HIP_OK(hipMalloc(&ksc, (size_t)K * 4));
  HIP_OK(hipMalloc(&desc, (size_t)K * 256 * 4));
  HIP_OK(hipMalloc(&counts, 4));
  HIP_OK(hipMalloc(&ws, wsb));
  HIP_OK(hipMalloc(&ws2, wsb2));
  GFC_OK_(gfc_sp_dense(&p, d_img, B, 1, H, W, heat, desc_raw, ws, wsb, nullptr, st));
  GFC_OK_(gfc_sp_nms_select(heat, B, h8 * 8, w8 * 8, 3, 4, nullptr, 0.0f, K, K, nullptr, kpts, ksc, counts, ws2, wsb2, st));
  GFC_OK_(gfc_sp_sample(desc_raw, B, h8, w8, 256, kpts, counts, K, GFC_SAMPLE_OPEN, desc, kout, st));
  HIP_OK(hipStreamSynchronize(st));
  // error behaviour of the ABI: bad arguments come back as status codes, nothing is launched
  if (gfc_sp_dense(&p, d_img, B, 2, H, W, heat, desc_raw, ws, wsb, nullptr, st) != GFC_ERR_INVALID) return 4;
  if (gfc_sp_dense(&p, d_img, B, 1, H, W, heat, desc_raw, ws, 16, nullptr, st) != GFC_ERR_WORKSPACE) return 4;
  if (gfc_sp_nms(heat, B, H, W, 9, 4, nullptr, heat, st) != GFC_ERR_UNSUPPORTED) return 4;
  int32_t n = 0;
  std::vector<float> hk((size_t)K * 2), hs(K), hd((size_t)K * 256);
  HIP_OK(hipMemcpy(&n, counts, 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hk.data(), kout, hk.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hs.data(), ksc, hs.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hd.data(), desc, hd.size() * 4, hipMemcpyDeviceToHost));
  printf("%s\ncount %d\n", gfc_version(), n);
  for (int i = 0; i < n; ++i) {
    double ds = 0;
    for (int c = 0; c < 256; ++c) ds += hd[(size_t)i * 256 + c] * (c % 7 + 1);
    printf("kp %.1f %.1f %.9g %.9g\n", hk[2 * i], hk[2 * i + 1], hs[i], ds);
  }
  return 0;
}
