// Pure C-ABI consumer of the MATCHER: no Python, no torch -- HIP runtime + include/gfc_amd.h only.
// Reads a flat blob written by tests/test_gpu_cabi.py (pair sizes, key points, descriptors, image sizes, and the
// LightGlue parameters in the layouts gfc_lg_params documents), runs gfc_lg_forward_ragged over B pairs with their own
// key-point counts and prints matches0 / matching_scores0 of every pair, which the test compares with the Python
// boundary module (LightGlue.forward_pairs) bit for bit.
// Build: hipcc --offload-arch=gfx950 cabi_matcher.cpp -I include -L <pkg> -lgfc_amd
#include <hip/hip_runtime.h>

#include <cstdint>
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

static FILE* g_f = nullptr;
static const float* rd_dev(size_t n) {  // next n floats of the blob -> device memory
  std::vector<float> v(n);
  if (fread(v.data(), 4, n, g_f) != n) { fprintf(stderr, "short blob\n"); exit(1); }
  float* d = nullptr;
  if (hipMalloc(&d, n * 4) != hipSuccess || hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) exit(2);
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: cabi_matcher blob.bin\n"); return 1; }
  g_f = fopen(argv[1], "rb");
  if (!g_f) return 1;
  int32_t hdr[2];  // B, n_layers
  if (fread(hdr, 4, 2, g_f) != 2) return 1;
  const int B = hdr[0], NL = hdr[1];
  std::vector<int32_t> m(B), n(B);
  if (fread(m.data(), 4, B, g_f) != (size_t)B || fread(n.data(), 4, B, g_f) != (size_t)B) return 1;
  size_t sm = 0, sn = 0, sla = 0;
  for (int i = 0; i < B; ++i) { sm += m[i]; sn += n[i]; sla += (size_t)(m[i] + 1) * (n[i] + 1); }
  const size_t R = sm + sn;
  const float* kpts = rd_dev(R * 2);
  const float* desc = rd_dev(R * 256);
  const float* size0 = rd_dev((size_t)B * 2);
  const float* size1 = rd_dev((size_t)B * 2);

  gfc_lg_params p = {};
  p.n_layers = NL;
  p.input_dim = 256;
  p.posenc_dim = 2;
  p.posenc_wr = rd_dev(32 * 2);
  for (int l = 0; l < NL; ++l) {
    p.wqkv[l] = rd_dev(768 * 256);   p.bqkv[l] = rd_dev(768);          // rows re-ordered to [q | k | v], head-major
    p.s_ffn0_w[l] = rd_dev(512 * 512); p.s_ffn0_b[l] = rd_dev(512);      // out_proj folded in (s_out_w stays NULL)
    p.s_ln_g[l] = rd_dev(512);       p.s_ln_b[l] = rd_dev(512);
    p.s_ffn3_w[l] = rd_dev(256 * 512); p.s_ffn3_b[l] = rd_dev(256);
    p.c_qkv_w[l] = rd_dev(512 * 256); p.c_qkv_b[l] = rd_dev(512);       // to_qk | to_v stacked
    p.c_ffn0_w[l] = rd_dev(512 * 512); p.c_ffn0_b[l] = rd_dev(512);
    p.c_ln_g[l] = rd_dev(512);       p.c_ln_b[l] = rd_dev(512);
    p.c_ffn3_w[l] = rd_dev(256 * 512); p.c_ffn3_b[l] = rd_dev(256);
  }
  p.final_proj_w[NL - 1] = rd_dev(256 * 256);
  p.final_proj_b[NL - 1] = rd_dev(256);
  p.matchability_w[NL - 1] = rd_dev(256);
  p.matchability_b[NL - 1] = rd_dev(1);
  fclose(g_f);

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int64_t *m0, *m1;
  float *ms0, *ms1, *la, *rows;
  void* ws;
  const size_t wsb = gfc_lg_ragged_workspace_bytes(B, m.data(), n.data());
  if (wsb == 0) return 4;
  HIP_OK(hipMalloc(&m0, sm * 8));
  HIP_OK(hipMalloc(&m1, sn * 8));
  HIP_OK(hipMalloc(&ms0, sm * 4));
  HIP_OK(hipMalloc(&ms1, sn * 4));
  HIP_OK(hipMalloc(&la, sla * 4));
  HIP_OK(hipMalloc(&rows, R * 256 * 4));
  HIP_OK(hipMalloc(&ws, wsb));
  GFC_OK_(gfc_lg_forward_ragged(&p, kpts, desc, size0, size1, nullptr, B, m.data(), n.data(), 0.1f, m0, m1, ms0, ms1, la, rows,
                                ws, wsb, nullptr, st));
  HIP_OK(hipStreamSynchronize(st));
  // error behaviour: bad arguments come back as status codes
  if (gfc_lg_forward_ragged(&p, kpts, desc, size0, size1, nullptr, B, m.data(), n.data(), 0.1f, m0, m1, ms0, ms1, la, rows, ws, 64,
                            nullptr, st) != GFC_ERR_WORKSPACE) return 5;
  if (gfc_lg_forward_ragged(&p, kpts, desc, size0, size1, nullptr, 0, m.data(), n.data(), 0.1f, m0, m1, ms0, ms1, la, rows, ws, wsb,
                            nullptr, st) != GFC_ERR_INVALID) return 5;
  if (gfc_lg_forward_ragged(&p, kpts, kpts, size0, size1, nullptr, B, m.data(), n.data(), 0.1f, m0, m1, ms0, ms1, la,
                            const_cast<float*>(kpts), ws, wsb, nullptr, st) != GFC_ERR_INVALID) return 5;  // rows == desc
  std::vector<int64_t> hm0(sm);
  std::vector<float> hs0(sm);
  HIP_OK(hipMemcpy(hm0.data(), m0, sm * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hs0.data(), ms0, sm * 4, hipMemcpyDeviceToHost));
  printf("%s\npairs %d\n", gfc_version(), B);
  size_t o = 0;
  for (int i = 0; i < B; ++i) {
    for (int j = 0; j < m[i]; ++j) printf("m %d %d %lld %.9g\n", i, j, (long long)hm0[o + j], hs0[o + j]);
    o += m[i];
  }
  return 0;
}
