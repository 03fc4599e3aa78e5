// Process-wide, thread-safe host state of the library (all of it read-only after its first use):
//  * the tuning knobs ($GFC_GEMM_TILE, $GFC_ATTN_CFG, ...) are read ONCE, by one std::call_once, for every
//    translation unit -- host threads that drive their own HIP streams (export_predictions(workers=N)) all see
//    the same values;
//  * per-device facts (CU count) and per-(kernel, device) attributes (dynamic LDS above 64 KB) are keyed by the
//    current HIP device, so one process may drive several GPUs.
// Nothing here influences results: the knobs select between kernel variants that are held to the same parity
// tests (tests/test_gpu_primitives.py::test_gemm_tile_variants_via_knob etc.).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

struct GfcKnobs {
  int gemm_tile;     // GFC_GEMM_TILE: 0 = by problem size, 3 = 64x64 tiles everywhere, 4 = 128x128 tiles everywhere
  int attn_cfg;      // GFC_ATTN_CFG: 0 = automatic
  int conv_kc;       // GFC_CONV_KC: 0 = automatic
  int conv_persist;  // GFC_CONV_PERSIST: -1 = automatic
  int ffn_fused;     // GFC_FFN_FUSED: -1 = automatic, 0 = GEMM + layernorm_gelu pass, 1 = row-owning fused GEMM
  int ffn_mlp;       // GFC_FFN_MLP: -1 = automatic, 0 = ffn[3] as a GEMM of its own, 1 = whole FFN + residual in one kernel
  int assign_mode;   // GFC_ASSIGN_MODE: 0 = automatic, 1 = five-pass tail, 2 = two-pass tail
  int gemm_epi;      // GFC_GEMM_EPI: 0 = automatic, 1 = float4 stores through the LDS transpose for every epilogue, 2 = direct
  int gemm_stagger;  // GFC_GEMM_STAGGER: start skew of the first-round GEMM workgroups in units of 8128 cycles per wave slot
  int nms_mode;      // GFC_NMS_MODE: 0 (default) = by problem size, 1 = LDS-image kernel, 2 = streaming kernel (waves walk column bands, rings in registers)
  int stem_f43;      // GFC_STEM_F43: 1 (default) = Winograd F(4x4,3x3) stem when its filters are supplied, 0 = F(2x2,3x3) stem
  int xcd_remap;     // GFC_XCD_REMAP: 1 (default) = XCD-aware work order (common.h: gfc_xcd_chunk), 0 = dispatch order
};
const GfcKnobs& gfc_knobs();

// number of compute units of the CURRENT device (cached per device)
int gfc_device_cus();

// hipFuncAttributeMaxDynamicSharedMemorySize, applied once per (kernel instantiation, device).
// `done` is one word per kernel instantiation (bit d = device d configured); safe from any thread.
inline void gfc_allow_dynamic_lds(const void* kernel, size_t bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  const unsigned long long bit = 1ull << dev;
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  done.fetch_or(bit, std::memory_order_release);
}
