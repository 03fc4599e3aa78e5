// See runtime.h: the one place where the environment is read and per-device facts are cached.
#include "runtime.h"

#include <stdlib.h>

#include <mutex>

static GfcKnobs g_knobs;
static std::once_flag g_knobs_once;

static int env_int(const char* name, int fallback) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : fallback;
}

const GfcKnobs& gfc_knobs() {
  std::call_once(g_knobs_once, [] {
    g_knobs.gemm_tile = env_int("GFC_GEMM_TILE", 0);
    g_knobs.attn_cfg = env_int("GFC_ATTN_CFG", 0);
    g_knobs.conv_kc = env_int("GFC_CONV_KC", 0);
    g_knobs.conv_persist = env_int("GFC_CONV_PERSIST", -1);
    g_knobs.ffn_fused = env_int("GFC_FFN_FUSED", -1);
    g_knobs.ffn_mlp = env_int("GFC_FFN_MLP", -1);
    g_knobs.assign_mode = env_int("GFC_ASSIGN_MODE", 0);
    g_knobs.gemm_epi = env_int("GFC_GEMM_EPI", 0);
    g_knobs.gemm_stagger = env_int("GFC_GEMM_STAGGER", 0);
    g_knobs.nms_mode = env_int("GFC_NMS_MODE", 0);
    g_knobs.stem_f43 = env_int("GFC_STEM_F43", 1);
    g_knobs.xcd_remap = env_int("GFC_XCD_REMAP", 1);
  });
  return g_knobs;
}

int gfc_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cus[dev].store(n, std::memory_order_relaxed);
  return n;
}
