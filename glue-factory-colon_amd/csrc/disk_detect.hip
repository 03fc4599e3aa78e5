// DISK extractor, everything behind the network (reference gluefactory/models/extractors/disk_kornia.py:29-53,55-137;
// the stages themselves live in the third-party `kornia` package, unpinned >= 0.6.12 and absent from the build
// container: `kornia.feature.disk.detector.heatmap_to_keypoints` / `nms` and `Keypoints.merge_with_descriptors` are
// restated from their published source -- parity for these is pinned by the oracle's torch restatement only).
//
//   heat-map [B,H,W], dense descriptors [B,D,H,W] (NCHW, as the U-Net emits them)
//     -> window NMS: a pixel survives iff it is the arg-max `F.max_pool2d(..., return_indices=True)` reports for the
//        window centred on it (first maximum in row-major window order wins ties) and its score > cutoff
//     -> top-n: threshold = the (n+1)-th largest surviving score (`torch.kthvalue(-s, min(n+1, count))`), keep
//        scores STRICTLY above it (with count <= n this drops the minimum: kornia's behaviour), row-major order,
//        clipped to the first n
//     -> descriptors read at the integer pixel, L2-normalised over D (F.normalize, eps 1e-12).
// HBM-bound byte / index work: coalesced row reads, LDS halo tile for the window test, one workgroup per image for
// the order-preserving compaction (wave-contiguous segments of the map, then a compact survivor list).
#include "common.h"

__device__ __forceinline__ unsigned int dk_order_bits(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dk_from_order_bits(unsigned int o) {
  unsigned int u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}

#define DK_T 32  // output tile edge

// cand[b][y][x] = score if the pixel survives the window NMS and the cutoff, else -inf
template <int RAD>
__global__ __launch_bounds__(256) void disk_nms_kernel(const float* __restrict__ heat, int H, int W, float cutoff,
                                                       float* __restrict__ cand) {
  constexpr int R = DK_T + 2 * RAD;
  __shared__ float t[R][R + 1];
  const int b = blockIdx.z, x0 = blockIdx.x * DK_T, y0 = blockIdx.y * DK_T;
  const float* src = heat + (size_t)b * H * W;
  for (int i = threadIdx.x; i < R * R; i += 256) {
    const int ly = i / R, lx = i % R;
    const int gy = y0 - RAD + ly, gx = x0 - RAD + lx;
    t[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : -INFINITY;  // -inf padding
  }
  __syncthreads();
  for (int i = threadIdx.x; i < DK_T * DK_T; i += 256) {
    const int ly = i / DK_T, lx = i % DK_T;
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy >= H || gx >= W) continue;
    const float v = t[ly + RAD][lx + RAD];
    // max_pool2d scans the window row-major and replaces its running maximum on `val > max` (or NaN): the reported
    // index is the FIRST maximum.  The centre is that index iff nothing before it is >= v and nothing after it is > v.
    bool keep = v > cutoff;
#pragma unroll
    for (int dy = 0; dy < 2 * RAD + 1; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2 * RAD + 1; ++dx) {
        const float o = t[ly + dy][lx + dx];
        const bool before = dy < RAD || (dy == RAD && dx < RAD);
        const bool centre = dy == RAD && dx == RAD;
        if (!centre) keep = keep && (before ? o < v : o <= v);
      }
    cand[((size_t)b * H + gy) * W + gx] = keep ? v : -INFINITY;
  }
}

// One workgroup (1024 threads = 16 waves) per image.  The dense survivor map is read TWICE, by wave-contiguous segments
// with eight independent loads in flight per lane (a single workgroup streaming 1.2 MB with one load per wave in
// flight is latency-bound: 0.13 ms per VGA image in the first version, whose compaction also took three workgroup
// barriers per 1024 pixels): pass 1 counts the survivors of each wave's segment, pass 2 writes them -- in row-major
// order, because the segments are contiguous and ordered -- to a compact (index, score) list.  Everything after that
// (radix select of the threshold, the final ordered filter) walks the list, a few thousand entries.
#define DKS_U 8
__global__ __launch_bounds__(1024) void disk_select_kernel(const float* __restrict__ cand, int H, int W, int n, int cap,
                                                           float* __restrict__ kpts, float* __restrict__ kscores,
                                                           int* __restrict__ counts, int* __restrict__ list_idx_all,
                                                           float* __restrict__ list_sc_all) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_want;
  __shared__ unsigned int wsum[16];
  __shared__ int wave_cnt[16];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = H * W;
  const float* src = cand + (size_t)b * total;
  int* list_idx = list_idx_all + (size_t)b * total;
  float* list_sc = list_sc_all + (size_t)b * total;
  // ---- pass 1: survivors per wave segment ----
  const int seg = (((total + 15) / 16) + 63) & ~63;
  const int s0 = wave * seg, s1 = min(s0 + seg, total);
  int c = 0;
  for (int i0 = s0; i0 < s1; i0 += 64 * DKS_U) {
    float v[DKS_U];
#pragma unroll
    for (int u = 0; u < DKS_U; ++u) {
      const int i = i0 + u * 64 + lane;
      v[u] = i < s1 ? src[i] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < DKS_U; ++u) c += __popcll(__ballot(v[u] > -INFINITY));
  }
  if (lane == 0) wave_cnt[wave] = c;
  __syncthreads();
  int base = 0, count = 0;
  for (int w = 0; w < 16; ++w) {
    const int t = wave_cnt[w];
    if (w < wave) base += t;
    count += t;
  }
  // ---- pass 2: the ordered survivor list ----
  for (int i0 = s0; i0 < s1; i0 += 64 * DKS_U) {
    float v[DKS_U];
#pragma unroll
    for (int u = 0; u < DKS_U; ++u) {
      const int i = i0 + u * 64 + lane;
      v[u] = i < s1 ? src[i] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < DKS_U; ++u) {
      const bool keep = v[u] > -INFINITY;
      const unsigned long long bal = __ballot(keep);
      if (keep) {
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        list_idx[slot] = i0 + u * 64 + lane;
        list_sc[slot] = v[u];
      }
      base += __popcll(bal);
    }
  }
  __threadfence();
  __syncthreads();
  float thr = -INFINITY;  // keep score > thr
  bool drop_all = false;
  if (n >= 0) {
    if (count == 0) {
      drop_all = true;  // (torch.kthvalue on an empty tensor raises in the reference; nothing to return either way)
    } else {
      // threshold = the k-th largest survivor, k = min(n + 1, count): MSB-first radix select on the order bits; the
      // bin that holds it is found by a 256-thread scan (all four passes run: the threshold is an exact score)
      if (tid == 0) { s_prefix = 0u; s_want = (unsigned int)min(n + 1, count); }
      __syncthreads();
      for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        const unsigned int prefix = s_prefix, want = s_want;
        __syncthreads();
        const unsigned int hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < count; i += 1024) {
          const unsigned int o = dk_order_bits(list_sc[i]);
          if ((o & hi_mask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
        }
        __syncthreads();
        unsigned int hv = 0, inc = 0;
        if (tid < 256) {  // waves 0..3, whole waves; bins in descending order
          hv = hist[255 - tid];
          inc = hv;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const unsigned int up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
          }
          if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (tid < 256) {
          for (int i = 0; i < wave; ++i) inc += wsum[i];
          if (inc >= want && inc - hv < want) {  // the first bin from the top whose running count reaches `want`
            s_prefix = prefix | ((unsigned int)(255 - tid) << shift);
            s_want = want - (inc - hv);
          }
        }
        __syncthreads();
      }
      thr = dk_from_order_bits(s_prefix);
    }
  }
  // ---- ordered filter of the list: {score > thr} in row-major order, clipped to the first `lim` ----
  const int lim = n >= 0 ? min(n, cap) : cap;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < count; i0 += 1024) {
    const int i = i0 + tid;
    const float v = i < count ? list_sc[i] : -INFINITY;
    const bool keep = !drop_all && i < count && v > thr;
    const unsigned long long bal = __ballot(keep);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base, all = 0;
    for (int w = 0; w < 16; ++w) {
      const int t = wave_cnt[w];
      if (w < wave) off += t;
      all += t;
    }
    const int slot = off + in_wave;
    if (keep && slot < lim) {
      const int idx = list_idx[i];
      kpts[((size_t)b * cap + slot) * 2] = (float)(idx % W);
      kpts[((size_t)b * cap + slot) * 2 + 1] = (float)(idx / W);
      kscores[(size_t)b * cap + slot] = v;
    }
    __syncthreads();
    if (tid == 0) s_base += all;
    __syncthreads();
    if (s_base >= lim) break;  // uniform
  }
  if (tid == 0) counts[b] = min(s_base, lim);
}

extern "C" size_t gfc_disk_select_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return 3 * gfc_align((size_t)B * H * W * sizeof(float));  // survivor map, survivor list (index, score)
}

extern "C" int gfc_disk_nms_select(const float* heatmap, int B, int H, int W, int window, float cutoff, int n, int cap,
                                   float* kpts, float* kscores, int32_t* counts, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!heatmap || !kpts || !kscores || !counts || !ws || B <= 0 || H <= 0 || W <= 0 || cap <= 0) return GFC_ERR_INVALID;
  if (window % 2 != 1 || window < 1) return GFC_ERR_INVALID;  // kornia raises for even windows
  if (n >= 0 && cap < n) return GFC_ERR_INVALID;
  if (n < 0 && cap < H * W) return GFC_ERR_INVALID;
  if (ws_bytes < gfc_disk_select_workspace_bytes(B, H, W)) return GFC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* cand = (float*)ws;
  dim3 grid((W + DK_T - 1) / DK_T, (H + DK_T - 1) / DK_T, B);
  switch (window) {
    case 1: hipLaunchKernelGGL(disk_nms_kernel<0>, grid, dim3(256), 0, st, heatmap, H, W, cutoff, cand); break;
    case 3: hipLaunchKernelGGL(disk_nms_kernel<1>, grid, dim3(256), 0, st, heatmap, H, W, cutoff, cand); break;
    case 5: hipLaunchKernelGGL(disk_nms_kernel<2>, grid, dim3(256), 0, st, heatmap, H, W, cutoff, cand); break;
    case 7: hipLaunchKernelGGL(disk_nms_kernel<3>, grid, dim3(256), 0, st, heatmap, H, W, cutoff, cand); break;
    case 9: hipLaunchKernelGGL(disk_nms_kernel<4>, grid, dim3(256), 0, st, heatmap, H, W, cutoff, cand); break;
    default: return GFC_ERR_UNSUPPORTED;
  }
  const size_t plane = gfc_align((size_t)B * H * W * sizeof(float));
  int* list_idx = reinterpret_cast<int*>((char*)ws + plane);
  float* list_sc = reinterpret_cast<float*>((char*)ws + 2 * plane);
  hipLaunchKernelGGL(disk_select_kernel, dim3(B), dim3(1024), 0, st, cand, H, W, n, cap, kpts, kscores, counts, list_idx,
                     list_sc);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// descriptors[b, :, y, x] -> out[b, slot, :] / max(||.||, 1e-12); one wave per key point, lanes over channels.
// The dense array is addressed through strides: NCHW (kornia's network: pixel 1, channel H*W) or NHWC (the native
// network of csrc/disk_unet.hip: pixel D, channel 1 -- one 512-byte row per key point).
__global__ __launch_bounds__(256) void disk_gather_desc_kernel(const float* __restrict__ dense, int D, int H, int W,
                                                               long long pix_stride, long long ch_stride,
                                                               const float* __restrict__ kpts,
                                                               const int* __restrict__ counts, int cap,
                                                               float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (slot >= cap) return;
  float* o = out + ((size_t)b * cap + slot) * D;
  if (counts && slot >= counts[b]) {  // beyond this image's key points: zeros (pad_and_stack mode "zeros")
    for (int c = lane; c < D; c += 64) o[c] = 0.f;
    return;
  }
  const int x = (int)kpts[((size_t)b * cap + slot) * 2], y = (int)kpts[((size_t)b * cap + slot) * 2 + 1];
  const float* src = dense + (size_t)b * D * H * W + ((size_t)y * W + x) * pix_stride;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float v = src[(size_t)c * ch_stride];
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  for (int c = lane; c < D; c += 64) o[c] = src[(size_t)c * ch_stride] * inv;
}

extern "C" int gfc_disk_gather_descriptors(const float* dense_nchw, int B, int D, int H, int W, const float* kpts,
                                           const int32_t* counts, int cap, float* out, void* stream) {
  if (!dense_nchw || !kpts || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || cap <= 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(disk_gather_desc_kernel, dim3((cap + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, dense_nchw, D, H,
                     W, 1ll, (long long)H * W, kpts, counts, cap, out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_disk_gather_descriptors_nhwc(const float* dense_nhwc, int B, int D, int H, int W, const float* kpts,
                                                const int32_t* counts, int cap, float* out, void* stream) {
  if (!dense_nhwc || !kpts || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || cap <= 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(disk_gather_desc_kernel, dim3((cap + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, dense_nhwc, D, H,
                     W, (long long)D, 1ll, kpts, counts, cap, out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
