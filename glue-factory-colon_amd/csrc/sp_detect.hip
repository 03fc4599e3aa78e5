// SuperPoint detection tail: heat-map decode, NMS, key-point selection, descriptor sampling.
// All HBM-bound byte/compare work: coalesced loads into LDS, wavefront reductions, no GEMM shapes.
#include "common.h"

#include <type_traits>
#include <utility>

// (The softmax over the 65 logits of a cell + depth-to-space lives in sp_heads.hip since round 5, fused with the
// detector's 1x1 convolution.)
// ------------------------------------------------------------------------------------------
// NMS (superpoint_open.py:36-51 == superpoint.py:63-83) fused with the border kill
// (superpoint_open.py:148-154; superpoint.py:249-260).
//   keep  = s == mp(s)
//   twice: near = mp(keep) > 0;  t = near ? 0 : s;  keep |= (t == mp(t)) & ~near
//   out   = keep ? s : 0
// mp = (2r+1)^2 max-pool, stride 1, -inf padding.  The 5 dependent pools need a 5r halo, so a
// 64x64 output tile works on a (64+10r)^2 LDS image; every pool is separable (row pass, column pass)
// and the pools of the 0/1 keep mask are byte ORs.  10 LDS passes per tile, 1024 threads.
// Float equality is evaluated on the same fp32 values the reference compares: bit-exact.
// ------------------------------------------------------------------------------------------
#define NT 64
#define NR_MAX 4

// order-preserving float -> uint map (selection keys)
__device__ __forceinline__ unsigned int float_order_bits(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_bits(unsigned int o) {
  unsigned int u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}

// (no NaNs reach the pools: one v_max_f32 / v_max3_f32 instead of compare + select; the keep masks are 0 / 1 bytes)
__device__ __forceinline__ float window_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ unsigned char window_max(unsigned char a, unsigned char b) { return a | b; }

// One separable max / OR pass over the LDS image.  Each work item owns SEG consecutive positions of
// one line and slides the (2*RAD+1) window through registers: SEG + 2*RAD LDS reads for SEG outputs.
// ROW passes map consecutive lanes to consecutive lines (odd line stride RS -> conflict-free);
// COLUMN passes map consecutive lanes to consecutive columns.
template <int RAD, int R, int RS, int SEG, int NSEG, bool ROW, typename T, typename F>
__device__ __forceinline__ void window_pass(const T* __restrict__ src, T lowest, int tid, F&& emit) {
  for (int item = tid; item < R * NSEG; item += 1024) {
    const int line = item % R, seg = item / R;
    const int p0 = seg * SEG;
    T w[SEG + 2 * RAD];
#pragma unroll
    for (int j = 0; j < SEG + 2 * RAD; ++j) {
      const int p = p0 - RAD + j;
      const bool ok = p >= 0 && p < R;
      const int idx = ROW ? line * RS + p : p * RS + line;
      w[j] = ok ? src[idx] : lowest;
    }
    // max over 2 RAD + 1 taps from maxima of three: t3[j] = max(w[j..j+2]); 3 taps: t3[j]; 5: t3[j] | t3[j+2];
    // 7: t3[j] | t3[j+2] | t3[j+4]; 9: t3[j] | t3[j+3] | t3[j+6]  (two v_max3 per output instead of 2 RAD compares)
    static_assert(RAD >= 1 && RAD <= 4, "window_pass");
    T t3[SEG + 2 * RAD - 2];
#pragma unroll
    for (int j = 0; j < SEG + 2 * RAD - 2; ++j) t3[j] = window_max(window_max(w[j], w[j + 1]), w[j + 2]);
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      if (p0 + j < R) {
        T m;
        if constexpr (RAD == 1) m = t3[j];
        else if constexpr (RAD == 2) m = window_max(t3[j], t3[j + 2]);
        else if constexpr (RAD == 3) m = window_max(window_max(t3[j], t3[j + 2]), t3[j + 4]);
        else m = window_max(window_max(t3[j], t3[j + 3]), t3[j + 6]);
        emit(ROW ? line * RS + p0 + j : (p0 + j) * RS + line, m);
      }
    }
  }
}

template <int RAD>
__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ heat, int H, int W, int border,
                                                   const int* __restrict__ valid_wh, float* __restrict__ out,
                                                   float cand_thr, unsigned long long* __restrict__ cand,
                                                   int* __restrict__ cand_count, int xcd_remap) {
  constexpr int HALO = 5 * RAD;
  constexpr int R = NT + 2 * HALO;
  constexpr int RS = R | 1;
  constexpr int NSEG = 1024 / R;
  constexpr int SEG = (R + NSEG - 1) / NSEG;
  extern __shared__ __attribute__((aligned(16))) float nsm[];
  float* s = nsm;            // scores, -inf outside the image
  float* t0 = s + R * RS;
  float* t1 = t0 + R * RS;
  unsigned char* keep = reinterpret_cast<unsigned char*>(t1 + R * RS);  // bit0 keep, bit1 near
  unsigned char* tb = keep + R * RS;

  const int tid = threadIdx.x;
  const int tiles_x = (W + NT - 1) / NT;
  // XCD-aware order (common.h): an XCD walks whole images tile by tile, the 5r halo a tile shares with its
  // neighbours is then served by that XCD's L2
  unsigned tile = blockIdx.x, img = blockIdx.y;
  if (xcd_remap) {
    const unsigned t = gfc_xcd_chunk(tile + gridDim.x * img, gridDim.x * gridDim.y);
    img = t / gridDim.x;
    tile = t - img * gridDim.x;
  }
  const int tx = tile % tiles_x, ty = tile / tiles_x, b = img;
  const int gx0 = tx * NT - HALO, gy0 = ty * NT - HALO;
  const float* hb = heat + (size_t)b * H * W;
  for (int i = tid; i < R * R; i += 1024) {
    const int y = i / R, x = i % R;
    const int gy = gy0 + y, gx = gx0 + x;
    s[y * RS + x] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? hb[(size_t)gy * W + gx] : -INFINITY;
  }
  __syncthreads();
  // keep = (s == mp(s)), restricted to in-image positions
  window_pass<RAD, R, RS, SEG, NSEG, true, float>(s, -INFINITY, tid, [&](int i, float m) { t0[i] = m; });
  __syncthreads();
  window_pass<RAD, R, RS, SEG, NSEG, false, float>(t0, -INFINITY, tid, [&](int i, float m) {
    keep[i] = (s[i] == m && s[i] != -INFINITY) ? 1 : 0;
  });
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    // near = mp(keep) > 0: separable OR of the keep bytes
    window_pass<RAD, R, RS, SEG, NSEG, true, unsigned char>(keep, (unsigned char)0, tid,
                                                            [&](int i, unsigned char m) { tb[i] = m & 1; });
    __syncthreads();
    window_pass<RAD, R, RS, SEG, NSEG, false, unsigned char>(tb, (unsigned char)0, tid, [&](int i, unsigned char m) {
      const bool near = m != 0;
      keep[i] = (keep[i] & 1) | (near ? 2 : 0);
      const float v = s[i];
      t0[i] = (v == -INFINITY) ? v : (near ? 0.f : v);  // supp_scores, -inf padding outside the image
    });
    __syncthreads();
    window_pass<RAD, R, RS, SEG, NSEG, true, float>(t0, -INFINITY, tid, [&](int i, float m) { t1[i] = m; });
    __syncthreads();
    window_pass<RAD, R, RS, SEG, NSEG, false, float>(t1, -INFINITY, tid, [&](int i, float m) {
      unsigned char k = keep[i];
      if (!(k & 2) && t0[i] == m && s[i] != -INFINITY) k |= 1;
      keep[i] = k & 1;
    });
    __syncthreads();
  }
  int vw = W, vh = H;
  if (valid_wh) { vw = valid_wh[2 * b]; vh = valid_wh[2 * b + 1]; }
  float* ob = out ? out + (size_t)b * H * W : nullptr;
  // fused selection: pixels above the threshold go straight to the per-image key list.  Slots are reserved
  // per workgroup (LDS counter, then ONE global atomic per tile): per-candidate global atomics on the image's
  // counter serialise (measured 2.5 ms instead of 0.4 ms per call).  Order inside the list is irrelevant:
  // keys are unique and the selection kernel sorts / selects on them.
  __shared__ int tile_count, tile_base;
  if (tid == 0) tile_count = 0;
  __syncthreads();
  float vals[NT * NT / 1024];
  int slots[NT * NT / 1024];
#pragma unroll
  for (int it = 0; it < NT * NT / 1024; ++it) {
    const int i = tid + it * 1024;
    const int y = i / NT, x = i % NT;
    const int gy = ty * NT + y, gx = tx * NT + x;
    slots[it] = -1;
    vals[it] = 0.f;
    if (gy >= H || gx >= W) continue;
    const int li = (y + HALO) * RS + (x + HALO);
    float v = keep[li] ? s[li] : 0.f;
    if (border > 0 && (gy < border || gx < border || gy >= vh - border || gx >= vw - border)) v = -1.f;
    if (ob) ob[(size_t)gy * W + gx] = v;
    vals[it] = v;
    if (cand && v > cand_thr) slots[it] = atomicAdd(&tile_count, 1);
  }
  if (cand) {
    __syncthreads();
    if (tid == 0) tile_base = tile_count > 0 ? atomicAdd(&cand_count[b], tile_count) : 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NT * NT / 1024; ++it) {
      if (slots[it] >= 0) {
        const int i = tid + it * 1024;
        const unsigned int idx = (unsigned int)((ty * NT + i / NT) * W + tx * NT + i % NT);
        cand[(size_t)b * H * W + tile_base + slots[it]] =
            ((unsigned long long)float_order_bits(vals[it]) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Streaming NMS (round 4): the same five dependent pools, but without the LDS image and its ten passes.  One WAVE owns a
// band of 64 columns (lane = column, 64 - 10 r of them stored) and walks down a segment of 64 output rows (+ 5 r halo
// rows above and below): the horizontal half of every pool is two or three wave shifts (DPP wave_shr / wave_shl) and a
// v_max3 / v_or3, the vertical half a v_max3 tree over a ring of the last 2 r + 1 rows kept IN REGISTERS (the row loop is
// unrolled by the ring length, so every ring index is a compile-time constant).  Stage k of the chain runs k r rows
// behind the row being loaded:
//   row y1 = row - r:   keep0 = s == mp(s)                    row y2 = row - 2r:  near1 = mp(keep0) > 0, t1 = near1 ? 0 : s
//   row y3 = row - 3r:  keep1 = keep0 | (t1 == mp(t1) & ~near1)   row y4 = row - 4r:  near2 = mp(keep1) > 0, t2 = ...
//   row y5 = row - 5r:  keep2 = keep1 | (t2 == mp(t2) & ~near2);  out = keep2 ? s : 0, border kill, candidate emission
// The same fp32 equality tests on the same values as nms_kernel: bit-identical output (test_nms_golden_bit_exact runs
// both).  No barrier, no shared image: 64 VGA images 344 us -> see DESIGN.md section 9.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_shr1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }  // x[lane - 1]
__device__ __forceinline__ int dpp_shl1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false); }  // x[lane + 1]
template <int K>
__device__ __forceinline__ float fshr(float x) {
  int v = __float_as_int(x);
#pragma unroll
  for (int i = 0; i < K; ++i) v = dpp_shr1(v, (int)0xff800000);
  return __int_as_float(v);
}
template <int K>
__device__ __forceinline__ float fshl(float x) {
  int v = __float_as_int(x);
#pragma unroll
  for (int i = 0; i < K; ++i) v = dpp_shl1(v, (int)0xff800000);
  return __int_as_float(v);
}
template <int K>
__device__ __forceinline__ int ishr(int v) {
#pragma unroll
  for (int i = 0; i < K; ++i) v = dpp_shr1(v, 0);
  return v;
}
template <int K>
__device__ __forceinline__ int ishl(int v) {
#pragma unroll
  for (int i = 0; i < K; ++i) v = dpp_shl1(v, 0);
  return v;
}
// max / OR over lanes l - RAD .. l + RAD
template <int RAD>
__device__ __forceinline__ float hwin(float x) {
  const float a = fmaxf(fmaxf(x, fshr<1>(x)), fshl<1>(x));
  if constexpr (RAD == 1) return a;
  else if constexpr (RAD == 2) return fmaxf(fmaxf(a, fshr<1>(a)), fshl<1>(a));
  else if constexpr (RAD == 3) return fmaxf(fmaxf(a, fshr<2>(a)), fshl<2>(a));
  else return fmaxf(fmaxf(a, fshr<3>(a)), fshl<3>(a));
}
template <int RAD>
__device__ __forceinline__ int hwin(int x) {
  const int a = x | ishr<1>(x) | ishl<1>(x);
  if constexpr (RAD == 1) return a;
  else if constexpr (RAD == 2) return a | ishr<1>(a) | ishl<1>(a);
  else if constexpr (RAD == 3) return a | ishr<2>(a) | ishl<2>(a);
  else return a | ishr<3>(a) | ishl<3>(a);
}
template <int L>
__device__ __forceinline__ float vwin(const float (&r)[L]) {
  float m = r[0];
#pragma unroll
  for (int i = 1; i < L; ++i) m = fmaxf(m, r[i]);
  return m;
}
template <int L>
__device__ __forceinline__ int vwin(const int (&r)[L]) {
  int m = r[0];
#pragma unroll
  for (int i = 1; i < L; ++i) m |= r[i];
  return m;
}

#define NMS_SEGH 64
// L consecutive rows with compile-time ring indices 0 .. L - 1
template <typename F, int... Is>
__device__ __forceinline__ void nms_unrolled_rows(F& f, int r, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}, r + Is), ...);
}
template <int RAD>
struct NmsStreamState {
  static constexpr int L = 2 * RAD + 1;
  float sr[L], srb[L], src[L], hs[L], t1[L], ht1[L], t2[L], ht2[L];  // sr / srb / src: s of rows r, r - L, r - 2L
  int k0[L], hk0[L], n1[L], k1[L], hk1[L], n2[L];
};

template <int RAD>
__global__ __launch_bounds__(256) void nms_stream_kernel(const float* __restrict__ heat, int H, int W, int border,
                                                         const int* __restrict__ valid_wh, float* __restrict__ out,
                                                         float cand_thr, unsigned long long* __restrict__ cand,
                                                         int* __restrict__ cand_count, int bands, int ntasks, int segh) {
  constexpr int L = 2 * RAD + 1, HALO = 5 * RAD, BW = 64 - 2 * HALO;
  __shared__ unsigned long long cbuf[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int task = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (task >= ntasks) return;  // (wave-uniform; no barrier in this kernel)
  const int bx = task % bands, sy = task / bands;
  const int col = bx * BW - HALO + lane;
  const bool col_in = col >= 0 && col < W;
  const bool lane_out = lane >= HALO && lane < 64 - HALO && col < W;
  const int y0 = sy * segh;
  const int y_end = min(y0 + segh, H);
  const float* hb = heat + (size_t)b * H * W;
  float* ob = out ? out + (size_t)b * H * W : nullptr;
  int vw = W, vh = H;
  if (valid_wh) { vw = valid_wh[2 * b]; vh = valid_wh[2 * b + 1]; }
  unsigned long long* cb = cbuf[wave];
  int ccnt = 0;
  auto flush = [&]() __attribute__((always_inline)) {
    if (ccnt > 0) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&cand_count[b], ccnt);
      base = __builtin_amdgcn_readfirstlane(base);
      for (int i = lane; i < ccnt; i += 64) cand[(size_t)b * H * W + base + i] = cb[i];
      ccnt = 0;
    }
  };
  auto load_s = [&](int y) __attribute__((always_inline)) -> float {
    return (y >= 0 && y < H && col_in) ? hb[(size_t)y * W + col] : -INFINITY;
  };

  NmsStreamState<RAD> st;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    st.sr[i] = st.srb[i] = st.src[i] = st.hs[i] = st.t1[i] = st.ht1[i] = st.t2[i] = st.ht2[i] = -INFINITY;
    st.k0[i] = st.hk0[i] = st.n1[i] = st.k1[i] = st.hk1[i] = st.n2[i] = 0;
  }
  // one row of the chain; I = (row - first row) mod L (compile time): ring slot of row - k RAD is (I - k RAD) mod L
  float pre[L];  // the next L rows of s, requested one group of rows ahead
  auto step = [&](auto Ic, int row) __attribute__((always_inline)) {
    constexpr int I = decltype(Ic)::value;
    constexpr int I1 = ((I - RAD) % L + L) % L, I2 = ((I - 2 * RAD) % L + L) % L, I3 = ((I - 3 * RAD) % L + L) % L;
    constexpr int I4 = ((I - 4 * RAD) % L + L) % L, I5 = ((I - 5 * RAD) % L + L) % L;
    const float s0 = pre[I];
    pre[I] = load_s(row + L);
    // s of rows r - L and r - 2L moves on through two more rings: the far delays 4 RAD (L <= 4 RAD < 2L) and 5 RAD
    // (L < 5 RAD < 3L) are read from them instead of from memory again.  Slot I of each ring is free at this point.
    st.src[I] = st.srb[I];
    st.srb[I] = st.sr[I];
    st.sr[I] = s0;
    const float s4 = st.srb[I4];  // row - 4 RAD = (row - L) - (4 RAD - L): written 4 RAD - L steps ago
    const float s5 = 5 * RAD < 2 * L ? st.srb[I5] : st.src[I5];
    st.hs[I] = hwin<RAD>(s0);
    {  // y1 = row - RAD: keep0
      const float m1 = vwin<L>(st.hs), sv = st.sr[I1];
      const int k = (sv == m1 && sv != -INFINITY) ? 1 : 0;
      st.k0[I1] = k;
      st.hk0[I1] = hwin<RAD>(k);
    }
    {  // y2 = row - 2 RAD: near1, supp scores t1
      const int near = vwin<L>(st.hk0);
      const float sv = st.sr[I2];
      st.n1[I2] = near;
      const float t = (sv == -INFINITY) ? sv : (near ? 0.f : sv);
      st.t1[I2] = t;
      st.ht1[I2] = hwin<RAD>(t);
    }
    {  // y3 = row - 3 RAD: keep1
      const float m2 = vwin<L>(st.ht1), t = st.t1[I3];
      int k = st.k0[I3];
      if (!st.n1[I3] && t == m2 && t != -INFINITY) k = 1;  // (t != -inf <=> the pixel is inside the image)
      st.k1[I3] = k;
      st.hk1[I3] = hwin<RAD>(k);
    }
    {  // y4 = row - 4 RAD: near2, supp scores t2
      const int near = vwin<L>(st.hk1);
      st.n2[I4] = near;
      const float t = (s4 == -INFINITY) ? s4 : (near ? 0.f : s4);
      st.t2[I4] = t;
      st.ht2[I4] = hwin<RAD>(t);
    }
    {  // y5 = row - 5 RAD: keep2 -> output
      const float m3 = vwin<L>(st.ht2), t = st.t2[I5];
      int k = st.k1[I5];
      if (!st.n2[I5] && t == m3 && t != -INFINITY) k = 1;
      const int y = row - 5 * RAD;
      if (y >= y0 && y < y_end) {  // wave-uniform
        float v = k ? s5 : 0.f;
        if (border > 0 && (y < border || col < border || y >= vh - border || col >= vw - border)) v = -1.f;
        if (lane_out && ob) ob[(size_t)y * W + col] = v;
        if (cand) {
          const bool c = lane_out && v > cand_thr;
          const unsigned long long mask = __ballot(c);
          const int n = __popcll(mask);
          if (ccnt + n > 256) flush();
          if (c) {
            const unsigned int idx = (unsigned int)(y * W + col);
            cb[ccnt + __popcll(mask & ((1ull << lane) - 1ull))] =
                ((unsigned long long)float_order_bits(v) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
          }
          ccnt += n;
        }
      }
    }
  };
  // k1 at y3 needs the keep0 of y3, which was written 2 RAD rows ago, and st.k0[I3] is the slot written then: the ring
  // slots of a quantity are L = 2 RAD + 1 apart in rows, so a value written at row r is intact until row r + 2 RAD.
  const int r_begin = y0 - 5 * RAD, r_stop = y_end + 5 * RAD;  // rows r_begin .. r_stop - 1 are loaded
#pragma unroll
  for (int i = 0; i < L; ++i) pre[i] = load_s(r_begin + i);
  for (int r = r_begin; r < r_stop; r += L) {
    nms_unrolled_rows(step, r, std::make_integer_sequence<int, L>{});
  }
  if (cand) flush();
}

// radius 0: the pools are identities -> only the border kill remains
__global__ void nms_r0_kernel(const float* __restrict__ heat, int H, int W, int border,
                              const int* __restrict__ valid_wh, float* __restrict__ out, float cand_thr,
                              unsigned long long* __restrict__ cand, int* __restrict__ cand_count) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)H * W) return;
  const int gy = (int)(i / W), gx = (int)(i % W);
  int vw = W, vh = H;
  if (valid_wh) { vw = valid_wh[2 * b]; vh = valid_wh[2 * b + 1]; }
  float v = heat[(size_t)b * H * W + i];
  if (border > 0 && (gy < border || gx < border || gy >= vh - border || gx >= vw - border)) v = -1.f;
  if (out) out[(size_t)b * H * W + i] = v;
  if (cand && v > cand_thr) {
    const int slot = atomicAdd(&cand_count[b], 1);
    cand[(size_t)b * H * W + slot] =
        ((unsigned long long)float_order_bits(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i);
  }
}

template <int RAD>
static int launch_nms(const float* heat, int B, int H, int W, int border, const int32_t* valid_wh, float* out,
                      float cand_thr, unsigned long long* cand, int* cand_count, hipStream_t st) {
  constexpr int R = NT + 10 * RAD, RS = R | 1;
  const size_t lds = (size_t)R * RS * (3 * sizeof(float) + 2);
  // default: the streaming kernel once its waves cover the chip a few times over (a wave walks ~64 + 10 r rows in
  // sequence: with few tasks the LDS-image kernel's wider parallelism wins -- 2 VGA images: 17 us against 44 us, 64
  // images: 328 us against 194 us).  GFC_NMS_MODE = 1: always the LDS-image kernel of round 1, 2: always streaming.
  constexpr int BW = 64 - 10 * RAD;
  const int bands = (W + BW - 1) / BW;
  // rows per wave task: 64 (+ 10 r halo rows recomputed), or 128 once that still leaves >= 4 tasks per SIMD
  int segh = NMS_SEGH;
  if ((long long)bands * ((H + 127) / 128) * B >= 4096) segh = 128;
  const int segs = (H + segh - 1) / segh, ntasks = bands * segs;
  const int mode = gfc_knobs().nms_mode;
  const long long thr = 1800;
  if (mode == 2 || (mode == 0 && (long long)bands * ((H + NMS_SEGH - 1) / NMS_SEGH) * B >= thr)) {
    hipLaunchKernelGGL(nms_stream_kernel<RAD>, dim3((ntasks + 3) / 4, B), dim3(256), 0, st, heat, H, W, border, valid_wh, out,
                       cand_thr, cand, cand_count, bands, ntasks, segh);
    GFC_LAUNCH_CHECK();
    return GFC_OK;
  }
  static std::atomic<unsigned long long> lds_ok{0};  // per (instantiation, device): runtime.h
  if (lds > 64 * 1024) gfc_allow_dynamic_lds((const void*)nms_kernel<RAD>, lds, lds_ok);
  dim3 grid(((W + NT - 1) / NT) * ((H + NT - 1) / NT), B);
  hipLaunchKernelGGL(nms_kernel<RAD>, grid, dim3(1024), lds, st, heat, H, W, border, valid_wh, out, cand_thr, cand,
                     cand_count, gfc_knobs().xcd_remap != 0 && (long long)grid.x * grid.y >= 16);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

static int nms_dispatch(const float* heatmap, int B, int H, int W, int radius, int border, const int32_t* valid_wh,
                        float* out, float cand_thr, unsigned long long* cand, int* cand_count, hipStream_t st) {
  switch (radius) {
    case 0: {
      dim3 grid((unsigned)(((long long)H * W + 255) / 256), B);
      hipLaunchKernelGGL(nms_r0_kernel, grid, dim3(256), 0, st, heatmap, H, W, border, valid_wh, out, cand_thr, cand,
                         cand_count);
      GFC_LAUNCH_CHECK();
      return GFC_OK;
    }
    case 1: return launch_nms<1>(heatmap, B, H, W, border, valid_wh, out, cand_thr, cand, cand_count, st);
    case 2: return launch_nms<2>(heatmap, B, H, W, border, valid_wh, out, cand_thr, cand, cand_count, st);
    case 3: return launch_nms<3>(heatmap, B, H, W, border, valid_wh, out, cand_thr, cand, cand_count, st);
    default: return launch_nms<4>(heatmap, B, H, W, border, valid_wh, out, cand_thr, cand, cand_count, st);
  }
}

extern "C" int gfc_sp_nms(const float* heatmap, int B, int H, int W, int radius, int border,
                          const int32_t* valid_wh, float* out, void* stream) {
  if (!heatmap || !out || B <= 0 || H <= 0 || W <= 0 || radius < 0 || border < 0) return GFC_ERR_INVALID;
  if (radius > NR_MAX) return GFC_ERR_UNSUPPORTED;
  return nms_dispatch(heatmap, B, H, W, radius, border, valid_wh, out, 0.f, nullptr, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Selection (superpoint_open.py:156-192,54-58; superpoint.py:262-300,86-90): one workgroup
// (1024 threads) per image.
//   pass A  ordered compaction of the pixels with score > th (row-major) into 64-bit keys
//           key = score_bits << 32 | (0xFFFFFFFF - linear_index)   (scores > th >= ... may be any
//           float; order-preserving transform below) -- keys are unique, so the top-k set and its
//           order (score descending, lower index first on ties) are fully determined;
//   N <= k  emit the candidates as they are (row-major, unsorted -- the reference returns them so);
//   N >  k  radix-select the k-th largest key (8 passes of 8 bits over the candidate list), gather
//           the k winners into LDS, bitonic-sort them descending.
// ------------------------------------------------------------------------------------------
#define SEL_THREADS 1024
#define SEL_MAXK 8192


__device__ __forceinline__ unsigned int block_exclusive_scan(unsigned int v, unsigned int* total, unsigned int* wsum) {
  // 1024 threads = 16 waves; returns exclusive prefix of v, *total = block sum
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned int n = __shfl_up(inc, o);
    if (lane >= o) inc += n;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned int base = 0, tot = 0;
  for (int i = 0; i < SEL_THREADS / 64; ++i) {
    unsigned int w = wsum[i];
    if (i < wave) base += w;
    tot += w;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const float* __restrict__ scores, int H, int W, float th,
                                                             int k, int cap, float* __restrict__ kpts,
                                                             float* __restrict__ kscores, int* __restrict__ counts,
                                                             unsigned long long* __restrict__ cand_all,
                                                             const int* __restrict__ precount) {
  __shared__ unsigned long long keys[SEL_MAXK];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int wsum[SEL_THREADS / 64];
  __shared__ unsigned int sh_cnt;
  __shared__ unsigned long long sh_prefix;
  __shared__ unsigned int sh_remaining;
  __shared__ unsigned int sh_done;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long HW = (long long)H * W;
  const float* sb = scores + (size_t)b * HW;
  unsigned long long* cand = cand_all + (size_t)b * HW;

  // ---- pass A: ordered compaction (skipped when the NMS kernel already emitted the candidates) ----
  unsigned int n = precount ? (unsigned int)precount[b] : 0;
  for (long long base = 0; !precount && base < HW; base += SEL_THREADS * 4) {
    long long i0 = base + (long long)tid * 4;
    float v[4];
    unsigned int flags = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long i = i0 + j;
      v[j] = (i < HW) ? sb[i] : -INFINITY;
      if (i < HW && v[j] > th) flags |= 1u << j;
    }
    unsigned int cnt = __popc(flags), tot;
    unsigned int off = n + block_exclusive_scan(cnt, &tot, wsum);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (flags & (1u << j)) {
        unsigned int idx = (unsigned int)(i0 + j);
        cand[off++] = ((unsigned long long)float_order_bits(v[j]) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
      }
    n += tot;
  }
  __syncthreads();

  float* kp = kpts + (size_t)b * cap * 2;
  float* ks = kscores + (size_t)b * cap;
  if (precount && n <= (unsigned int)k) {
    // candidates arrived unordered: restore the row-major order (ascending pixel index = descending low word)
    unsigned int P = 1;
    while (P < n) P <<= 1;
    for (unsigned int i = tid; i < P; i += SEL_THREADS) {
      const unsigned long long key = i < n ? cand[i] : 0ull;
      keys[i] = i < n ? ((key & 0xFFFFFFFFull) << 32) | (key >> 32) : 0ull;  // (~idx, score bits)
    }
    __syncthreads();
    for (unsigned int sz = 2; sz <= P; sz <<= 1)
      for (unsigned int st = sz >> 1; st > 0; st >>= 1) {
        for (unsigned int i = tid; i < P / 2; i += SEL_THREADS) {
          unsigned int lo = (i / st) * (st * 2) + (i % st), hi = lo + st;
          bool desc = ((lo & sz) == 0);
          unsigned long long a = keys[lo], c = keys[hi];
          if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
        }
        __syncthreads();
      }
    for (unsigned int i = tid; i < n; i += SEL_THREADS) {
      const unsigned long long key = keys[i];
      const unsigned int idx = 0xFFFFFFFFu - (unsigned int)(key >> 32);
      kp[2 * i] = (float)(idx % W);
      kp[2 * i + 1] = (float)(idx / W);
      ks[i] = float_from_order_bits((unsigned int)(key & 0xFFFFFFFFull));
    }
    for (unsigned int i = n + tid; i < (unsigned int)cap; i += SEL_THREADS) {
      kp[2 * i] = 0.f; kp[2 * i + 1] = 0.f; ks[i] = 0.f;
    }
    if (tid == 0) counts[b] = (int)n;
    return;
  }
  if (k < 0 || n <= (unsigned int)k) {
    // all candidates, row-major (unsorted)
    unsigned int cnt = min(n, (unsigned int)cap);
    for (unsigned int i = tid; i < cnt; i += SEL_THREADS) {
      unsigned long long key = cand[i];
      unsigned int idx = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
      kp[2 * i] = (float)(idx % W);
      kp[2 * i + 1] = (float)(idx / W);
      ks[i] = float_from_order_bits((unsigned int)(key >> 32));
    }
    for (unsigned int i = cnt + tid; i < (unsigned int)cap; i += SEL_THREADS) {
      kp[2 * i] = 0.f; kp[2 * i + 1] = 0.f; ks[i] = 0.f;
    }
    if (tid == 0) counts[b] = (int)cnt;
    return;
  }

  // ---- radix select: find the k-th largest key ----
  // Per pass: a 256-bin histogram of the next byte of the keys that still match the prefix, then the bin that holds the
  // k-th key -- found by 256 threads (inclusive scan over the bins in descending order; a serial walk by one thread
  // costs a dependent LDS read per bin, ~10 k cycles per pass).  When the bin's count equals what is still missing, ALL
  // its keys are winners: k-th = prefix with the lower bytes zero, and the remaining passes are skipped (the low word
  // is the unique pixel index, so this happens at the latest once the score bytes are through, unless scores tie).
  if (tid == 0) { sh_prefix = 0ull; sh_remaining = (unsigned int)k; sh_done = 0u; }
  __syncthreads();
  for (int pass = 7; pass >= 0; --pass) {
    const int shift = pass * 8;
    if (tid < 256) hist[tid] = 0;
    const unsigned long long prefix = sh_prefix;
    const unsigned int rem = sh_remaining;
    __syncthreads();
    const unsigned long long himask = (pass == 7) ? 0ull : (~0ull << (shift + 8));
    for (unsigned int i = tid; i < n; i += SEL_THREADS) {
      unsigned long long key = cand[i];
      if ((key & himask) == prefix) atomicAdd(&hist[(unsigned int)(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    unsigned int v = 0, inc = 0;
    if (tid < 256) {  // waves 0..3, whole waves
      v = hist[255 - tid];  // bins in descending order
      inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        unsigned int up = __shfl_up(inc, o);
        if ((tid & 63) >= o) inc += up;
      }
      if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    }
    __syncthreads();
    if (tid < 256) {
      for (int i = 0; i < (tid >> 6); ++i) inc += wsum[i];
      if (inc >= rem && inc - v < rem) {  // exactly one bin: the first (from the top) whose running count reaches rem
        sh_prefix = prefix | ((unsigned long long)(255 - tid) << shift);
        sh_remaining = rem - (inc - v);
        sh_done = (inc == rem) ? 1u : 0u;
      }
    }
    __syncthreads();
    if (sh_done) break;  // uniform
  }
  const unsigned long long kth = sh_prefix;  // exactly k keys are >= kth (keys are unique)
  if (tid == 0) sh_cnt = 0;
  __syncthreads();
  for (unsigned int i = tid; i < n; i += SEL_THREADS) {
    unsigned long long key = cand[i];
    if (key >= kth) {
      unsigned int slot = atomicAdd(&sh_cnt, 1u);
      if (slot < SEL_MAXK) keys[slot] = key;
    }
  }
  // pad to a power of two with zeros (smaller than any real key) and bitonic sort descending
  unsigned int P = 1;
  while (P < (unsigned int)k) P <<= 1;
  __syncthreads();
  for (unsigned int i = (unsigned int)k + tid; i < P; i += SEL_THREADS) keys[i] = 0ull;
  __syncthreads();
  for (unsigned int sz = 2; sz <= P; sz <<= 1) {
    for (unsigned int st = sz >> 1; st > 0; st >>= 1) {
      for (unsigned int i = tid; i < P / 2; i += SEL_THREADS) {
        unsigned int lo = (i / st) * (st * 2) + (i % st);
        unsigned int hi = lo + st;
        bool desc = ((lo & sz) == 0);
        unsigned long long a = keys[lo], c = keys[hi];
        if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (unsigned int i = tid; i < (unsigned int)k; i += SEL_THREADS) {
    unsigned long long key = keys[i];
    unsigned int idx = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
    kp[2 * i] = (float)(idx % W);
    kp[2 * i + 1] = (float)(idx / W);
    ks[i] = float_from_order_bits((unsigned int)(key >> 32));
  }
  for (unsigned int i = (unsigned int)k + tid; i < (unsigned int)cap; i += SEL_THREADS) {
    kp[2 * i] = 0.f; kp[2 * i + 1] = 0.f; ks[i] = 0.f;
  }
  if (tid == 0) counts[b] = k;
}

extern "C" size_t gfc_sp_select_workspace_bytes(int B, int H, int W) {
  return gfc_align((size_t)B * H * W * sizeof(unsigned long long));
}

extern "C" int gfc_sp_select(const float* scores, int B, int H, int W, float threshold, int k, int cap, float* kpts,
                             float* kscores, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
  if (!scores || !kpts || !kscores || !counts || !ws || B <= 0 || H <= 0 || W <= 0 || cap <= 0) return GFC_ERR_INVALID;
  if (k >= 0 && cap < k) return GFC_ERR_INVALID;
  if (k < 0 && (long long)cap < (long long)H * W) return GFC_ERR_INVALID;
  if (k > SEL_MAXK) return GFC_ERR_UNSUPPORTED;
  if (ws_bytes < gfc_sp_select_workspace_bytes(B, H, W)) return GFC_ERR_WORKSPACE;
  if (k == 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, scores, H, W, threshold, k, cap,
                     kpts, kscores, counts, reinterpret_cast<unsigned long long*>(ws), (const int*)nullptr);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// Fused NMS + selection: the NMS kernel appends every pixel above the threshold to the per-image key list
// (one atomic per wave), so the selection kernel skips its scan of the dense map, and the suppressed map itself
// is only written when the caller asks for it (nms_out != NULL).  Needs a finite k (1..8192).
extern "C" size_t gfc_sp_nms_select_workspace_bytes(int B, int H, int W) {
  return gfc_align((size_t)B * sizeof(int)) + gfc_sp_select_workspace_bytes(B, H, W);
}

extern "C" int gfc_sp_nms_select(const float* heatmap, int B, int H, int W, int radius, int border,
                                 const int32_t* valid_wh, float threshold, int k, int cap, float* nms_out, float* kpts,
                                 float* kscores, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
  if (!heatmap || !kpts || !kscores || !counts || !ws || B <= 0 || H <= 0 || W <= 0 || radius < 0 || border < 0)
    return GFC_ERR_INVALID;
  if (k <= 0 || cap < k) return GFC_ERR_INVALID;
  // radius 0 keeps every pixel above the threshold (10^5 candidates per image): use the two-stage path there
  if (radius < 1 || radius > NR_MAX || k > SEL_MAXK) return GFC_ERR_UNSUPPORTED;
  if (ws_bytes < gfc_sp_nms_select_workspace_bytes(B, H, W)) return GFC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = (int*)ws;
  unsigned long long* cand = (unsigned long long*)((char*)ws + gfc_align((size_t)B * sizeof(int)));
  if (hipMemsetAsync(cnt, 0, (size_t)B * sizeof(int), st) != hipSuccess) return GFC_ERR_LAUNCH;
  int s = nms_dispatch(heatmap, B, H, W, radius, border, valid_wh, nms_out, threshold, cand, cnt, st);
  if (s != GFC_OK) return s;
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_THREADS), 0, st, (const float*)nullptr, H, W, threshold, k, cap,
                     kpts, kscores, counts, cand, (const int*)cnt);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ------------------------------------------------------------------------------------------
// Descriptor sampling: grid_sample(bilinear, zeros padding) of the L2-normalised dense map at the
// key points, then L2 normalisation (superpoint_open.py:22-33; superpoint.py:120-152).
// One wave per key point, 4 channels per lane (D = 256).  The dense map is stored un-normalised;
// each corner is normalised on the fly (x / max(||x||, 1e-12), F.normalize), which removes a
// full pass over the map.
// ------------------------------------------------------------------------------------------
template <int VPL>  // float4 per lane: D = 256 * VPL
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ dense, int B, int h, int w,
                                                     const float* __restrict__ kpts, const int* __restrict__ n_kpts,
                                                     int cap, int mode, float* __restrict__ out,
                                                     float* __restrict__ kpts_out) {
  const int D = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long long)B * cap) return;
  const int b = (int)(wid / cap), i = (int)(wid % cap);
  const int n = n_kpts ? n_kpts[b] : cap;
  float* o = out + (size_t)wid * D;
  const float kx = kpts[2 * wid], ky = kpts[2 * wid + 1];
  if (i >= n) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) *reinterpret_cast<float4*>(o + (v * 64 + lane) * 4) = make_float4(0, 0, 0, 0);
    return;
  }
  // normalised grid coordinate, op by op as the reference computes it in fp32
  float gx, gy, ix, iy;
  const float s = 8.f;
  if (mode == GFC_SAMPLE_OPEN) {
    gx = (kx + 0.5f) / ((float)w * s);
    gy = (ky + 0.5f) / ((float)h * s);
  } else if (mode == GFC_SAMPLE_LEGACY) {
    gx = ((kx - 4.f) + 0.5f) / (float)((double)w * 8.0 - 4.0 - 0.5);
    gy = ((ky - 4.f) + 0.5f) / (float)((double)h * 8.0 - 4.0 - 0.5);
  } else {
    gx = kx / ((float)w * s);
    gy = ky / ((float)h * s);
  }
  gx = gx * 2.f - 1.f;
  gy = gy * 2.f - 1.f;
  if (mode == GFC_SAMPLE_LEGACY) {  // align_corners=True
    ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
    iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
  } else {
    ix = ((gx + 1.f) * (float)w - 1.f) / 2.f;
    iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
  }
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
  const float cw[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};  // nw, ne, sw, se
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0, 0, 0, 0);
  const float* db = dense + (size_t)b * h * w * D;
  // the four corners are requested together (clamped addresses; a corner outside the map is skipped below): one
  // memory round trip per key point instead of four dependent ones
  float4 cval[4][VPL];
  bool inside[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
    inside[c] = !(xx < 0 || xx >= w || yy < 0 || yy >= h);  // zeros padding (wave-uniform)
    const float* p = db + ((size_t)min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)) * D;
#pragma unroll
    for (int v = 0; v < VPL; ++v) cval[c][v] = *reinterpret_cast<const float4*>(p + (v * 64 + lane) * 4);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (!inside[c]) continue;
    float4 val[VPL];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      val[v] = cval[c][v];
      ss += val[v].x * val[v].x + val[v].y * val[v].y + val[v].z * val[v].z + val[v].w * val[v].w;
    }
    ss = wave_sum(ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      acc[v].x += (val[v].x / den) * cw[c];
      acc[v].y += (val[v].y / den) * cw[c];
      acc[v].z += (val[v].z / den) * cw[c];
      acc[v].w += (val[v].w / den) * cw[c];
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) ss += acc[v].x * acc[v].x + acc[v].y * acc[v].y + acc[v].z * acc[v].z + acc[v].w * acc[v].w;
  ss = wave_sum(ss);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
  for (int v = 0; v < VPL; ++v)
    *reinterpret_cast<float4*>(o + (v * 64 + lane) * 4) =
        make_float4(acc[v].x / den, acc[v].y / den, acc[v].z / den, acc[v].w / den);
  if (kpts_out && lane == 0) {
    kpts_out[2 * wid] = kx + 0.5f;
    kpts_out[2 * wid + 1] = ky + 0.5f;
  }
}

extern "C" int gfc_sp_sample(const float* desc_raw, int B, int h, int w, int D, const float* kpts,
                             const int32_t* n_kpts, int cap, int mode, float* out, float* kpts_out, void* stream) {
  if (!desc_raw || !kpts || !out || B <= 0 || h <= 0 || w <= 0 || cap <= 0) return GFC_ERR_INVALID;
  if (mode < 0 || mode > 2) return GFC_ERR_INVALID;
  if (D != 256) return GFC_ERR_UNSUPPORTED;
  long long waves = (long long)B * cap;
  hipLaunchKernelGGL(sample_kernel<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, desc_raw, B,
                     h, w, kpts, n_kpts, cap, mode, out, kpts_out);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, long long rows, int width) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = x + (size_t)row * width;
  float ss = 0.f;
  for (int c = lane; c < width; c += 64) ss += p[c] * p[c];
  ss = wave_sum(ss);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  for (int c = lane; c < width; c += 64) p[c] = p[c] / den;
}

extern "C" int gfc_l2norm_rows(float* x, long long rows, int width, void* stream) {
  if (!x || rows <= 0 || width <= 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows,
                     width);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ---- specular-mask filtering of key points (the Endomapper addition of this reference) --------------------------
// Replaces filter_keypoints_by_specular_mask (reference gluefactory/models/extractors/utils.py:4-42): a key point
// survives when the pixels floor/ceil(kp - offset) lie inside the mask (cropped to image_size when given) and the
// mask is set on all of them.  Two call orders exist in the reference:
//   * superpoint_open.py:177-188 filters the candidates BEFORE the top-k selection  -> gfc_sp_mask_scores on the
//     suppressed score map (integer candidates: floor == ceil, one mask pixel per candidate), then gfc_sp_select;
//   * gluefactory_nonfree/superpoint.py:310-328 filters the selected key points AFTER top-k -> gfc_sp_filter_keypoints
//     (stable in-place compaction, counts updated).
__global__ __launch_bounds__(256) void mask_scores_kernel(float* __restrict__ scores, const unsigned char* __restrict__ mask,
                                                          const int* __restrict__ wh, int H, int W, int Hm, int Wm) {
  const int b = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int eh = Hm, ew = Wm;
  if (wh) { ew = min(ew, wh[2 * b]); eh = min(eh, wh[2 * b + 1]); }  // mask[..., :h, :w]
  const bool keep = y < eh && x < ew && mask[((size_t)b * Hm + y) * Wm + x] != 0;
  if (!keep) scores[((size_t)b * H + y) * W + x] = -INFINITY;  // below every detection threshold
}

extern "C" int gfc_sp_mask_scores(float* scores, int B, int H, int W, const uint8_t* mask, int Hm, int Wm,
                                  const int32_t* image_wh, void* stream) {
  if (!scores || !mask || B <= 0 || H <= 0 || W <= 0 || Hm <= 0 || Wm <= 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(mask_scores_kernel, dim3((W + 255) / 256, H, B), dim3(256), 0, (hipStream_t)stream, scores, mask,
                     image_wh, H, W, Hm, Wm);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

#define FK_THREADS 1024
__global__ __launch_bounds__(FK_THREADS) void filter_keypoints_kernel(float* __restrict__ kpts, float* __restrict__ kscores,
                                                                      int* __restrict__ counts, int cap,
                                                                      const unsigned char* __restrict__ mask, int Hm, int Wm,
                                                                      const int* __restrict__ wh, float offset) {
  __shared__ int wave_tot[FK_THREADS / 64];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* kp = kpts + (size_t)b * cap * 2;
  float* ks = kscores + (size_t)b * cap;
  const unsigned char* m = mask + (size_t)b * Hm * Wm;
  int eh = Hm, ew = Wm;
  if (wh) { ew = min(ew, wh[2 * b]); eh = min(eh, wh[2 * b + 1]); }
  const int n = min(counts[b], cap);
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int start = 0; start < n; start += FK_THREADS) {
    const int i = start + tid;
    float x = 0.f, y = 0.f, s = 0.f;
    bool keep = false;
    if (i < n) {
      x = kp[2 * i]; y = kp[2 * i + 1]; s = ks[i];
      const float fx = x - offset, fy = y - offset;
      const int x0 = (int)floorf(fx), x1 = (int)ceilf(fx), y0 = (int)floorf(fy), y1 = (int)ceilf(fy);
      if (x0 >= 0 && x1 < ew && y0 >= 0 && y1 < eh)
        keep = m[(size_t)y0 * Wm + x0] && m[(size_t)y0 * Wm + x1] && m[(size_t)y1 * Wm + x0] && m[(size_t)y1 * Wm + x1];
    }
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();  // all loads of this chunk are done, wave totals visible
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (keep) { kp[2 * (off + before)] = x; kp[2 * (off + before) + 1] = y; ks[off + before] = s; }
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < FK_THREADS / 64; ++w) t += wave_tot[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (tid == 0) counts[b] = base_s;
}

extern "C" int gfc_sp_filter_keypoints(float* kpts, float* kscores, int32_t* counts, int B, int cap, const uint8_t* mask,
                                       int Hm, int Wm, const int32_t* image_wh, float keypoint_offset, void* stream) {
  if (!kpts || !kscores || !counts || !mask || B <= 0 || cap <= 0 || Hm <= 0 || Wm <= 0) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(filter_keypoints_kernel, dim3(B), dim3(FK_THREADS), 0, (hipStream_t)stream, kpts, kscores,
                     (int*)counts, cap, mask, Hm, Wm, image_wh, keypoint_offset);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

// ---- soft-argmax refinement of the selected key points (official variant only) -----------------------------------
// Replaces soft_argmax_refinement (reference gluefactory_nonfree/superpoint.py:100-116, applied at :302-305 after the
// top-k selection): kp += (sum dy*s, sum dx*s) / sum s over the (2r+1)^2 window of the DENSE score map (before NMS),
// zero outside the map.  The reference builds three dense maps with avg_pool2d / conv2d and gathers at the key points;
// here each key point evaluates its own window (K << H*W).
__global__ __launch_bounds__(256) void refine_keypoints_kernel(const float* __restrict__ heat, int H, int W,
                                                               float* __restrict__ kpts, const int* __restrict__ counts,
                                                               int cap, int radius) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = counts ? min(counts[b], cap) : cap;
  if (i >= n) return;
  float* kp = kpts + ((size_t)b * cap + i) * 2;
  const int x = (int)kp[0], y = (int)kp[1];
  const float* hm = heat + (size_t)b * H * W;
  float sum = 0.f, sx = 0.f, sy = 0.f;
  for (int dy = -radius; dy <= radius; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -radius; dx <= radius; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const float s = hm[(size_t)yy * W + xx];
      sum += s;
      sx += (float)dx * s;
      sy += (float)dy * s;
    }
  }
  kp[0] = (float)x + sx / sum;
  kp[1] = (float)y + sy / sum;
}

extern "C" int gfc_sp_refine_keypoints(const float* heatmap, int B, int H, int W, float* kpts, const int32_t* counts,
                                       int cap, int radius, void* stream) {
  if (!heatmap || !kpts || B <= 0 || H <= 0 || W <= 0 || cap <= 0 || radius < 1) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(refine_keypoints_kernel, dim3((cap + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, heatmap, H,
                     W, kpts, (const int*)counts, cap, radius);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}


// ------------------------------------------------------------------------------------------------------------
// pad_and_stack(..., mode="random_c") + zeros for the scores (gluefactory/models/utils/misc.py:19-62,103-113, called
// with force_num_keypoints at superpoint_open.py:193-219 / superpoint.py:330-365 / disk_kornia.py:109-124): slots at
// or beyond an image's count get, per coordinate, a uniform sample in [min, max] of the image's own key points (the
// fallback bounds [low, high] when it has none); their scores become 0.  One workgroup per image, in place.
// The samples come from a counter-based generator keyed by (seed, image, slot, coordinate): the reference draws
// them from torch's CPU generator, so the values differ by construction -- they are random padding there too.
// high = min over `sizes` (n_sizes floats: data["image_size"].min()) when given, else `high_fallback`.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pad_uniform(unsigned int a, unsigned int b, unsigned int c, unsigned int d) {
  unsigned int x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du ^ d * 0x27D4EB2Fu;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;  // lowbias32-style finaliser
  return (float)(x >> 8) * (1.0f / 16777216.0f);                                 // [0, 1)
}

__global__ __launch_bounds__(256) void pad_keypoints_kernel(float* __restrict__ kpts, float* __restrict__ kscores,
                                                            const int* __restrict__ counts, int cap, int k, float low,
                                                            const float* __restrict__ sizes, int n_sizes,
                                                            float high_fallback, unsigned int seed) {
  __shared__ float red[4][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* kp = kpts + (size_t)b * cap * 2;
  float* ks = kscores + (size_t)b * cap;
  const int n = min(counts[b], k);
  if (n >= k) return;  // nothing to pad (workgroup-uniform)
  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
  for (int i = tid; i < n; i += 256) {
    const float x = kp[2 * i], y = kp[2 * i + 1];
    mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o)); mny = fminf(mny, __shfl_xor(mny, o));
    mxx = fmaxf(mxx, __shfl_xor(mxx, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
  }
  if (lane == 0) { red[wave][0] = mnx; red[wave][1] = mny; red[wave][2] = mxx; red[wave][3] = mxy; }
  __syncthreads();
  mnx = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
  mny = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1]));
  mxx = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
  mxy = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
  if (n == 0) {  // the bounds are the fallback for an empty sequence (misc.py:47-49)
    float high = high_fallback;
    if (sizes) {
      high = INFINITY;
      for (int i = 0; i < n_sizes; ++i) high = fminf(high, sizes[i]);
    }
    mnx = mny = low;
    mxx = mxy = high;
  }
  for (int i = n + tid; i < k; i += 256) {
    kp[2 * i] = mnx + pad_uniform(seed, (unsigned)b, (unsigned)i, 0u) * (mxx - mnx);
    kp[2 * i + 1] = mny + pad_uniform(seed, (unsigned)b, (unsigned)i, 1u) * (mxy - mny);
    ks[i] = 0.f;
  }
}

extern "C" int gfc_sp_pad_keypoints(float* kpts, float* kscores, const int32_t* counts, int B, int cap, int k, float low,
                                    const float* sizes, int n_sizes, float high_fallback, unsigned int seed,
                                    void* stream) {
  if (!kpts || !kscores || !counts || B <= 0 || k <= 0 || cap < k || (sizes && n_sizes <= 0)) return GFC_ERR_INVALID;
  hipLaunchKernelGGL(pad_keypoints_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, kpts, kscores, counts, cap, k, low,
                     sizes, n_sizes, high_fallback, seed);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}
