// fp32-MFMA "NT" GEMM  Y[M,N] = [A0|A1][M,K] * W[N,K]^T  with fused epilogues.
//
// Replaces every nn.Linear on the LightGlue path (reference
// gluefactory/models/matchers/lightglue.py:139-148,158,163-164,181-189,276-288: F.linear -> addmm),
// the 1x1 convolutions of the SuperPoint heads (superpoint_open.py:112-118, NHWC makes them
// plain GEMMs) and, in batched form, einsum("bmd,bnd->bmn") (lightglue.py:285).
//
// Workgroup = 4 waves = 128x128 output tile (64x64 for small problems), K stepped by 16 (or 32) through
// double-buffered LDS (row stride K tile + 4 floats: conflict-free ds_read_b128 of 4 consecutive k per lane).
// Each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32 tiles; per 8-deep k group it issues 4 LDS reads
// and 16 v_mfma_f32_32x32x2_f32.  The next K tile is prefetched global->registers while the current
// one is multiplied; one barrier per K tile.
#include <stdlib.h>

#include "common.h"

#define GBK 32  // K granularity of the API (every variant's K tile divides it)

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
  const float* rot_cs;  // alternative to rot_cos / rot_sin: packed [M][32][cos, sin] (library-internal, gfc_linear_rot_packed)
  float* Y;
  long long strideA, strideW, strideY;  // batch strides (blockIdx.z)
  int lda0, lda1, ldw, ldy;
  int K0, K1, M, N, rot_cols;
  float alpha;
  int xcd_remap;  // gemm_nt_kernel: walk the tiles in XCD-contiguous order (runtime.h: GFC_XCD_REMAP)
  int wide_stores;  // every epilogue through the per-wave LDS transpose: float4 stores (runtime.h: GFC_GEMM_EPI)
  int stagger;      // first-round workgroups start (wave slot & 3) * stagger * 8128 cycles late (GFC_GEMM_STAGGER)
  int first_round;  // number of workgroups resident at once (4 per CU)
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  unsigned long long* stamps;  // diagnostic build (tools/micro/gemm_timeline.py): 8 words per wave
#endif
};
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
static unsigned long long* g_diag_stamps = nullptr;
extern "C" void gfc_diag_set_gemm_stamps(void* p) { g_diag_stamps = (unsigned long long*)p; }
#define GEMM_STAMP(i_)                                                                               \
  do {                                                                                               \
    if (g.stamps && lane == 0) stamp_base[i_] = __builtin_readcyclecounter();                        \
  } while (0)
#else
#define GEMM_STAMP(i_) do {} while (0)
#endif

// Fast path of the transposed epilogue for one 32-row round of a wave (whole patch inside the matrix, float4-aligned),
// specialised by MODE (0 = bias / affine only, 1 = rotary, 2 = residual) so that the body is ONE basic block and the
// wait counts hipcc inserts are exact.  The rotary / residual operands of step i + 1 are requested BEFORE the store of
// step i: vmcnt retires in order and counts stores too, so a load issued behind a store cannot be waited for without
// waiting for that store's acknowledgement (the per-step form -- load, wait, store, load, wait -- serialises 16 store
// round trips per wave: 46 k cycles, tools/micro/gemm_timeline.py).
template <int MODE, int NS, int RPS, int ELD>
__device__ __forceinline__ void gemm_epilogue_round_fast(const GemmArgs& g, const float* patch, float* Y, int row0, int er,
                                                         int ec, int colb, int rd, float4 bi4, float4 sc4, float4 sh4) {
  float4 opa = make_float4(0.f, 0.f, 0.f, 0.f), opb = opa;
  // ONE running pointer per stream, bumped by a constant stride per step: with per-step 64-bit address arithmetic
  // (row0 + RPS * i) * ld the unrolled loop kept an address pair per step alive and the 128-register kernel
  // (gemm_nt_kernel<2,2,16,2>, four workgroups per CU) spilled two of them to scratch
  const float* pa = nullptr;
  const float* pb = nullptr;
  size_t pstep = (size_t)RPS * 64;
  if constexpr (MODE == 1) {
    pa = g.rot_cos + (size_t)row0 * 64 + rd;
    pb = g.rot_sin + (size_t)row0 * 64 + rd;
    opa = *reinterpret_cast<const float4*>(pa);
    opb = *reinterpret_cast<const float4*>(pb);
  } else if constexpr (MODE == 2) {
    pa = g.residual + (size_t)row0 * g.ldy + colb;
    pstep = (size_t)RPS * g.ldy;
    opa = *reinterpret_cast<const float4*>(pa);
  } else if constexpr (MODE == 3) {
    pa = g.rot_cs + (size_t)row0 * 64 + rd;
    opa = *reinterpret_cast<const float4*>(pa);
  }
  float* yp = Y + (size_t)row0 * g.ldy + colb;
  const size_t ystep = (size_t)RPS * g.ldy;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    float4 na = opa, nb = opb;
    if (i + 1 < NS) {
      if constexpr (MODE != 0) {
        pa += pstep;
        na = *reinterpret_cast<const float4*>(pa);
      }
      if constexpr (MODE == 1) {
        pb += pstep;
        nb = *reinterpret_cast<const float4*>(pb);
      }
    }
    float4 v = *reinterpret_cast<const float4*>(patch + (er + RPS * i) * ELD + ec);
    v.x += bi4.x; v.y += bi4.y; v.z += bi4.z; v.w += bi4.w;
    if constexpr (MODE == 1) {
      // rotary: out[d] = t[d]*cos[d] + rot(t)[d]*sin[d], rot(t)[2i] = -t[2i+1], rot(t)[2i+1] = t[2i]
      const float x = v.x, y = v.y, zz = v.z, w = v.w;
      v.x = x * opa.x + (-y) * opb.x;
      v.y = y * opa.y + x * opb.y;
      v.z = zz * opa.z + (-w) * opb.z;
      v.w = w * opa.w + zz * opb.w;
    }
    if constexpr (MODE == 3) {  // opa = (cos f, sin f, cos f+1, sin f+1): the same products as MODE 1
      const float x = v.x, y = v.y, zz = v.z, w = v.w;
      v.x = x * opa.x + (-y) * opa.y;
      v.y = y * opa.x + x * opa.y;
      v.z = zz * opa.z + (-w) * opa.w;
      v.w = w * opa.z + zz * opa.w;
    }
    v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
    v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha;
    if constexpr (MODE == 2) { v.x = opa.x + v.x; v.y = opa.y + v.y; v.z = opa.z + v.z; v.w = opa.w + v.w; }
    *reinterpret_cast<float4*>(yp) = v;
    yp += ystep;
    opa = na; opb = nb;
  }
}

// Epilogue shared by the GEMM kernels: bias / BN affine / alpha directly from the accumulator layout, or -- when
// rotary or residual operands have to be loaded per element -- through a per-wave LDS transpose with float4 traffic.
// Called after the K loop's final workgroup barrier (smem is free).
template <int NW, int MT, int MTN = MT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[MT][MTN], float* smem, float* Y, int m0,
                                              int n0, int wm, int wn, int lane, int wave) {
  constexpr int WT = 32 * MTN;   // columns per wave
  constexpr int WTM = 32 * MT;   // rows per wave
  const int l31 = lane & 31, h = lane >> 5;
  // (workgroup-uniform: column tiles at or beyond rot_cols -- the V third of the fused QKV projection -- carry no
  // rotary and take the direct path as well)
  if (((g.rot_cos == nullptr && g.rot_cs == nullptr) || n0 >= g.rot_cols) && g.residual == nullptr && !g.wide_stores) {
    // plain epilogue: straight from the accumulator layout (column on the lane, rows in registers);
    // measured faster than the LDS transpose below when nothing has to be loaded per element
#pragma unroll
    for (int nt = 0; nt < MTN; ++nt) {
      const int col = n0 + wn * WT + nt * 32 + l31;
      const bool col_ok = col < g.N;
      const int cc = col_ok ? col : g.N - 1;
      float bi = g.bias ? g.bias[cc] : 0.f;
      float sc = g.scale ? g.scale[cc] : 1.f;
      float sh = g.shift ? g.shift[cc] : 0.f;
      // The three loads must be WAITED FOR here, once, in straight-line code.  Left to hipcc, their first use sits
      // inside the first predicated store block, the wait-count pass cannot prove at the block joins that it has
      // happened, and it puts `s_waitcnt vmcnt(0)` in front of every one of the 64 stores -- which also waits for the
      // previous STORE to be acknowledged: 64 serialized round trips, 46 k cycles per wave (tools/micro/gemm_timeline.py).
      asm volatile("" : "+v"(bi), "+v"(sc), "+v"(sh));
      const bool full = m0 + wm * WTM + MT * 32 <= g.M && n0 + wn * WT + MTN * 32 <= g.N;  // wave-uniform
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row0 = m0 + wm * WTM + mt * 32;
        if (full) {  // no predicates: one basic block, stores issue back to back
          float* yp = Y + (size_t)(row0 + 4 * h) * g.ldy + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = ((acc[mt][nt][r] + bi) * sc + sh) * g.alpha;
#if defined(GEMM_DIAG) && (GEMM_DIAG & 1)
            if (v == 12345.678f)  // diagnostic: no stores
#endif
            yp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldy] = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + acc_row(r, h);
            const float v = ((acc[mt][nt][r] + bi) * sc + sh) * g.alpha;
#if defined(GEMM_DIAG) && (GEMM_DIAG & 1)
            if (row < g.M && col_ok && v == 12345.678f) Y[(size_t)row * g.ldy + col] = v;  // diagnostic: no stores
#else
            if (row < g.M && col_ok) Y[(size_t)row * g.ldy + col] = v;
#endif
          }
        }
      }
    }
    return;
  }
  // Rotary / residual epilogues load per element: in the accumulator layout that is one float per
  // lane per instruction.  Each wave transposes its tile through its own LDS patch instead (the
  // K-loop buffers are free now) and handles whole float4 row segments: 4x fewer loads and stores.
  constexpr int ELD = WT + 4;        // patch row stride (floats): 32 rows x WT cols per round
  constexpr int LPR = WT / 4;        // lanes per patch row (one float4 each)
  constexpr int RPS = 64 / LPR;      // rows per step
  float* patch = smem + wave * 32 * ELD;
  const bool vec_ok = (g.ldy % 4 == 0) && ((reinterpret_cast<size_t>(Y) & 15) == 0) &&
                      (!g.residual || (reinterpret_cast<size_t>(g.residual) & 15) == 0);
  const int er = lane / LPR, ec = (lane % LPR) * 4;  // this lane's row (+RPS per step) and 4 columns
  const int colb = n0 + wn * WT + ec;
  float4 bi4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = bi4;
  {
    float* bp4 = reinterpret_cast<float*>(&bi4);
    float* sp4 = reinterpret_cast<float*>(&sc4);
    float* hp4 = reinterpret_cast<float*>(&sh4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = min(colb + j, g.N - 1);
      if (g.bias) bp4[j] = g.bias[cc];
      if (g.scale) { sp4[j] = g.scale[cc]; hp4[j] = g.shift[cc]; }
    }
  }
  const bool rot = (g.rot_cos != nullptr || g.rot_cs != nullptr) && colb < g.rot_cols;
  const int rd = colb & 63;
  // as in the direct path: the per-column operands are waited for here, once, outside the store blocks
  asm volatile("" : "+v"(bi4.x), "+v"(bi4.y), "+v"(bi4.z), "+v"(bi4.w));
  asm volatile("" : "+v"(sc4.x), "+v"(sc4.y), "+v"(sc4.z), "+v"(sc4.w));
  asm volatile("" : "+v"(sh4.x), "+v"(sh4.y), "+v"(sh4.z), "+v"(sh4.w));
  // wave-uniform fast path (whole 32 x WT patch inside the matrix, float4-aligned): the rotary / residual operands of
  // ALL steps of a round are requested first, then the steps compute and store back to back.  vmcnt retires in order
  // and counts stores too: a load issued behind a store cannot be waited for without waiting for that store's
  // acknowledgement, so the per-step form (load, wait, store, load, wait, ...) serialises 16 store round trips per wave.
  const bool rot_w = (g.rot_cos != nullptr || g.rot_cs != nullptr) && n0 + wn * WT < g.rot_cols;  // rot_cols % 64 == 0
  const bool fast = vec_ok && m0 + wm * WTM + MT * 32 <= g.M && n0 + wn * WT + WT <= g.N && !(rot_w && g.residual);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (fast) {
      constexpr int NS = 32 / RPS;
      const int row0 = m0 + wm * WTM + mt * 32 + er;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int nt = 0; nt < MTN; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[acc_row(r, h) * ELD + nt * 32 + l31] = acc[mt][nt][r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (rot_w && g.rot_cs) gemm_epilogue_round_fast<3, NS, RPS, ELD>(g, patch, Y, row0, er, ec, colb, rd, bi4, sc4, sh4);
      else if (rot_w) gemm_epilogue_round_fast<1, NS, RPS, ELD>(g, patch, Y, row0, er, ec, colb, rd, bi4, sc4, sh4);
      else if (g.residual) gemm_epilogue_round_fast<2, NS, RPS, ELD>(g, patch, Y, row0, er, ec, colb, rd, bi4, sc4, sh4);
      else gemm_epilogue_round_fast<0, NS, RPS, ELD>(g, patch, Y, row0, er, ec, colb, rd, bi4, sc4, sh4);
      continue;
    }
    // The patch is private to the wave and the K loop ended on a workgroup barrier: LDS operations of one wave
    // complete in order, so only the compiler has to be kept from reordering across the transpose.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < MTN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[acc_row(r, h) * ELD + nt * 32 + l31] = acc[mt][nt][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 32 / RPS; ++i) {
      const int lr = er + RPS * i;
      const int row = m0 + wm * WTM + mt * 32 + lr;
      float4 v = *reinterpret_cast<const float4*>(patch + lr * ELD + ec);
      if (row >= g.M) continue;
      v.x += bi4.x; v.y += bi4.y; v.z += bi4.z; v.w += bi4.w;
      if (rot) {
        // rotary: out[d] = t[d]*cos[d] + rot(t)[d]*sin[d], rot(t)[2i] = -t[2i+1], rot(t)[2i+1] = t[2i]
        float4 c, sn;
        if (g.rot_cs) {
          const float4 cs = *reinterpret_cast<const float4*>(g.rot_cs + (size_t)row * 64 + rd);
          c = make_float4(cs.x, cs.x, cs.z, cs.z);
          sn = make_float4(cs.y, cs.y, cs.w, cs.w);
        } else {
          c = *reinterpret_cast<const float4*>(g.rot_cos + (size_t)row * 64 + rd);
          sn = *reinterpret_cast<const float4*>(g.rot_sin + (size_t)row * 64 + rd);
        }
        const float x = v.x, y = v.y, zz = v.z, w = v.w;
        v.x = x * c.x + (-y) * sn.x;
        v.y = y * c.y + x * sn.y;
        v.z = zz * c.z + (-w) * sn.z;
        v.w = w * c.w + zz * sn.w;
      }
      v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
      v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha;
      const size_t o = (size_t)row * g.ldy + colb;
      if (vec_ok && colb + 3 < g.N) {
        if (g.residual) {
          const float4 rs = *reinterpret_cast<const float4*>(g.residual + o);
          v.x = rs.x + v.x; v.y = rs.y + v.y; v.z = rs.z + v.z; v.w = rs.w + v.w;
        }
        *reinterpret_cast<float4*>(Y + o) = v;
      } else {
        const float* vp = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (colb + j < g.N) Y[o + j] = g.residual ? g.residual[o + j] + vp[j] : vp[j];
      }
    }
  }
}

// 2 x NW waves; every wave owns MT x MT MFMA tiles of 32x32.
//   MT = 2, NW = 2, BK = 16: 128x128 tile, 256 threads, 40 KB LDS (4 workgroups / CU)    default for large problems
//   MT = 1, NW = 2, BK = 32:  64x64  tile, 256 threads,  36 KB LDS (4 workgroups / CU)   small M (batch 1..4):
//                    4x the workgroups, so that a [2048, 256] GEMM still covers the chip
// LDS is double-buffered: one barrier per K tile, the next tile travels global -> VGPR -> LDS
// underneath the MFMAs of the current one.
template <int NW, int MT, int BK, int MTN = MT>
__global__ __launch_bounds__(128 * NW, MT * MTN > 4 ? 2 : (BK == 16 ? 4 : 2)) void gemm_nt_kernel(GemmArgs g) {
  constexpr int T = 128 * NW;        // threads
  constexpr int GBM = 64 * MT;       // tile height (2 waves)
  constexpr int BN = 32 * MTN * NW;  // tile width
  constexpr int GLD = BK + 4;        // LDS row stride (floats)
  constexpr int C4 = BK / 4;         // float4 per staged row
  constexpr int RPP = T / C4;        // rows staged per pass
  constexpr int NA = GBM / RPP;      // float4 of A per thread per K tile
  constexpr int NB = BN / RPP;       // float4 of W per thread per K tile
  constexpr int TILE = (GBM + BN) * GLD;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / NW, wn = wave % NW;
  // XCD-aware tile order (column tile fastest): the gridDim.x column tiles of a row panel run at the same time behind
  // ONE L2, so the panel of A is fetched from HBM once instead of once per column tile
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.xcd_remap) {
    const unsigned per_z = gridDim.x * gridDim.y;
    unsigned t = gfc_xcd_chunk(bx + gridDim.x * (by + gridDim.y * bz), per_z * gridDim.z);
    bz = t / per_z;
    t -= bz * per_z;
    by = t / gridDim.x;
    bx = t - by * gridDim.x;
  }
  // De-phasing (K = 256 GEMMs are two lock-step rounds of four workgroups per CU: every CU stores its 64 KB tiles at
  // the same moment, the chip-wide burst is HBM-write bound and the matrix pipe waits): the workgroups of the FIRST
  // round start staggered by their wave slot, so that later K loops and epilogues of a CU interleave.
  if (g.stagger > 0 && (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) < g.first_round) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int naps = (int)(hw & 3u) * g.stagger;
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const int m0 = by * GBM, n0 = bx * BN;
  const long long z = bz;
  const float* A0 = g.A0 + z * g.strideA;
  const float* A1 = g.A1 ? g.A1 + z * g.strideA : nullptr;
  const float* W = g.W + z * g.strideW;
  float* Y = g.Y + z * g.strideY;
  const int ktiles = (g.K0 + g.K1) / BK;
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  unsigned long long* stamp_base =
      g.stamps ? g.stamps + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (T / 64) + wave) * 8
               : nullptr;
  if (g.stamps && lane == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamp_base[5] = hw;
    stamp_base[6] = xcc;
  }
  GEMM_STAMP(0);
#endif

  const int s_c4 = (tid % C4) * 4;
  // staged row of this thread.  With a 16-deep K tile a ds_write_b128 lane group (8 consecutive lanes) covers two
  // rows; at a row pitch of 20 floats adjacent rows overlap by 4 of the 32 banks (2-way conflict, PMC: 33 % of the
  // LDS cycles), rows FOUR apart do not (20*4 = 80 = 16 mod 32): permute the rows inside each block of 8.
  const int q_ = tid / C4;
  const int s_r0 = C4 == 4 ? ((q_ & ~7) | ((q_ & 1) << 2) | ((q_ >> 1) & 3)) : q_;
  // staged rows (clamped; out-of-range rows are never stored by the epilogue).  Named registers and
  // wave-uniform pointer selection: arrays / per-branch loads were demoted to private memory by hipcc.
  const int ar0 = min(m0 + s_r0, g.M - 1), ar1 = min(m0 + s_r0 + RPP, g.M - 1);
  const int ar2 = min(m0 + s_r0 + 2 * RPP, g.M - 1), ar3 = min(m0 + s_r0 + 3 * RPP, g.M - 1);
  const float* w0p = W + (size_t)min(n0 + s_r0, g.N - 1) * g.ldw + s_c4;
  const float* w1p = W + (size_t)min(n0 + s_r0 + RPP, g.N - 1) * g.ldw + s_c4;
  const float* w2p = W + (size_t)min(n0 + s_r0 + 2 * RPP, g.N - 1) * g.ldw + s_c4;
  const float* w3p = W + (size_t)min(n0 + s_r0 + 3 * RPP, g.N - 1) * g.ldw + s_c4;
  static_assert((NB == 1 || NB == 2 || NB == 4) && (NA == 1 || NA == 2 || NA == 4), "staging layout");
  float4 areg0, areg1, areg2, areg3, wreg0, wreg1, wreg2, wreg3;
  areg1 = wreg1 = areg2 = areg3 = wreg2 = wreg3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define GEMM_LOAD_TILE(kt)                                                                           \
  do {                                                                                               \
    const int k0_ = (kt) * BK;                                                                       \
    const bool first_ = k0_ < g.K0;                                                                  \
    const float* ab_ = (first_ ? A0 : A1) + (first_ ? k0_ : k0_ - g.K0) + s_c4;                      \
    const size_t ld_ = first_ ? g.lda0 : g.lda1;                                                     \
    areg0 = *reinterpret_cast<const float4*>(ab_ + ar0 * ld_);                                       \
    if constexpr (NA >= 2) areg1 = *reinterpret_cast<const float4*>(ab_ + ar1 * ld_);                \
    if constexpr (NA == 4) {                                                                         \
      areg2 = *reinterpret_cast<const float4*>(ab_ + ar2 * ld_);                                     \
      areg3 = *reinterpret_cast<const float4*>(ab_ + ar3 * ld_);                                     \
    }                                                                                                \
    wreg0 = *reinterpret_cast<const float4*>(w0p + k0_);                                             \
    if constexpr (NB >= 2) wreg1 = *reinterpret_cast<const float4*>(w1p + k0_);                      \
    if constexpr (NB == 4) {                                                                         \
      wreg2 = *reinterpret_cast<const float4*>(w2p + k0_);                                           \
      wreg3 = *reinterpret_cast<const float4*>(w3p + k0_);                                           \
    }                                                                                                \
  } while (0)
#define GEMM_STORE_TILE(buf_)                                                                        \
  do {                                                                                               \
    float* as_ = smem + (buf_) * TILE + s_r0 * GLD + s_c4;                                           \
    float* bs_ = as_ + GBM * GLD;                                                                    \
    *reinterpret_cast<float4*>(as_) = areg0;                                                         \
    if constexpr (NA >= 2) *reinterpret_cast<float4*>(as_ + RPP * GLD) = areg1;                      \
    if constexpr (NA == 4) {                                                                         \
      *reinterpret_cast<float4*>(as_ + 2 * RPP * GLD) = areg2;                                       \
      *reinterpret_cast<float4*>(as_ + 3 * RPP * GLD) = areg3;                                       \
    }                                                                                                \
    *reinterpret_cast<float4*>(bs_) = wreg0;                                                         \
    if constexpr (NB >= 2) *reinterpret_cast<float4*>(bs_ + RPP * GLD) = wreg1;                      \
    if constexpr (NB == 4) {                                                                         \
      *reinterpret_cast<float4*>(bs_ + 2 * RPP * GLD) = wreg2;                                       \
      *reinterpret_cast<float4*>(bs_ + 3 * RPP * GLD) = wreg3;                                       \
    }                                                                                                \
  } while (0)

  f32x16 acc[MT][MTN];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < MTN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int a_off = (wm * 32 * MT + l31) * GLD + 4 * h;
  const int b_off = GBM * GLD + (wn * 32 * MTN + l31) * GLD + 4 * h;

  // Staging order ("write after the barrier"): at the top of K step kt the registers hold tile kt+1 (requested one
  // whole step earlier); they go to the other LDS buffer first, then tile kt+2 is requested, then the MFMAs of tile kt
  // run and the step ends on the barrier with no load wait or LDS write in front of it (measured on the loop shape
  // in tools/micro/mfma_feed.hip: 91.8 -> 93.1 % of the MFMA peak at two workgroups per CU).
  GEMM_LOAD_TILE(0);
  GEMM_STORE_TILE(0);
  if (ktiles > 1) GEMM_LOAD_TILE(1);
  __syncthreads();
  GEMM_STAMP(1);
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + 1 < ktiles) {
      GEMM_STORE_TILE((kt + 1) & 1);
      if (kt + 2 < ktiles) GEMM_LOAD_TILE(kt + 2);
    }
    const float* ap = smem + (kt & 1) * TILE + a_off;
    const float* bp = smem + (kt & 1) * TILE + b_off;
#pragma unroll
    for (int gk = 0; gk < BK / 8; ++gk) {
      float4 af[MT], bf[MTN];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const float4*>(ap + mt * 32 * GLD + 8 * gk);
#pragma unroll
      for (int nt = 0; nt < MTN; ++nt) bf[nt] = *reinterpret_cast<const float4*>(bp + nt * 32 * GLD + 8 * gk);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < MTN; ++nt) {
          acc[mt][nt] = mfma32(af[mt].x, bf[nt].x, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].y, bf[nt].y, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].z, bf[nt].z, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].w, bf[nt].w, acc[mt][nt]);
        }
    }
    __syncthreads();
  }
  GEMM_STAMP(2);

  gemm_epilogue<NW, MT, MTN>(g, acc, smem, Y, m0, n0, wm, wn, lane, wave);
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  GEMM_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GEMM_STAMP(4);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Row-owning GEMM with LayerNorm(512, eps 1e-5) + GELU(erf) in the epilogue: the first two stages of the LightGlue
// FFN, Linear(512,512) -> LayerNorm -> GELU (lightglue.py:143-148), in ONE kernel.  The workgroup tile is
// 128 rows x all 512 columns, so the mean / variance of a row never leave the workgroup and the [M,512]
// pre-activation makes no round trip through HBM (round 1: GEMM store + a separate in-place LayerNorm pass,
// 18 launches and 18 x 268 MB per 32-pair step).
//
// 8 waves: wm = wave>>2 owns 64 rows (2 MFMA tiles), wn = wave&3 owns 128 columns (4 MFMA tiles): 8 accumulator
// tiles = 128 registers per lane; per 8-deep k group a wave issues 6 ds_read_b128 for 32 MFMAs.  K tile 16,
// double-buffered LDS: (128 + 512) x 20 floats x 2 = 100 KB -> one workgroup (2 waves per SIMD) per CU.
// Row statistics: every lane holds 32 row partials (its 4 columns of 32 rows); a 5-step butterfly over the 32
// lanes of a half-wave (16+8+4+2+1 = 31 exchanges) leaves each lane with the wave-complete sum of ONE row, the
// four column waves are combined through LDS.  Two passes (mean, then centred squares) as torch's layer_norm.
// ---------------------------------------------------------------------------------------------------------------
#define GW_BN 512

// sum over the 32 lanes (same half-wave) of 32 per-lane values: lane l31 returns the total of value index
// v(l31) = 16*b0 + 8*b1 + 4*b2 + 2*b3 + b4 (b_k = bit k of l31)
__device__ __forceinline__ float halfwave_transpose_sum32(const float (&x)[32], int l31) {
  float y16[16], y8[8], y4[4], y2[2];
  const bool b0 = l31 & 1, b1 = l31 & 2, b2 = l31 & 4, b3 = l31 & 8, b4 = l31 & 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float mine = b0 ? x[16 + i] : x[i], theirs = b0 ? x[i] : x[16 + i];
    y16[i] = mine + __shfl_xor(theirs, 1);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float mine = b1 ? y16[8 + i] : y16[i], theirs = b1 ? y16[i] : y16[8 + i];
    y8[i] = mine + __shfl_xor(theirs, 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float mine = b2 ? y8[4 + i] : y8[i], theirs = b2 ? y8[i] : y8[4 + i];
    y4[i] = mine + __shfl_xor(theirs, 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float mine = b3 ? y4[2 + i] : y4[i], theirs = b3 ? y4[i] : y4[2 + i];
    y2[i] = mine + __shfl_xor(theirs, 8);
  }
  const float mine = b4 ? y2[1] : y2[0], theirs = b4 ? y2[0] : y2[1];
  return mine + __shfl_xor(theirs, 16);
}

// WM = row groups of 64 per workgroup:
//   WM = 2: 128 x 512 tile, 8 waves, K tile 16, 100 KB LDS: one workgroup per CU;
//   WM = 1:  64 x 512 tile, 4 waves, K tile 8,   54 KB LDS: two workgroups per CU that run out of phase, so the
//            VALU-heavy epilogue (LayerNorm + erf, ~50 instructions per element) and the store drain of one
//            overlap the MFMAs of the other.
//
// MLP = true (WM = 2 only; round 4): the WHOLE LightGlue FFN in the kernel -- Linear(512,512) -> LayerNorm -> GELU ->
// Linear(512,256) + residual (lightglue.py:143-148,162-164).  The activated 128 x 512 hidden tile stays in the
// accumulator registers; it is handed to the second GEMM through LDS in four 128-column chunks (the column group wn = c
// owns chunk c), which the eight waves consume as the A operand of Y[128,256] += H[:, chunk] . W3[:, chunk]^T with W3
// staged in 16-deep K tiles exactly like W0 in the first loop.  Second-GEMM wave tile: wm -> 64 rows, wn -> 64 output
// columns (2 x 2 MFMA tiles, 64 more accumulator registers).  Same k order per output element as gemm_nt_kernel
// (ascending 8-deep groups, k = 4h + s inside a group), same epilogue association (residual + (acc + bias)): the
// results are bit-identical to the two-kernel path.  The [M,512] hidden activation never exists in HBM.
#define FF_HLD 148  // hidden-chunk row stride in floats (= 20 mod 64: the conflict-free ds_read_b128 pattern of the K-tile rows)
template <int WM, bool MLP = false>
__global__ __launch_bounds__(256 * WM, 2) void gemm_rows512_ln_gelu_kernel(GemmArgs g, const float* __restrict__ gamma,
                                                                           const float* __restrict__ beta,
                                                                           const float* __restrict__ W3, int ldw3,
                                                                           const float* __restrict__ b3) {
  static_assert(!MLP || WM == 2, "the fused MLP is built for the 128-row tile");
  constexpr int BM = 64 * WM, BK = WM == 2 ? 16 : 8, LD = BK + 4, C4 = BK / 4;
  constexpr int TILE = (BM + GW_BN) * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.x * BM;
  const float* A0 = g.A0;
  const float* A1 = g.A1;
  const int ktiles = (g.K0 + g.K1) / BK;
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  unsigned long long* stamp_base = g.stamps ? g.stamps + ((size_t)blockIdx.x * (4 * WM) + wave) * 8 : nullptr;
  if (g.stamps && lane == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamp_base[5] = hw;
    stamp_base[6] = xcc;
  }
  GEMM_STAMP(0);
#endif

  // staging: thread -> float4 column s_c4 of tile row s_r0 (weights: rows s_r0 + 128 i).  For the 16-deep tile the
  // rows are permuted inside blocks of 8 so that the ds_write_b128 is conflict-free (see gemm_nt_kernel).
  const int s_c4 = (tid % C4) * 4;
  const int q_ = tid / C4;                       // 0..127
  const int s_r0 = C4 == 4 ? ((q_ & ~7) | ((q_ & 1) << 2) | ((q_ >> 1) & 3)) : q_;
  const bool stage_a = s_r0 < BM;                // WM = 1: half of the threads carry an A row
  const size_t ar = (size_t)min(m0 + (stage_a ? s_r0 : 0), g.M - 1);
  const float* w0p = g.W + (size_t)(s_r0)*g.ldw + s_c4;
  const float* w1p = w0p + (size_t)128 * g.ldw;
  const float* w2p = w0p + (size_t)256 * g.ldw;
  const float* w3p = w0p + (size_t)384 * g.ldw;
  float4 areg = make_float4(0.f, 0.f, 0.f, 0.f), wreg0, wreg1, wreg2, wreg3;
#define GW_LOAD(kt)                                                              \
  do {                                                                           \
    const int k0_ = (kt) * BK;                                                   \
    const bool first_ = k0_ < g.K0;                                              \
    const float* ab_ = (first_ ? A0 : A1) + (first_ ? k0_ : k0_ - g.K0) + s_c4;  \
    const size_t ld_ = first_ ? g.lda0 : g.lda1;                                 \
    if (WM == 2 || stage_a) areg = *reinterpret_cast<const float4*>(ab_ + ar * ld_); \
    wreg0 = *reinterpret_cast<const float4*>(w0p + k0_);                         \
    wreg1 = *reinterpret_cast<const float4*>(w1p + k0_);                         \
    wreg2 = *reinterpret_cast<const float4*>(w2p + k0_);                         \
    wreg3 = *reinterpret_cast<const float4*>(w3p + k0_);                         \
  } while (0)
#define GW_STORE(buf_)                                                           \
  do {                                                                           \
    float* bs_ = smem + (buf_) * TILE + BM * LD + s_r0 * LD + s_c4;              \
    if (WM == 2 || stage_a) *reinterpret_cast<float4*>(smem + (buf_) * TILE + s_r0 * LD + s_c4) = areg; \
    *reinterpret_cast<float4*>(bs_) = wreg0;                                     \
    *reinterpret_cast<float4*>(bs_ + 128 * LD) = wreg1;                          \
    *reinterpret_cast<float4*>(bs_ + 256 * LD) = wreg2;                          \
    *reinterpret_cast<float4*>(bs_ + 384 * LD) = wreg3;                          \
  } while (0)

  f32x16 acc[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int a_off = (wm * 64 + l31) * LD + 4 * h;
  const int b_off = BM * LD + (wn * 128 + l31) * LD + 4 * h;

  GW_LOAD(0);
  GW_STORE(0);
  if (ktiles > 1) GW_LOAD(1);
  __syncthreads();
  GEMM_STAMP(1);
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + 1 < ktiles) {
      GW_STORE((kt + 1) & 1);
      if (kt + 2 < ktiles) GW_LOAD(kt + 2);
    }
    const float* ap = smem + (kt & 1) * TILE + a_off;
    const float* bp = smem + (kt & 1) * TILE + b_off;
#pragma unroll
    for (int gk = 0; gk < BK / 8; ++gk) {
      float4 af[2], bf[4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const float4*>(ap + mt * 32 * LD + 8 * gk);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bf[nt] = *reinterpret_cast<const float4*>(bp + nt * 32 * LD + 8 * gk);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          acc[mt][nt] = mfma32(af[mt].x, bf[nt].x, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].y, bf[nt].y, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].z, bf[nt].z, acc[mt][nt]);
          acc[mt][nt] = mfma32(af[mt].w, bf[nt].w, acc[mt][nt]);
        }
    }
    __syncthreads();
  }
#undef GW_LOAD
#undef GW_STORE
  GEMM_STAMP(2);
  // fused MLP: W3's K tiles are staged through LDS behind the hidden chunk; tile 0 is requested now and lands under
  // the LayerNorm / GELU epilogue (all reads of the first loop's buffers are behind its closing barrier)
  constexpr int W3T = 256 * LD;                         // one staged W3 K tile: 256 output columns x 16 k
  float* const Hs = smem;                               // [128][FF_HLD]
  float* const Ws = smem + 128 * FF_HLD;                // 2 x [256][LD]
  const float* w3p0 = MLP ? W3 + (size_t)s_r0 * ldw3 + s_c4 : nullptr;
  const float* w3p1 = MLP ? w3p0 + (size_t)128 * ldw3 : nullptr;
#define FF_LOAD(T_)                                                   \
  do {                                                                \
    wreg0 = *reinterpret_cast<const float4*>(w3p0 + (T_) * 16);       \
    wreg1 = *reinterpret_cast<const float4*>(w3p1 + (T_) * 16);       \
  } while (0)
#define FF_STORE(buf_)                                                \
  do {                                                                \
    float* bs_ = Ws + (buf_) * W3T + s_r0 * LD + s_c4;                \
    *reinterpret_cast<float4*>(bs_) = wreg0;                          \
    *reinterpret_cast<float4*>(bs_ + 128 * LD) = wreg1;               \
  } while (0)
  if constexpr (MLP) FF_LOAD(0);

  // ---- epilogue: + bias, LayerNorm over the 512 columns of each row, GELU, store ----
  float* red = smem;          // [4][BM] per-column-wave row partials
  float* stat = smem + 4 * BM;  // [BM] mean, then [BM] 1/std
  float bi[4], ga[4], be[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = wn * 128 + nt * 32 + l31;
    bi[nt] = g.bias ? g.bias[col] : 0.f;
    ga[nt] = gamma[col];
    be[nt] = beta[col];
  }
  // waited for HERE, once (see gemm_epilogue: otherwise hipcc re-waits for gamma / beta inside every predicated store
  // block, and a vmcnt wait behind stores also waits for those stores)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(bi[nt]), "+v"(ga[nt]), "+v"(be[nt]));
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bi[nt];
  // row of value index v = 16 mt + r held by this lane after the butterfly
  const int vb = ((l31 & 1) << 4) | ((l31 & 2) << 2) | (l31 & 4) | ((l31 & 8) >> 2) | ((l31 & 16) >> 4);
  const int my_row = wm * 64 + (vb >> 4) * 32 + acc_row(vb & 15, h);
  float mu[2][16];
  {
    float x[32];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        x[mt * 16 + r] = (acc[mt][0][r] + acc[mt][1][r]) + (acc[mt][2][r] + acc[mt][3][r]);
    red[wn * BM + my_row] = halfwave_transpose_sum32(x, l31);
  }
  __syncthreads();
  if (tid < BM) stat[tid] = ((red[tid] + red[BM + tid]) + (red[2 * BM + tid] + red[3 * BM + tid])) * (1.f / 512.f);
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 m4 = *reinterpret_cast<const float4*>(stat + wm * 64 + mt * 32 + 8 * gq + 4 * h);
      mu[mt][4 * gq] = m4.x; mu[mt][4 * gq + 1] = m4.y; mu[mt][4 * gq + 2] = m4.z; mu[mt][4 * gq + 3] = m4.w;
    }
  {
    float x[32];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d0 = acc[mt][0][r] - mu[mt][r], d1 = acc[mt][1][r] - mu[mt][r];
        const float d2 = acc[mt][2][r] - mu[mt][r], d3 = acc[mt][3][r] - mu[mt][r];
        x[mt * 16 + r] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    red[wn * BM + my_row] = halfwave_transpose_sum32(x, l31);  // `red` was last read before the previous barrier
  }
  __syncthreads();
  if (tid < BM) {
    const float var = ((red[tid] + red[BM + tid]) + (red[2 * BM + tid] + red[3 * BM + tid])) * (1.f / 512.f);
    stat[BM + tid] = 1.f / sqrtf(var + 1e-5f);
  }
  __syncthreads();
  if constexpr (!MLP) GEMM_STAMP(3);  // statistics done
  const bool full_rows = m0 + BM <= g.M;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 r4 = *reinterpret_cast<const float4*>(stat + BM + wm * 64 + mt * 32 + 8 * gq + 4 * h);
      const float rs[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * gq + j;
        const int row = m0 + wm * 64 + mt * 32 + acc_row(r, h);
        if constexpr (MLP) {  // the activated value replaces the pre-activation in its accumulator register
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[mt][nt][r] = gfc_gelu((acc[mt][nt][r] - mu[mt][r]) * rs[j] * ga[nt] + be[nt]);
        } else if (full_rows || row < g.M) {  // full_rows: workgroup-uniform, no per-row predicate blocks
          float* yp = g.Y + (size_t)row * g.ldy + wn * 128 + l31;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            float y = (acc[mt][nt][r] - mu[mt][r]) * rs[j] * ga[nt] + be[nt];
#if !defined(GW_DIAG) || GW_DIAG < 1  // diagnostic builds (tools/ab_build.sh): 1 = no erf, 2 = no store either
            y = gfc_gelu(y);
#endif
#if !defined(GW_DIAG) || GW_DIAG < 2
            yp[nt * 32] = y;
#else
            if (y == 1234.5678f) yp[nt * 32] = y;
#endif
          }
        }
      }
    }
  }
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  if constexpr (MLP) GEMM_STAMP(3);  // MLP stamps: 3 = LayerNorm + GELU done, 4 = second K loop done, 7 = stores issued
  else GEMM_STAMP(4);
#endif
  if constexpr (MLP) {
    // ---- second GEMM: Y[128,256] = H[128,512] . W3[256,512]^T, K walked in four chunks of 128 hidden columns ----
    f32x16 acc2[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mt][nt][r] = 0.f;
    const int a_off3 = (wm * 64 + l31) * FF_HLD + 4 * h;
    const int b_off3 = (wn * 64 + l31) * LD + 4 * h;
    FF_STORE(0);  // tile 0 (requested before the epilogue); `stat` / `red` live below Ws and were last read above
    FF_LOAD(1);
    __syncthreads();  // every wave is past its last read of `stat` (which aliases the hidden chunk)
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      if (wn == c) {  // wave-uniform: this column group's 128 hidden columns -> LDS (row = accumulator row, column on the lane)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Hs[(wm * 64 + mt * 32 + acc_row(r, h)) * FF_HLD + nt * 32 + l31] = acc[mt][nt][r];
      }
      __syncthreads();
#pragma unroll 1
      for (int t = 0; t < 8; ++t) {
        const int T = c * 8 + t;
        if (T + 1 < 32) {
          FF_STORE((T + 1) & 1);
          if (T + 2 < 32) FF_LOAD(T + 2);
        }
        const float* ap = Hs + a_off3 + 16 * t;
        const float* bp = Ws + (T & 1) * W3T + b_off3;
#pragma unroll
        for (int gk = 0; gk < 2; ++gk) {
          float4 af[2], bf[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const float4*>(ap + mt * 32 * FF_HLD + 8 * gk);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) bf[nt] = *reinterpret_cast<const float4*>(bp + nt * 32 * LD + 8 * gk);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              acc2[mt][nt] = mfma32(af[mt].x, bf[nt].x, acc2[mt][nt]);
              acc2[mt][nt] = mfma32(af[mt].y, bf[nt].y, acc2[mt][nt]);
              acc2[mt][nt] = mfma32(af[mt].z, bf[nt].z, acc2[mt][nt]);
              acc2[mt][nt] = mfma32(af[mt].w, bf[nt].w, acc2[mt][nt]);
            }
        }
        __syncthreads();
      }
    }
    GEMM_STAMP(4);
    // ---- epilogue: Y = residual + (acc + b3), straight from the accumulator layout (128-byte row segments) ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = wn * 64 + nt * 32 + l31;
      float bi3 = b3 ? b3[col] : 0.f;
      asm volatile("" : "+v"(bi3));
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row0 = m0 + wm * 64 + mt * 32 + 4 * h;
        const size_t o0 = (size_t)row0 * g.ldy + col;
        float rv[16];
        if (full_rows) {  // workgroup-uniform: all residual loads first, then the stores back to back
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = g.residual ? g.residual[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * g.ldy] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) g.Y[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * g.ldy] = rv[r] + (acc2[mt][nt][r] + bi3);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            if (row < g.M) {
              const size_t o = (size_t)row * g.ldy + col;
              g.Y[o] = (g.residual ? g.residual[o] : 0.f) + (acc2[mt][nt][r] + bi3);
            }
          }
        }
      }
    }
    GEMM_STAMP(7);
  }
#undef FF_LOAD
#undef FF_STORE
}

template <int WM>
static int launch_rows512(const GemmArgs& g, const float* gamma, const float* beta, hipStream_t st) {
  constexpr int BM = 64 * WM, BK = WM == 2 ? 16 : 8;
  constexpr size_t lds = (size_t)2 * (BM + GW_BN) * (BK + 4) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};
  if (lds > 64 * 1024) gfc_allow_dynamic_lds((const void*)gemm_rows512_ln_gelu_kernel<WM>, lds, lds_ok);
  GemmArgs gd = g;
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  gd.stamps = g_diag_stamps;
#endif
  hipLaunchKernelGGL(gemm_rows512_ln_gelu_kernel<WM>, dim3((g.M + BM - 1) / BM), dim3(256 * WM), lds, st, gd, gamma, beta,
                     (const float*)nullptr, 0, (const float*)nullptr);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_ffn_fused(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1, const float* W0,
                             int ldw0, const float* b0, const float* gamma, const float* beta, const float* W3, int ldw3,
                             const float* b3, const float* residual, float* Y, int ldy, int M, void* stream) {
  if (!A0 || !W0 || !gamma || !beta || !W3 || !Y || M <= 0 || K0 <= 0 || K0 % GBK || K1 % GBK || K1 < 0)
    return GFC_ERR_INVALID;
  if ((K1 > 0) != (A1 != nullptr)) return GFC_ERR_INVALID;
  if (lda0 % 4 || (A1 && lda1 % 4) || ldw0 % 4 || ldw3 % 4 || ldw3 < GW_BN || ldy < 256) return GFC_ERR_INVALID;
  GemmArgs g = {};
  g.A0 = A0; g.A1 = A1; g.W = W0; g.bias = b0; g.residual = residual; g.Y = Y;
  g.lda0 = lda0; g.lda1 = lda1; g.ldw = ldw0; g.ldy = ldy;
  g.K0 = K0; g.K1 = K1; g.M = M; g.N = GW_BN; g.alpha = 1.f;
  constexpr size_t k1 = (size_t)2 * (128 + GW_BN) * 20, k2 = (size_t)128 * FF_HLD + (size_t)2 * 256 * 20;
  constexpr size_t lds = (k1 > k2 ? k1 : k2) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};
  gfc_allow_dynamic_lds((const void*)gemm_rows512_ln_gelu_kernel<2, true>, lds, lds_ok);
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  g.stamps = g_diag_stamps;
#endif
  hipLaunchKernelGGL((gemm_rows512_ln_gelu_kernel<2, true>), dim3((M + 127) / 128), dim3(512), lds, (hipStream_t)stream, g,
                     gamma, beta, W3, ldw3, b3);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

extern "C" int gfc_linear_layernorm_gelu(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1,
                                         const float* W, int ldw, const float* bias, const float* gamma,
                                         const float* beta, float* Y, int ldy, int M, int N, void* stream) {
  if (!A0 || !W || !gamma || !beta || !Y || M <= 0 || K0 <= 0 || K0 % GBK || K1 % GBK || K1 < 0) return GFC_ERR_INVALID;
  if ((K1 > 0) != (A1 != nullptr)) return GFC_ERR_INVALID;
  if (lda0 % 4 || (A1 && lda1 % 4) || ldw % 4 || ldy < N) return GFC_ERR_INVALID;
  if (N != GW_BN) return GFC_ERR_UNSUPPORTED;  // the row statistics span exactly one workgroup tile
  GemmArgs g = {};
  g.A0 = A0; g.A1 = A1; g.W = W; g.bias = bias; g.Y = Y;
  g.lda0 = lda0; g.lda1 = lda1; g.ldw = ldw; g.ldy = ldy;
  g.K0 = K0; g.K1 = K1; g.M = M; g.N = N; g.alpha = 1.f;
  // 128-row / 8-wave tiles by default (same-box A/B at M = 65536: 336 us; 64-row tiles with two workgroups per CU
  // 357 us -- the 8-deep K tile doubles the barriers; GEMM + LayerNorm pass 390 us).  GFC_FFN_FUSED = 1 forces the
  // 64-row variant.
  if (gfc_knobs().ffn_fused == 1) return launch_rows512<1>(g, gamma, beta, (hipStream_t)stream);
  return launch_rows512<2>(g, gamma, beta, (hipStream_t)stream);
}

template <int NW, int MT, int BK, int MTN = MT>
static int launch_gemm_t(const GemmArgs& g, int batch, hipStream_t st) {
  constexpr int BM = 64 * MT, BN = 32 * MTN * NW;
  // K-loop buffers, or the per-wave transpose patches of the rotary / residual epilogue if those are larger
  constexpr size_t kloop = (size_t)2 * (BM + BN) * (BK + 4), patches = (size_t)2 * NW * 32 * (32 * MTN + 4);
  const size_t lds = (kloop > patches ? kloop : patches) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok{0};  // per (instantiation, device): runtime.h
  if (lds > 64 * 1024) gfc_allow_dynamic_lds((const void*)gemm_nt_kernel<NW, MT, BK, MTN>, lds, lds_ok);
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch);
  GemmArgs ga = g;
  ga.xcd_remap = gfc_knobs().xcd_remap != 0 && (long long)grid.x * grid.y * grid.z >= 16;
  ga.stagger = gfc_knobs().gemm_stagger;
  ga.first_round = gfc_device_cus() * 4;
  ga.wide_stores = gfc_knobs().gemm_epi == 1 && g.ldy % 4 == 0 && (reinterpret_cast<size_t>(g.Y) & 15) == 0 &&
                   g.strideY % 4 == 0;
#if defined(GEMM_DIAG) && (GEMM_DIAG & 16)
  ga.stamps = g_diag_stamps;
#endif
  hipLaunchKernelGGL((gemm_nt_kernel<NW, MT, BK, MTN>), grid, dim3(128 * NW), lds, st, ga);
  GFC_LAUNCH_CHECK();
  return GFC_OK;
}

static int launch_gemm(const GemmArgs& g, int batch, hipStream_t st) {
  // GFC_GEMM_TILE forces one of the two tiles for every problem size: 4 = 128x128 (K tile 16), 3 = 64x64 (K tile 32);
  // 0 / unset / anything else = by problem size.  (The 128x256, 256x128, 64x32, 16-deep 64x64 and LDS-DMA variants of
  // rounds 2-5 measured no faster anywhere and were removed in round 6: TUNING_LOG.md.)
  const int forced = gfc_knobs().gemm_tile;
  auto tiles = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * batch; };
  int choice = (forced == 3 || forced == 4) ? forced : 0;
  if (!choice) {
    // 128x128 tiles with a 16-deep K tile (40 KB LDS, four workgroups per CU: while one drains its stores the
    // others keep the MFMA pipe busy; +5..11 % over the 32-deep variants at M = 65536) once there are three per CU,
    // else 64x64 tiles so that small-batch GEMMs still cover the chip
    choice = tiles(128, 128) >= 768 ? 4 : 3;
  }
  if (choice == 4) return launch_gemm_t<2, 2, 16>(g, batch, st);  // 40 KB LDS: 4 workgroups / CU
  return launch_gemm_t<2, 1, 32>(g, batch, st);
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
  GemmArgs g = {};
  g.A0 = A0; g.A1 = A1; g.W = W; g.bias = bias; g.scale = scale; g.shift = shift; g.residual = residual;
  g.rot_cos = rot_cos; g.rot_sin = rot_sin; g.Y = Y;
  g.strideA = g.strideW = g.strideY = 0;
  g.lda0 = lda0; g.lda1 = lda1; g.ldw = ldw; g.ldy = ldy;
  g.K0 = K0; g.K1 = K1; g.M = M; g.N = N; g.rot_cols = rot_cols; g.alpha = alpha;
  return launch_gemm(g, 1, (hipStream_t)stream);
}

int gfc_linear_rot_packed(const float* A0, int lda0, int K0, const float* W, int ldw, const float* bias, const float* rot_cs,
                          int rot_cols, float* Y, int ldy, int M, int N, void* stream) {
  if (!A0 || !W || !Y || !rot_cs || M <= 0 || N <= 0 || K0 <= 0 || K0 % GBK || rot_cols % 64 || lda0 % 4 || ldw % 4)
    return GFC_ERR_INVALID;
  GemmArgs g = {};
  g.A0 = A0; g.W = W; g.bias = bias; g.rot_cs = rot_cs; g.Y = Y;
  g.lda0 = lda0; g.ldw = ldw; g.ldy = ldy;
  g.K0 = K0; g.K1 = 0; g.M = M; g.N = N; g.rot_cols = rot_cols; g.alpha = 1.f;
  return launch_gemm(g, 1, (hipStream_t)stream);
}

extern "C" int gfc_batched_nt(const float* A, int lda, long long strideA, const float* Bm, int ldb, long long strideB,
                              float* Y, int ldy, long long strideY, int M, int N, int K, int batch, void* stream) {
  if (!A || !Bm || !Y || M <= 0 || N <= 0 || K <= 0 || K % GBK || batch <= 0) return GFC_ERR_INVALID;
  if (lda % 4 || ldb % 4 || strideA % 4 || strideB % 4) return GFC_ERR_INVALID;
  GemmArgs g = {};
  g.A0 = A; g.A1 = nullptr; g.W = Bm; g.bias = nullptr; g.scale = nullptr; g.shift = nullptr; g.residual = nullptr;
  g.rot_cos = nullptr; g.rot_sin = nullptr; g.Y = Y;
  g.strideA = strideA; g.strideW = strideB; g.strideY = strideY;
  g.lda0 = lda; g.lda1 = 0; g.ldw = ldb; g.ldy = ldy;
  g.K0 = K; g.K1 = 0; g.M = M; g.N = N; g.rot_cols = 0; g.alpha = 1.f;
  return launch_gemm(g, batch, (hipStream_t)stream);
}
