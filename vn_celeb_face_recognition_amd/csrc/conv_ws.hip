// Wave-specialised implicit-GEMM convolution for gfx950 (MI355X): big tiles at one workgroup per CU.
//
// With one 8-wave workgroup per CU the plain ring kernel (conv_igemm.hip) phase-locks: after the
// per-K-tile barrier every wave first queues its LDS-DMA pieces -- the texture path accepts about
// 28 B/clk/CU, so a 48-KiB K tile holds the waves ~1700 cycles at issue -- and only then
// multiplies, so the MFMA pipes idle during the issue phase and the DMA path idles during the
// MFMA phase.  Here the roles are split (MI355X_MICROARCH.md "ring-gemm"):
//   * WM x WN consumer waves (one per SIMD) only read fragments and issue MFMAs (software-
//     pipelined over half K tiles, so LDS latency hides under the previous half's MFMAs);
//   * LW loader waves only issue LDS-DMA pieces (im2col gather + swizzle in the source address,
//     exactly the ring kernel's image) and wait for them with counted vmcnt.
// One raw s_barrier per K tile joins both roles: after it tile kt has landed (the loaders waited
// for it) and stage (kt-1) % S is free (the consumers drained their reads of it), so the loaders
// refill it while the consumers multiply tile kt.  Same K order, operand roles and epilogue as the
// ring kernel, so results are bit-identical to it.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_device.h"

namespace vnf {

template <typename T, int BM, int BN, int WM, int WN, int S, int LW, bool DBG = false>
__global__ __launch_bounds__((WM* WN + LW) * 64) void conv_igemm_ws_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int CH = 16 / ES, BKE = 128 / ES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int NC = WM * WN, NT = (NC + LW) * 64;
  constexpr int PA = BM / 8, PB = BN / 8;          // 1-KiB pieces (8 rows x 128 B) per K tile
  constexpr int LA = PA / LW, LB = PB / LW;        // pieces per loader wave
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int CST = BN + 4;
  // epilogue staging budget handed to conv_epilogue: one pass over the whole tile only while its per-thread residual
  // prefetch (NIT x 8 floats, 256 consumer threads) stays within 64 registers -- beside 128 accumulators more spills
  constexpr int EPI_LDS = (BM * CST * 4 <= S * STAGE && (BM * (BN / 8)) / (NC * 64) <= 8) ? S * STAGE : (BM / WM) * CST * 4;
  constexpr int EPASS = (BM * CST * 4 <= EPI_LDS) ? 1 : WM;  // conv_epilogue's pass count
  static_assert(S >= 3 && PA % LW == 0 && PB % LW == 0, "pieces divide over the loader waves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = xcd_remap(blockIdx.x, a.nblk);
  const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = a.nkt;

  int4* sK = reinterpret_cast<int4*>(smem + S * STAGE);
  for (int i = tid; i < nkt * 8; i += NT) sK[i] = a.ktab[i];
  __syncthreads();  // gather table visible

  if (wave >= NC) {
    // ------------------------------------------------------------------ loader wave
    const int lw = wave - NC;
    const int prow = lane >> 3, lcol = lane & 7;
    const int lchunk = lcol ^ prow;  // piece rows start at multiples of 8: (row & 7) == prow
    const int HoWo = a.Ho * a.Wo;
    int abase[LA], ahi[LA], awi[LA];
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      const int m = m0 + (lw + p * LW) * 8 + prow;
      if (m < a.M) {
        const int n = m / HoWo, r = m - n * HoWo;
        const int ho = r / a.Wo, wo = r - ho * a.Wo;
        ahi[p] = ho * a.sh - a.ph;
        awi[p] = wo * a.sw - a.pw;
        abase[p] = ((n * a.H + ahi[p]) * a.W + awi[p]) * a.ldx;
      } else {
        ahi[p] = -(1 << 24);
        awi[p] = 0;
        abase[p] = 0;
      }
    }
    const char* wsrc = a.w + (size_t)(n0 + lw * 8 + prow) * a.wrs + lchunk * 16;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    auto issue = [&](int kt) {
      const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((kt % S) * STAGE) + (unsigned)(lw * 1024));
      const int4 e = sK[kt * 8 + lchunk];
#pragma unroll
      for (int p = 0; p < LA; ++p) {
        const int hi = ahi[p] + e.y, wi = awi[p] + e.z;
        const bool ok = e.w && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const char* src = ok ? a.x + (size_t)(abase[p] + e.x) * ES : a.zero;
        glds16(src, sbase + p * (LW * 1024));
      }
#pragma unroll
      for (int p = 0; p < LB; ++p)
        glds16(wsrc + (size_t)(p * LW * 8) * a.wrs + (size_t)kt * a.wts, sbase + BM * 128 + p * (LW * 1024));
    };
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
      if (t < nkt) issue(t);
    for (int kt = 0; kt < nkt; ++kt) {
      if constexpr (DBG) {
        const bool st = a.dbg && blockIdx.x == 600 && lane == 0;
        long long* d = a.dbg + (wave * 64 + kt) * 4;
        if (st) d[0] = __builtin_readcyclecounter();
        if (kt + S - 2 < nkt)
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * (LA + LB)) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (st) d[1] = __builtin_readcyclecounter();
        asm volatile("s_barrier" ::: "memory");
        if (st) d[2] = __builtin_readcyclecounter();
        if (kt + S - 1 < nkt) issue(kt + S - 1);
        if (st) d[3] = __builtin_readcyclecounter();
        continue;
      }
      if (kt + S - 2 < nkt)
        wait_dma_and_barrier<(S - 2) * (LA + LB)>();
      else
        wait_dma_and_barrier<0>();
      if (kt + S - 1 < nkt) issue(kt + S - 1);
    }
    __syncthreads();
    if (conv_epilogue_is_fast<T, BM, BN>(a, S * STAGE)) {  // the consumers' epilogue barriers
      __syncthreads();
      return;
    }
    for (int pass = 0; pass < EPASS; ++pass) {
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------- consumer wave
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fgrp = lane >> 4;
  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // Big wave tiles (8x4 MFMA tiles = 128 accumulator registers) keep ONE fragment set: two would spill.  Their 32
  // MFMAs per half K tile (512 cycles) dwarf the exposed read latency anyway; smaller tiles double-buffer.
  constexpr bool PIPE = (TM + TN) < 12;
  constexpr int NF = PIPE ? 2 : 1;
  uint4 xf[NF][TM], wf[NF][TN];
  auto read_frags = [&](int kt, int ks) {
    const int fs = PIPE ? ks : 0;
    const char* sA = smem + (kt % S) * STAGE;
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WTM + i * 16 + frow;
      xf[fs][i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * WTN + j * 16 + frow;
      wf[fs][j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
    }
  };
  auto mma = [&](int ks) {
    const int fs = PIPE ? ks : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[fs][j], xf[fs][i]);
  };
  if constexpr (is_planar<T>::value) {
    // planar split-f16 (see conv_igemm.hip): half 0 / 1 of a K tile = hi / lo plane of the same 32 k values
    uint4 xh[NF][TM], wh[NF][TN], xl[TM], wl[TN];
    auto rd = [&](int kt, int ks, uint4 (&x)[TM], uint4 (&w)[TN]) {
      const char* sA = smem + (kt % S) * STAGE;
      const char* sB = sA + BM * 128;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 16 + frow;
        x[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 16 + frow;
        w[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
    };
    auto cross = [&](uint4 (&x)[TM], uint4 (&w)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_cross(acc[i][j], w[j], wl[j], x[i], xl[i]);
    };
    auto hh = [&](uint4 (&x)[TM], uint4 (&w)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_hh(acc[i][j], w[j], x[i]);
    };
    if constexpr (PIPE) {
      auto step = [&](auto P, int kt) {
        constexpr int c = decltype(P)::value;
        wait_dma_and_barrier<0>();
        rd(kt, 0, xh[c], wh[c]);
        if (kt > 0) cross(xh[c ^ 1], wh[c ^ 1]);
        rd(kt, 1, xl, wl);
        hh(xh[c], wh[c]);
      };
      int kt = 0;
      for (; kt + 1 < nkt; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
      }
      if (kt < nkt) {
        step(std::integral_constant<int, 0>{}, kt);
        cross(xh[0], wh[0]);
      } else {
        cross(xh[NF - 1], wh[NF - 1]);
      }
    } else {
      for (int kt = 0; kt < nkt; ++kt) {
        wait_dma_and_barrier<0>();
        rd(kt, 0, xh[0], wh[0]);
        rd(kt, 1, xl, wl);
        hh(xh[0], wh[0]);
        cross(xh[0], wh[0]);
      }
    }
    __syncthreads();
    conv_epilogue<T, BM, BN, WM, WN, EPI_LDS>(a, acc, smem, m0, n0, S * STAGE);
    return;
  }
  bool pend = false;
  for (int kt = 0; kt < nkt; ++kt) {
    if constexpr (DBG) {
      const bool st = a.dbg && blockIdx.x == 600 && lane == 0;
      long long* d = a.dbg + (wave * 64 + kt) * 4;
      if (st) d[0] = __builtin_readcyclecounter();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (st) d[1] = __builtin_readcyclecounter();
      asm volatile("s_barrier" ::: "memory");
      if (st) d[2] = __builtin_readcyclecounter();
      read_frags(kt, 0);
      if (pend) mma(1);
      pend = BKE / 2 < a.K - kt * BKE;
      if (pend) read_frags(kt, 1);
      mma(0);
      if (st) d[3] = __builtin_readcyclecounter();
      continue;
    }
    wait_dma_and_barrier<0>();  // no DMA of its own: drains this wave's fragment reads, then joins
    if constexpr (PIPE) {
      read_frags(kt, 0);
      if (pend) mma(1);
      pend = BKE / 2 < a.K - kt * BKE;
      if (pend) read_frags(kt, 1);
      mma(0);
    } else {
      read_frags(kt, 0);
      mma(0);
      if (BKE / 2 < a.K - kt * BKE) {
        read_frags(kt, 1);
        mma(1);
      }
    }
  }
  if (pend) mma(1);
  __syncthreads();
  conv_epilogue<T, BM, BN, WM, WN, EPI_LDS>(a, acc, smem, m0, n0, S * STAGE);
}

// ---- persistent form -------------------------------------------------------------------------------------------
// One workgroup per CU walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The loader waves run the ring straight
// across tile boundaries (K tiles of the next output tile are in flight while the consumers finish this one), and the
// consumers store their accumulators from registers -- bias, ReLU, rounding to T, a v_permlane16_swap between the two
// lane rows that hold neighbouring channel quads so every lane owns 8 channels = one 16-byte store -- without touching
// LDS or a barrier.  In-kernel stamps of the one-tile-per-workgroup form (tools/stamp_patch.py): a workgroup spent 16 %
// of its life filling the pipeline and 18-29 % in the epilogue (the store burst of 256 workgroups in lockstep runs at
// HBM write speed); here both overlap the next tile's K loop.  Only for the layers of the fast epilogue (bias + ReLU,
// output in T, 2-byte T or planar split-f16).  RES: a residual operand (Block8's up projection, small tiles only): its
// 16-byte chunks are fetched at the START of the tile -- behind the K loop they would wait, vmcnt being in issue order,
// for the previous tile's stores -- and un-swapped into the accumulator layout in the epilogue.  Same K order, operand
// roles, fp32 sum, bias add, residual add, max and rounding: bit-identical results.
template <typename T, int BM, int BN, int WM, int WN, int S, int LW, bool RES = false>
__global__ __launch_bounds__((WM* WN + LW) * 64) void conv_igemm_wsp_kernel(const KArgs a) {
  static_assert(sizeof(T) == 2 || is_planar<T>::value, "register epilogue: 2-byte elements or planar split-f16 units");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int BKE = 128 / ES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int NC = WM * WN, NT = (NC + LW) * 64;
  constexpr int PA = BM / 8, PB = BN / 8;
  constexpr int LA = PA / LW, LB = PB / LW;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(S >= 3 && PA % LW == 0 && PB % LW == 0 && TN % 2 == 0, "pieces divide over the loader waves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkt = a.nkt;
  const int G = (int)gridDim.x;
  const int ntile = (a.nblk - (int)blockIdx.x + G - 1) / G;  // this workgroup's tiles
  const int Q = ntile * nkt;                                   // ... and K tiles, one barrier each

  int4* sK = reinterpret_cast<int4*>(smem + S * STAGE);
  for (int i = tid; i < nkt * 8; i += NT) sK[i] = a.ktab[i];
  __syncthreads();

  if (wave >= NC) {
    // ------------------------------------------------------------------ loader wave
    const int lw = wave - NC;
    const int prow = lane >> 3, lcol = lane & 7;
    const int lchunk = lcol ^ prow;
    const int HoWo = a.Ho * a.Wo;
    int abase[LA], ahi[LA], awi[LA];
    const char* wsrc = nullptr;
    auto setup = [&](int it) {
      const int bid = xcd_remap((int)blockIdx.x + it * G, a.nblk);
      const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
      const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
      for (int p = 0; p < LA; ++p) {
        const int m = m0 + (lw + p * LW) * 8 + prow;
        if (m < a.M) {
          const int n = m / HoWo, r = m - n * HoWo;
          const int ho = r / a.Wo, wo = r - ho * a.Wo;
          ahi[p] = ho * a.sh - a.ph;
          awi[p] = wo * a.sw - a.pw;
          abase[p] = ((n * a.H + ahi[p]) * a.W + awi[p]) * a.ldx;
        } else {
          ahi[p] = -(1 << 24);
          awi[p] = 0;
          abase[p] = 0;
        }
      }
      wsrc = a.w + (size_t)(n0 + lw * 8 + prow) * a.wrs + lchunk * 16;
    };
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    int it_i = 0, kt_i = 0, st_i = 0;  // issue cursor: tile ordinal, K tile inside it, ring stage
    setup(0);
    auto issue = [&]() {
      const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(st_i * STAGE) + (unsigned)(lw * 1024));
      const int4 e = sK[kt_i * 8 + lchunk];
#pragma unroll
      for (int p = 0; p < LA; ++p) {
        const int hi = ahi[p] + e.y, wi = awi[p] + e.z;
        const bool ok = e.w && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const char* src = ok ? a.x + (size_t)(abase[p] + e.x) * ES : a.zero;
        glds16(src, sbase + p * (LW * 1024));
      }
#pragma unroll
      for (int p = 0; p < LB; ++p)
        glds16(wsrc + (size_t)(p * LW * 8) * a.wrs + (size_t)kt_i * a.wts, sbase + BM * 128 + p * (LW * 1024));
      st_i = st_i + 1 == S ? 0 : st_i + 1;
      if (++kt_i == nkt) {
        kt_i = 0;
        if (++it_i < ntile) setup(it_i);
      }
    };
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
      if (t < Q) issue();
    for (int q = 0; q < Q; ++q) {
      if (q + S - 2 < Q)
        wait_dma_and_barrier<(S - 2) * (LA + LB)>();
      else
        wait_dma_and_barrier<0>();
      if (q + S - 1 < Q) issue();
    }
    return;
  }

  // -------------------------------------------------------------------- consumer wave
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fgrp = lane >> 4;
  // no staging epilogue here: the 4 x 8 wave tile double-buffers too (2-byte dtypes; the planar hi/lo sets would spill)
  constexpr bool PIPE = (TM + TN) < 12 || ((TM + TN) == 12 && !is_planar<T>::value);
  constexpr int NF = PIPE ? 2 : 1;
  typedef typename std::conditional<is_planar<T>::value, _Float16, T>::type elem_t;
  typedef elem_t tx4 __attribute__((ext_vector_type(4)));
  const bool relu = a.act == ACT_RELU, prelu = a.act == ACT_PRELU;
  int st = 0;
  for (int it = 0; it < ntile; ++it) {
    const int bid = xcd_remap((int)blockIdx.x + it * G, a.nblk);
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 rres[RES ? TM : 1][RES ? TN / 2 : 1];
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jp = 0; jp < TN / 2; ++jp) {
          const int c = n0 + wn * WTN + (2 * jp + (fgrp & 1)) * 16 + (fgrp >> 1) * 8;
          const int m = min(m0 + wm * WTM + i * 16 + frow, a.M - 1);
          rres[i][jp] = *reinterpret_cast<const uint4*>(a.res + ((size_t)m * a.ldres + (c < a.Cout ? c : 0)) * ES);
        }
    }
    uint4 xf[NF][TM], wf[NF][TN];
    auto read_frags = [&](int ks) {
      const int fs = PIPE ? ks : 0;
      const char* sA = smem + st * STAGE;
      const char* sB = sA + BM * 128;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 16 + frow;
        xf[fs][i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 16 + frow;
        wf[fs][j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
    };
    auto mma = [&](int ks) {
      const int fs = PIPE ? ks : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[fs][j], xf[fs][i]);
    };
    auto join = [&]() {
      // no vmcnt here: this wave's only vector-memory operations are the previous tile's stores, which may still drain
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    if constexpr (is_planar<T>::value) {
      // planar split-f16 (see conv_igemm.hip): half 0 / 1 of a K tile = hi / lo plane of the same 32 k values
      uint4 xh[NF][TM], wh[NF][TN], xl[TM], wl[TN];
      auto rd = [&](int ks, uint4 (&x)[TM], uint4 (&w)[TN]) {
        const char* sA = smem + st * STAGE;
        const char* sB = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WTM + i * 16 + frow;
          x[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * WTN + j * 16 + frow;
          w[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
        }
      };
      auto cross = [&](uint4 (&x)[TM], uint4 (&w)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_cross(acc[i][j], w[j], wl[j], x[i], xl[i]);
      };
      auto hh = [&](uint4 (&x)[TM], uint4 (&w)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_hh(acc[i][j], w[j], x[i]);
      };
      auto next_stage = [&]() { st = st + 1 == S ? 0 : st + 1; };
      if constexpr (PIPE) {
        auto step = [&](auto P, bool first) {
          constexpr int c = decltype(P)::value;
          join();
          rd(0, xh[c], wh[c]);
          if (!first) cross(xh[c ^ 1], wh[c ^ 1]);
          rd(1, xl, wl);
          hh(xh[c], wh[c]);
          next_stage();
        };
        int kt = 0;
        for (; kt + 1 < nkt; kt += 2) {
          step(std::integral_constant<int, 0>{}, kt == 0);
          step(std::integral_constant<int, 1>{}, false);
        }
        if (kt < nkt) {
          step(std::integral_constant<int, 0>{}, kt == 0);
          cross(xh[0], wh[0]);
        } else {
          cross(xh[NF - 1], wh[NF - 1]);
        }
      } else {
        for (int kt = 0; kt < nkt; ++kt) {
          join();
          rd(0, xh[0], wh[0]);
          rd(1, xl, wl);
          hh(xh[0], wh[0]);
          cross(xh[0], wh[0]);
          next_stage();
        }
      }
    } else {
    bool pend = false;
    for (int kt = 0; kt < nkt; ++kt) {
      join();
      if constexpr (PIPE) {
        read_frags(0);
        if (pend) mma(1);
        pend = BKE / 2 < a.K - kt * BKE;
        if (pend) read_frags(1);
        mma(0);
      } else {
        read_frags(0);
        mma(0);
        if (BKE / 2 < a.K - kt * BKE) {
          read_frags(1);
          mma(1);
        }
      }
      st = st + 1 == S ? 0 : st + 1;
    }
    if (pend) mma(1);
    }

    // register epilogue: tiles j (even) and j+1 of a lane-row pair are exchanged so each lane holds 8 channels
#pragma unroll
    for (int j = 0; j < TN; j += 2) {
      const int cq = n0 + wn * WTN + j * 16 + fgrp * 4;  // this lane's channel quad in tile j; tile j+1: +16
      const f32x4_t b0 = cq < a.Cout ? *reinterpret_cast<const f32x4_t*>(a.bias + cq) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      const f32x4_t b1 =
          cq + 16 < a.Cout ? *reinterpret_cast<const f32x4_t*>(a.bias + cq + 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
      if (prelu) {
        if (cq < a.Cout) s0 = *reinterpret_cast<const f32x4_t*>(a.slope + cq);
        if (cq + 16 < a.Cout) s1 = *reinterpret_cast<const f32x4_t*>(a.slope + cq + 16);
      }
      const int c = n0 + wn * WTN + (j + (fgrp & 1)) * 16 + (fgrp >> 1) * 8;  // the 8 channels it stores
      int sg = 0;
#pragma unroll
      for (int s2 = 1; s2 < 4; ++s2)
        if (s2 < a.nseg && c >= a.seg_c0[s2]) sg = s2;
      char* const dcol = a.seg_ptr[sg] + (size_t)(c - a.seg_c0[sg]) * ES;
      const int dld = a.seg_ld[sg];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = acc[i][j][e] + b0[e];
          v1[e] = acc[i][j + 1][e] + b1[e];
        }
        if constexpr (RES) {
          // the chunk this lane will store holds, swapped, the residual of its two accumulator quads
          const uint4 r = rres[i][j / 2];
          const auto rx = __builtin_amdgcn_permlane16_swap(r.x, r.z, false, false);
          const auto ry = __builtin_amdgcn_permlane16_swap(r.y, r.w, false, false);
          const tx4 q0 = __builtin_bit_cast(tx4, uint2{rx[0], ry[0]}), q1 = __builtin_bit_cast(tx4, uint2{rx[1], ry[1]});
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] += (float)q0[e]; v1[e] += (float)q1[e]; }
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
        }
        if (prelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = v0[e] > 0.f ? v0[e] : v0[e] * s0[e];
            v1[e] = v1[e] > 0.f ? v1[e] : v1[e] * s1[e];
          }
        }
        const int m = m0 + wm * WTM + i * 16 + frow;
        if constexpr (is_planar<T>::value) {
          // an 8-channel unit is [8 hi][8 lo]: the hi quads and the lo quads are exchanged separately
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          h4 h0, l0, h1, l1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const sf16 s0(v0[e]), s1(v1[e]);
            h0[e] = s0.hi; l0[e] = s0.lo;
            h1[e] = s1.hi; l1[e] = s1.lo;
          }
          const uint2 ph0 = __builtin_bit_cast(uint2, h0), ph1 = __builtin_bit_cast(uint2, h1);
          const uint2 pl0 = __builtin_bit_cast(uint2, l0), pl1 = __builtin_bit_cast(uint2, l1);
          const auto hx = __builtin_amdgcn_permlane16_swap(ph0.x, ph1.x, false, false);
          const auto hy = __builtin_amdgcn_permlane16_swap(ph0.y, ph1.y, false, false);
          const auto lx = __builtin_amdgcn_permlane16_swap(pl0.x, pl1.x, false, false);
          const auto ly = __builtin_amdgcn_permlane16_swap(pl0.y, pl1.y, false, false);
          if (m < a.M && c < a.Cout) {
            char* d = dcol + (size_t)m * dld * ES;
            *reinterpret_cast<uint4*>(d) = uint4{hx[0], hy[0], hx[1], hy[1]};
            *reinterpret_cast<uint4*>(d + 16) = uint4{lx[0], ly[0], lx[1], ly[1]};
          }
        } else {
          tx4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o0[e] = (elem_t)v0[e]; o1[e] = (elem_t)v1[e]; }
          const uint2 p0 = __builtin_bit_cast(uint2, o0), p1 = __builtin_bit_cast(uint2, o1);
          // odd lane rows of p0 <-> even lane rows of p1
          const auto sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
          const auto sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
          if (m < a.M && c < a.Cout)
            *reinterpret_cast<uint4*>(dcol + (size_t)m * dld * ES) = uint4{sx[0], sy[0], sx[1], sy[1]};
        }
      }
    }
  }
}

// ===================================================================== host side
struct WsCfg { int bm, bn, wm, wn, s, lw; };
static const WsCfg kWs[] = {
    {256, 128, 2, 2, 3, 4}, {256, 128, 4, 1, 3, 4}, {128, 128, 2, 2, 3, 4}, {128, 128, 2, 2, 4, 4},
    {256, 64, 4, 1, 3, 4},  {128, 192, 2, 2, 3, 4}, {128, 256, 2, 2, 3, 4}, {128, 64, 2, 2, 4, 4},
    {256, 128, 2, 2, 3, 2}, {128, 128, 2, 2, 4, 2}, {128, 192, 2, 2, 3, 2}, {192, 128, 2, 2, 3, 4},
    // 160-row tiles: the 17x17 layers' 73,984 pixels are 578 tiles of 128 = 2.26 rounds over 256 CUs (a third round for a
    // quarter of them) but 463 tiles of 160 = 1.81
    {160, 256, 2, 2, 3, 4}, {160, 192, 2, 2, 3, 4},
    // small tiles for the 3x3-pixel tail (M = 9 rows per image): persistent, so their 6-28 K tiles per output tile run
    // back to back instead of each paying a pipeline fill and an epilogue
    {64, 64, 2, 2, 4, 4},   {64, 128, 2, 2, 4, 4},  {32, 64, 2, 2, 6, 4},   {32, 128, 2, 2, 4, 4},  {64, 64, 2, 2, 8, 4},
};
constexpr int kNumWs = (int)(sizeof(kWs) / sizeof(kWs[0]));

int ws_num_cfgs() { return kNumWs; }

bool ws_cfg_ok(const ConvArgs& a, int wcfg) {
  if (wcfg < 0 || wcfg >= kNumWs) return false;
  const WsCfg& c = kWs[wcfg];
  if (a.Cout <= c.bn / 2) return false;
  if (((a.Cout + c.bn - 1) / c.bn) * c.bn > a.cout_pad) return false;
  if ((c.bn % 64) && a.Cout % c.bn) return false;
  const int lds = c.s * (c.bm + c.bn) * 128 + (a.Kpad / (128 / dtype_size(a.dtype))) * 8 * 16;
  return lds <= 160 * 1024;
}

template <typename T, int BM, int BN, int WM, int WN, int S, int LW>
static hipError_t launch_one_tile(const KArgs& k, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_ws_kernel<T, BM, BN, WM, WN, S, LW>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  KArgs kk = k;
  const int lds = S * (BM + BN) * 128 + k.nkt * 8 * 16;
  const int tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.Cout + BN - 1) / BN;
  kk.nblk = tiles_m * kk.tiles_n;
  hipLaunchKernelGGL((conv_igemm_ws_kernel<T, BM, BN, WM, WN, S, LW>), dim3(kk.nblk), dim3((WM * WN + LW) * 64), lds, s,
                     kk);
  return hipGetLastError();
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    n = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return n;
}

// the persistent form (bias + ReLU layers of the 2-byte plans; VNF_WS_PERSIST=0 keeps one tile per workgroup)
static bool ws_persistent(const KArgs& k) {
  const char* e = getenv("VNF_WS_PERSIST");   // read per launch: the parity test flips it inside one process
  return !(e && atoi(e) == 0) && k.ncls == 1 && !k.out_f32;
}

template <typename T, int BM, int BN, int WM, int WN, int S, int LW, bool RES>
static hipError_t launch_persistent(const KArgs& k, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_wsp_kernel<T, BM, BN, WM, WN, S, LW, RES>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  KArgs kk = k;
  const int lds = S * (BM + BN) * 128 + k.nkt * 8 * 16;
  kk.tiles_n = (k.Cout + BN - 1) / BN;
  kk.nblk = ((k.M + BM - 1) / BM) * kk.tiles_n;
  // one workgroup per CU per LDS-resident copy; whole rounds of 8 workgroups keep the blockIdx -> XCD map of xcd_remap
  // (tile t runs on XCD t % 8)
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int grid = kk.nblk < cu_count() * per_cu ? kk.nblk : (cu_count() * per_cu) & ~7;
  if (grid < 1) grid = kk.nblk;
  hipLaunchKernelGGL((conv_igemm_wsp_kernel<T, BM, BN, WM, WN, S, LW, RES>), dim3(grid), dim3((WM * WN + LW) * 64), lds, s,
                     kk);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN, int S, int LW>
static hipError_t launch_one(const KArgs& k, hipStream_t s) {
  if constexpr ((sizeof(T) == 2 || is_planar<T>::value) && (BN / WN / 16) % 2 == 0) {
    if (ws_persistent(k)) {
      if (!k.res) return launch_persistent<T, BM, BN, WM, WN, S, LW, false>(k, s);
      // residual chunks live in registers for the whole K loop: small tiles only
      if constexpr (sizeof(T) == 2 && (BM / WM / 16) * (BN / WN / 32) <= 8)
        return launch_persistent<T, BM, BN, WM, WN, S, LW, true>(k, s);
    }
  }
  return launch_one_tile<T, BM, BN, WM, WN, S, LW>(k, s);
}

template <typename T>
static hipError_t launch_ws_typed(int wcfg, const KArgs& k, hipStream_t s) {
  switch (wcfg) {
    case 0: return launch_one<T, 256, 128, 2, 2, 3, 4>(k, s);
    case 1: return launch_one<T, 256, 128, 4, 1, 3, 4>(k, s);
    case 2: return launch_one<T, 128, 128, 2, 2, 3, 4>(k, s);
    case 3: return launch_one<T, 128, 128, 2, 2, 4, 4>(k, s);
    case 4: return launch_one<T, 256, 64, 4, 1, 3, 4>(k, s);
    case 5: return launch_one<T, 128, 192, 2, 2, 3, 4>(k, s);
    case 6: return launch_one<T, 128, 256, 2, 2, 3, 4>(k, s);
    case 7: return launch_one<T, 128, 64, 2, 2, 4, 4>(k, s);
    case 8: return launch_one<T, 256, 128, 2, 2, 3, 2>(k, s);
    case 9: return launch_one<T, 128, 128, 2, 2, 4, 2>(k, s);
    case 10: return launch_one<T, 128, 192, 2, 2, 3, 2>(k, s);
    case 11: return launch_one<T, 192, 128, 2, 2, 3, 4>(k, s);
    case 12: return launch_one<T, 160, 256, 2, 2, 3, 4>(k, s);
    case 13: return launch_one<T, 160, 192, 2, 2, 3, 4>(k, s);
    case 14: return launch_one<T, 64, 64, 2, 2, 4, 4>(k, s);
    case 15: return launch_one<T, 64, 128, 2, 2, 4, 4>(k, s);
    case 16: return launch_one<T, 32, 64, 2, 2, 6, 4>(k, s);
    case 17: return launch_one<T, 32, 128, 2, 2, 4, 4>(k, s);
    case 18: return launch_one<T, 64, 64, 2, 2, 8, 4>(k, s);
  }
  return hipErrorInvalidValue;
}

// Instrumented launch (tools/stamp_ws.py): VNF_WS_STAMP=<file> dumps per-K-tile s_memtime stamps of workgroup 600
// of every bf16 {128,192,2,2,3,4} launch with more than 600 workgroups.
static hipError_t launch_stamped(const KArgs& k, hipStream_t s) {
  static long long* dbuf = nullptr;
  const int n = 8 * 64 * 4;
  if (!dbuf && hipMalloc((void**)&dbuf, n * 8) != hipSuccess) return hipErrorOutOfMemory;
  (void)hipMemsetAsync(dbuf, 0, n * 8, s);
  KArgs kk = k;
  kk.dbg = dbuf;
  const int lds = 3 * (128 + 192) * 128 + k.nkt * 8 * 16;
  const int tiles_m = (k.M + 127) / 128;
  kk.tiles_n = (k.Cout + 191) / 192;
  kk.nblk = tiles_m * kk.tiles_n;
  (void)hipFuncSetAttribute((const void*)conv_igemm_ws_kernel<__bf16, 128, 192, 2, 2, 3, 4, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_igemm_ws_kernel<__bf16, 128, 192, 2, 2, 3, 4, true>), dim3(kk.nblk), dim3(512), lds, s, kk);
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  static long long host[8 * 64 * 4];
  (void)hipMemcpy(host, dbuf, n * 8, hipMemcpyDeviceToHost);
  if (FILE* f = fopen(getenv("VNF_WS_STAMP"), "a")) {
    fprintf(f, "launch M=%d N=%d nkt=%d nblk=%d\n", k.M, k.Cout, k.nkt, kk.nblk);
    for (int w = 0; w < 8; ++w)
      for (int kt = 0; kt < k.nkt && kt < 64; ++kt)
        fprintf(f, "%d %d %lld %lld %lld %lld\n", w, kt, host[(w * 64 + kt) * 4], host[(w * 64 + kt) * 4 + 1],
                host[(w * 64 + kt) * 4 + 2], host[(w * 64 + kt) * 4 + 3]);
    fclose(f);
  }
  return hipSuccess;
}

hipError_t launch_ws(const ConvArgs& a, const KArgs& k, int wcfg, hipStream_t s) {
  if (!ws_cfg_ok(a, wcfg) || !k.zero) return hipErrorInvalidValue;
  if (getenv("VNF_WS_STAMP") && wcfg == 5 && a.dtype == BF16 && (k.M + 127) / 128 * ((k.Cout + 191) / 192) > 600)
    return launch_stamped(k, s);
  switch (a.dtype) {
    case BF16: return launch_ws_typed<__bf16>(wcfg, k, s);
    case F16: return launch_ws_typed<_Float16>(wcfg, k, s);
    case F32: return launch_ws_typed<float>(wcfg, k, s);
    case F16X2: return launch_ws_typed<sf16>(wcfg, k, s);
    case F16P: return launch_ws_typed<pf16>(wcfg, k, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace vnf
