// Implicit-GEMM convolution for gfx950 (MI355X), NHWC activations, MFMA 16x16 tiles.
//
// GEMM view:  C[m][co] = sum_k A[m][k] * Wt[co][k]
//   m  = output pixel (n, ho, wo)          -> "pixel" axis, BM per workgroup
//   co = output channel                     -> "channel" axis, BN per workgroup
//   k  = (kh, kw, c) flattened, c fastest  -> walked in 128-byte K tiles
// A is never materialised: every 16-byte k-chunk (8 x 16-bit or 4 x f32 channels of one filter tap)
// is gathered straight from the NHWC input (tap and channel of each lane's chunk are tracked
// incrementally; the patch / wave-specialised / fallback kernels use the per-chunk offset table
// ktab); padded borders and rows past M read a 16-byte zero page instead.
//
// Main kernel (conv_igemm_dma_kernel): both operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, one 1-KiB piece = 8 rows x 128 B per wave instruction), never through
// VGPRs, into a ring of S stages.  The per-lane SOURCE address carries both the im2col gather and
// the bank swizzle (the LDS image must be lane-linear: physical 16-byte slot p of row r holds
// logical k-chunk p ^ (r & 7)), so the ds_read_b128 fragment reads are conflict-free on CDNA4's
// 64-bank LDS (checked exhaustively).  S-1 tiles are in flight under the MFMAs of the
// current one: counted s_waitcnt vmcnt(N) + one raw s_barrier per K tile, no VGPR staging cost,
// so bytes-in-flight per CU -- what bounds these small-K, small-N problems -- is set by LDS
// capacity (160 KiB) instead of by register pressure.
//
// Fallback kernel (conv_igemm_kernel): register-staged double buffering (VNF_CONV_REG=1; the
// first version of the core, kept as a cross-check).
//
// The MFMA is issued with the WEIGHT fragment as the A operand and the ACTIVATION fragment as
// the B operand, so an accumulator register quad holds 4 consecutive output channels of one
// pixel.  The epilogue stages fp32 accumulators through LDS and writes whole NHWC rows with
// 16-byte stores: bias (folded BatchNorm; optionally one of 9 border classes), residual add,
// ReLU / PReLU, conversion, and routing of column ranges to different destination tensors
// (concat-free inception branches) all happen there.
//
// dtype paths: bf16 / f16 -> v_mfma_f32_16x16x32_{bf16,f16}; f32 -> v_mfma_f32_16x16x4_f32
// (exact f32 FMA chain; the <=1e-4 parity path).
#include <cstdlib>
#include <type_traits>

#include "block35.h"
#include "conv_device.h"

namespace vnf {

// per-thread pixel-row bookkeeping: thread (tid>>3) owns rows lrow + RS*p of the A tile
template <int AP, int RS>
__device__ __forceinline__ void row_setup(const KArgs& a, int m0, int lrow, int (&abase)[AP], int (&ahi)[AP],
                                          int (&awi)[AP]) {
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + lrow + RS * p;
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
}

// fragment reads + MFMAs of one K tile held in LDS (A rows at sA, weight rows at sB)
template <typename T, int TM, int TN>
__device__ __forceinline__ void tile_mma(const char* sA, const char* sB, int arow0, int brow0, int frow, int fgrp,
                                         int kleft, f32x4_t (&acc)[TM][TN]) {
  constexpr int BKE = 128 / (int)sizeof(T);
  if constexpr (is_planar<T>::value) {
    uint4 xh[TM], xl[TM], wh[TN], wl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = arow0 + i * 16 + frow;
      xh[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + ((fgrp ^ (row & 7)) << 4));
      xl[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((4 + fgrp) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = brow0 + j * 16 + frow;
      wh[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + ((fgrp ^ (row & 7)) << 4));
      wl[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((4 + fgrp) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        mma_hh(acc[i][j], wh[j], xh[i]);
        mma_cross(acc[i][j], wh[j], wl[j], xh[i], xl[i]);
      }
    return;
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if (ks * (BKE / 2) < kleft) {
      uint4 xf[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = arow0 + i * 16 + frow;
        xf[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = brow0 + j * 16 + frow;
        wf[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[j], xf[i]);
    }
  }
}


template <typename T, int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_dma_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int CH = 16 / ES, BKE = 128 / ES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int NW = WM * WN, NT = NW * 64, RS = NT / 8;  // waves, threads, tile rows covered per DMA pass
  constexpr int AP = BM / RS, BP = BN / RS, L = AP + BP;  // DMA pieces per wave per K tile
  constexpr int STAGE = (BM + BN) * 128;
  static_assert((NW == 4 || NW == 8) && S >= 3 && BM % RS == 0 && BN % RS == 0, "4 or 8 waves, ring of >= 3 stages");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int bid = xcd_remap(blockIdx.x, a.nblk);
  const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = a.nkt;

  const int lrow = tid >> 3, lcol = tid & 7;
  const int lchunk = lcol ^ (lrow & 7);  // logical k-chunk this lane fetches (physical slot = lcol)
  int abase[AP], ahi[AP], awi[AP];
  row_setup<AP, RS>(a, m0, lrow, abase, ahi, awi);
  const char* wsrc = a.w + (size_t)(n0 + lrow) * a.wrs + lchunk * 16;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);  // LDS byte address of the ring

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // This lane's k-chunk of the next K tile to issue, as (tap row, tap column, channel): advanced by one K tile per
  // issue() -- issue() is called with consecutive kt -- instead of looked up in the layer's gather table, so the
  // prologue has no table load and no barrier before the first DMA (the table's round trip was ~1.5 k cycles of
  // every workgroup; most launches of the inception blocks only live ~10 k).
  int g_c = chunk_chan<T>(lchunk), g_kh = 0, g_kw = 0;
  const int g_byte = chunk_byte<T>(lchunk);  // planar split-f16: the lo plane sits 16 bytes into its 8-channel unit
  auto g_norm = [&]() {
    while (g_c >= a.Cin) {
      g_c -= a.Cin;
      if (++g_kw == a.KW) { g_kw = 0; ++g_kh; }
    }
  };
  g_norm();
  auto issue = [&](int kt) {
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((kt % S) * STAGE) + (unsigned)(wave * 1024));
    const bool kvalid = g_kh < a.KH;  // k < K
    const int ex = (g_kh * a.W + g_kw) * a.ldx + g_c;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int hi = ahi[p] + g_kh, wi = awi[p] + g_kw;
      const bool ok = kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const char* src = ok ? a.x + (size_t)(abase[p] + ex) * ES + g_byte : a.zero;
      glds16(src, sbase + p * (NW * 1024));
    }
#pragma unroll
    for (int p = 0; p < BP; ++p)
      glds16(wsrc + (size_t)(RS * p) * a.wrs + (size_t)kt * a.wts, sbase + BM * 128 + p * (NW * 1024));
    g_c += BKE;
    g_norm();
  };

#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nkt) issue(t);

  const int frow = lane & 15, fgrp = lane >> 4;
  // Software-pipelined over half K tiles: the fragment reads of one half are in flight under the
  // MFMAs of the previous half (also across the barrier), so a wave never idles on LDS latency
  // even with one or two waves per SIMD.
  uint4 xf[2][TM], wf[2][TN];
  auto read_frags = [&](int kt, int ks) {
    const char* sA = smem + (kt % S) * STAGE;
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WTM + i * 16 + frow;
      xf[ks][i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * WTN + j * 16 + frow;
      wf[ks][j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
    }
  };
  auto mma = [&](int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[ks][j], xf[ks][i]);
  };
  auto sync_tile = [&](int kt) {
    // tile kt has landed once at most the S-2 younger tiles of this wave are still in flight
    if (kt + S - 2 < nkt)
      wait_dma_and_barrier<(S - 2) * L>();
    else
      wait_dma_and_barrier<0>();
    // every wave's pieces of tile kt have landed; stage (kt-1)%S is free for tile kt+S-1
    if (kt + S - 1 < nkt) issue(kt + S - 1);
  };
  if constexpr (is_planar<T>::value) {
    // planar split-f16: "half" 0 of a K tile is the hi plane, half 1 the lo plane of the same 32 k values.  Per tile
    // hi.hi' (needs the hi fragments only) and the two cross terms; the cross terms of tile kt-1 run under the hi reads of
    // tile kt, so the hi fragments are double-buffered (static indices: the loop is unrolled by two).
    uint4 xh[2][TM], wh[2][TN];
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
    auto step = [&](auto P, int kt) {
      constexpr int c = decltype(P)::value;
      sync_tile(kt);
      rd(kt, 0, xh[c], wh[c]);
      if (kt > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_cross(acc[i][j], wh[c ^ 1][j], wf[1][j], xh[c ^ 1][i], xf[1][i]);
      }
      rd(kt, 1, xf[1], wf[1]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_hh(acc[i][j], wh[c][j], xh[c][i]);
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
      step(std::integral_constant<int, 0>{}, kt);
      step(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < nkt) {
      step(std::integral_constant<int, 0>{}, kt);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_cross(acc[i][j], wh[0][j], wf[1][j], xh[0][i], xf[1][i]);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_cross(acc[i][j], wh[1][j], wf[1][j], xh[1][i], xf[1][i]);
    }
  } else {
    bool pend = false;  // second half of the previous K tile read but not yet multiplied
    for (int kt = 0; kt < nkt; ++kt) {
      sync_tile(kt);
      read_frags(kt, 0);
      if (pend) mma(1);
      pend = BKE / 2 < a.K - kt * BKE;
      if (pend) read_frags(kt, 1);
      mma(0);
    }
    if (pend) mma(1);
  }
  __syncthreads();
  conv_epilogue<T, BM, BN, WM, WN, S * STAGE>(a, acc, smem, m0, n0);
}

// ===================================================================== register-staged fallback
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int CH = 16 / ES, BKE = 128 / ES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(WM * WN == 4, "4 waves per workgroup");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bid = xcd_remap(blockIdx.x, a.nblk);
  const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = tid >> 3, lcol = tid & 7;
  int abase[AP], ahi[AP], awi[AP];
  row_setup<AP, 32>(a, m0, lrow, abase, ahi, awi);
  const char* wrow = a.w + ((size_t)(n0 + lrow) * a.Kpad + lcol * CH) * ES;

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[AP], rb[BP];
  auto gload = [&](int kt) {
    const int4 e = a.ktab[kt * 8 + lcol];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int hi = ahi[p] + e.y, wi = awi[p] + e.z;
      const bool ok = e.w && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      ra[p] = ok ? *reinterpret_cast<const uint4*>(a.x + (size_t)(abase[p] + e.x) * ES) : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int p = 0; p < BP; ++p)
      rb[p] = *reinterpret_cast<const uint4*>(wrow + ((size_t)(32 * p) * a.Kpad + (size_t)kt * BKE) * ES);
  };
  auto lstore = [&](int st) {
    char* sA = smem + st * STAGE;
    char* sB = sA + BM * 128;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int row = lrow + 32 * p;
      *reinterpret_cast<uint4*>(sA + row * 128 + ((lcol ^ (row & 7)) << 4)) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int row = lrow + 32 * p;
      *reinterpret_cast<uint4*>(sB + row * 128 + ((lcol ^ (row & 7)) << 4)) = rb[p];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 15, fgrp = lane >> 4;
  for (int kt = 0; kt < a.nkt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < a.nkt) gload(kt + 1);
    const char* sA = smem + st * STAGE;
    tile_mma<T, TM, TN>(sA, sA + BM * 128, wm * WTM, wn * WTN, frow, fgrp, a.K - kt * BKE, acc);
    if (kt + 1 < a.nkt) lstore(st ^ 1);
    __syncthreads();
  }
  conv_epilogue<T, BM, BN, WM, WN, 2 * STAGE>(a, acc, smem, m0, n0);
}

// ===================================================================== host launch
static const char* zero_page() {  // per-device 256 zero bytes for padded / out-of-range gather sources
  static char* z[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!z[dev]) {
    if (hipMalloc((void**)&z[dev], 256) != hipSuccess) return nullptr;
    (void)hipMemset(z[dev], 0, 256);
    (void)hipDeviceSynchronize();  // once per device: the page must be zero before any stream reads it
  }
  return z[dev];
}

const char* conv_zero_page() { return zero_page(); }

template <typename T, int BM, int BN, int WM, int WN, int S>
static hipError_t launch_dma(const KArgs& k, hipStream_t s) {
  constexpr int ring = S * (BM + BN) * 128;
  static bool attr_done = false;
  KArgs kk = k;
  const int lds = ring;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<T, BM, BN, WM, WN, S>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  const int tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.Cout + BN - 1) / BN;
  kk.nblk = tiles_m * kk.tiles_n;
  hipLaunchKernelGGL((conv_igemm_dma_kernel<T, BM, BN, WM, WN, S>), dim3(kk.nblk), dim3(WM * WN * 64), lds, s, kk);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN>
static hipError_t launch_reg(const KArgs& k, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 128;
  KArgs kk = k;
  const int tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.Cout + BN - 1) / BN;
  kk.nblk = tiles_m * kk.tiles_n;
  hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN>), dim3(kk.nblk), dim3(256), lds, s, kk);
  return hipGetLastError();
}

// Tile configurations of the LDS-DMA kernel: {BM, BN, waves along M, waves along N, ring stages}.
struct TileCfg { int bm, bn, wm, wn, s; };
static const TileCfg kCfgs[] = {
    {128, 128, 2, 2, 3}, {128, 64, 2, 2, 3}, {128, 32, 4, 1, 3}, {64, 64, 2, 2, 4}, {64, 32, 2, 2, 4},
    {256, 32, 4, 1, 3},  {256, 64, 4, 1, 3}, {64, 128, 1, 4, 3}, {32, 64, 2, 2, 4}, {32, 128, 1, 4, 4},
    {128, 64, 2, 2, 4},  {64, 64, 2, 2, 3},  {128, 32, 4, 1, 4},
    // 8-wave workgroups: two waves per SIMD, so one wave's DMA issue / LDS reads hide under the other's MFMAs
    {128, 128, 2, 4, 3}, {256, 128, 4, 2, 3}, {256, 64, 4, 2, 3}, {128, 64, 4, 2, 3}, {128, 256, 2, 4, 3},
    {128, 128, 2, 4, 4},
    // deep rings for problems with about one workgroup per CU: the K loop is a serial chain of
    // L2 / Infinity-Cache round trips, so more tiles in flight shorten it directly
    {128, 64, 4, 2, 6},  {64, 64, 2, 2, 8},   {64, 32, 2, 2, 8},  {32, 64, 2, 2, 8},  {128, 32, 4, 1, 6},
    {64, 128, 2, 4, 5},  {128, 64, 2, 2, 6},
    // whole-N tiles: the activation operand (which comes from beyond L2) is read exactly once
    {64, 256, 1, 8, 3},  {64, 256, 1, 8, 4},  {128, 256, 2, 4, 3},
    // 96- and 192-channel tiles: the inception widths, so one tile spans the whole N and the gathered
    // activation operand crosses the L2 -> LDS path once per tap instead of once per tap and N tile
    {128, 96, 4, 1, 3},  {64, 96, 2, 2, 4},   {128, 192, 2, 4, 3}, {64, 192, 2, 4, 3},  {256, 96, 4, 1, 3},
    {64, 192, 2, 4, 4},
};
constexpr int kNumCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

int conv_num_cfgs() { return kNumCfgs + patch_num_cfgs() + ws_num_cfgs(); }

static bool dma_capable(const ConvArgs& a) { return (a.Kpad / (128 / dtype_size(a.dtype))) * 8 * 16 <= 24 * 1024; }

bool conv_cfg_ok(const ConvArgs& a, int cfg) {
  if (cfg >= kNumCfgs + patch_num_cfgs()) return dma_capable(a) && ws_cfg_ok(a, cfg - kNumCfgs - patch_num_cfgs());
  if (cfg >= kNumCfgs) return dma_capable(a) && patch_cfg_ok(a, cfg - kNumCfgs);
  if (cfg < 0) return false;  // the ring kernel needs no table in LDS: any K
  const TileCfg& c = kCfgs[cfg];
  if (c.bn > 32 && a.Cout <= c.bn / 2) return false;  // more than half the tile would be padding
  if (((a.Cout + c.bn - 1) / c.bn) * c.bn > a.cout_pad) return false;  // weight rows n0..n0+BN-1 must exist in the packed buffer
  if ((c.bn % 64) && a.Cout % c.bn) return false;      // 96/192-wide tiles only where they divide N
  const int lds = c.s * (c.bm + c.bn) * 128;  // the ring (and, after the K loop, the epilogue staging)
  return lds <= 160 * 1024;
}

template <typename T>
static hipError_t launch_cfg(int cfg, const KArgs& k, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_dma<T, 128, 128, 2, 2, 3>(k, s);
    case 1: return launch_dma<T, 128, 64, 2, 2, 3>(k, s);
    case 2: return launch_dma<T, 128, 32, 4, 1, 3>(k, s);
    case 3: return launch_dma<T, 64, 64, 2, 2, 4>(k, s);
    case 4: return launch_dma<T, 64, 32, 2, 2, 4>(k, s);
    case 5: return launch_dma<T, 256, 32, 4, 1, 3>(k, s);
    case 6: return launch_dma<T, 256, 64, 4, 1, 3>(k, s);
    case 7: return launch_dma<T, 64, 128, 1, 4, 3>(k, s);
    case 8: return launch_dma<T, 32, 64, 2, 2, 4>(k, s);
    case 9: return launch_dma<T, 32, 128, 1, 4, 4>(k, s);
    case 10: return launch_dma<T, 128, 64, 2, 2, 4>(k, s);
    case 11: return launch_dma<T, 64, 64, 2, 2, 3>(k, s);
    case 12: return launch_dma<T, 128, 32, 4, 1, 4>(k, s);
    case 13: return launch_dma<T, 128, 128, 2, 4, 3>(k, s);
    case 14: return launch_dma<T, 256, 128, 4, 2, 3>(k, s);
    case 15: return launch_dma<T, 256, 64, 4, 2, 3>(k, s);
    case 16: return launch_dma<T, 128, 64, 4, 2, 3>(k, s);
    case 17: return launch_dma<T, 128, 256, 2, 4, 3>(k, s);
    case 18: return launch_dma<T, 128, 128, 2, 4, 4>(k, s);
    case 19: return launch_dma<T, 128, 64, 4, 2, 6>(k, s);
    case 20: return launch_dma<T, 64, 64, 2, 2, 8>(k, s);
    case 21: return launch_dma<T, 64, 32, 2, 2, 8>(k, s);
    case 22: return launch_dma<T, 32, 64, 2, 2, 8>(k, s);
    case 23: return launch_dma<T, 128, 32, 4, 1, 6>(k, s);
    case 24: return launch_dma<T, 64, 128, 2, 4, 5>(k, s);
    case 25: return launch_dma<T, 128, 64, 2, 2, 6>(k, s);
    case 26: return launch_dma<T, 64, 256, 1, 8, 3>(k, s);
    case 27: return launch_dma<T, 64, 256, 1, 8, 4>(k, s);
    case 28: return launch_dma<T, 128, 256, 2, 4, 3>(k, s);
    case 29: return launch_dma<T, 128, 96, 4, 1, 3>(k, s);
    case 30: return launch_dma<T, 64, 96, 2, 2, 4>(k, s);
    case 31: return launch_dma<T, 128, 192, 2, 4, 3>(k, s);
    case 32: return launch_dma<T, 64, 192, 2, 4, 3>(k, s);
    case 33: return launch_dma<T, 256, 96, 4, 1, 3>(k, s);
    case 34: return launch_dma<T, 64, 192, 2, 4, 4>(k, s);
  }
  return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_typed(const ConvArgs& a, const KArgs& k, hipStream_t s) {
  static const int env_reg = getenv("VNF_CONV_REG") ? atoi(getenv("VNF_CONV_REG")) : 0;
  if (a.cfg >= kNumCfgs + patch_num_cfgs() && !env_reg && k.zero && conv_cfg_ok(a, a.cfg))
    return launch_ws(a, k, a.cfg - kNumCfgs - patch_num_cfgs(), s);
  if (a.cfg >= kNumCfgs && !env_reg && k.zero && conv_cfg_ok(a, a.cfg)) return launch_patch(a, k, a.cfg - kNumCfgs, s);
  if (a.cfg >= 0 && !env_reg && k.zero && conv_cfg_ok(a, a.cfg)) return launch_cfg<T>(a.cfg, k, s);
  // heuristic: BN must divide every segment boundary; keep ~2 workgroups per CU when possible
  auto fits = [&](int) { return true; };
  int bn = 128;
  while (bn > 32 && !(fits(bn) && (a.Cout % bn == 0 || a.Cout > 2 * bn))) bn >>= 1;
  if (!fits(bn)) return hipErrorInvalidValue;
  auto blocks = [&](int bm_, int bn_) { return ((a.M + bm_ - 1) / bm_) * ((a.Cout + bn_ - 1) / bn_); };
  while (bn > 64 && blocks(128, bn) < 512) bn >>= 1;
  int bm = 128;
  if (bn <= 64 && blocks(128, bn) < 512) bm = 64;
  if (!env_reg && k.zero) {
    const int id = bm == 128 ? (bn == 128 ? 0 : bn == 64 ? 1 : 2) : (bn == 64 ? 3 : 4);
    return launch_cfg<T>(id, k, s);
  }
  if (bm == 128 && bn == 128) return launch_reg<T, 128, 128, 2, 2>(k, s);
  if (bm == 128 && bn == 64) return launch_reg<T, 128, 64, 2, 2>(k, s);
  if (bm == 128 && bn == 32) return launch_reg<T, 128, 32, 4, 1>(k, s);
  if (bm == 64 && bn == 64) return launch_reg<T, 64, 64, 2, 2>(k, s);
  if (bm == 64 && bn == 32) return launch_reg<T, 64, 32, 2, 2>(k, s);
  return hipErrorInvalidValue;
}

hipError_t launch_conv(const ConvArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int es = dtype_size(a.dtype);
  const int bke = 128 / es;
  if (a.Cout % 8 || a.Kpad % bke || a.nseg < 1 || a.nseg > 4) return hipErrorInvalidValue;
  KArgs k;
  k.x = (const char*)a.x; k.w = (const char*)a.w; k.bias = a.bias; k.ktab = a.ktab;
  k.res = (const char*)a.res; k.slope = a.slope; k.zero = zero_page();
  k.ldx = a.ldx; k.H = a.H; k.W = a.W; k.Ho = a.Ho; k.Wo = a.Wo;
  k.sh = a.sh; k.sw = a.sw; k.ph = a.ph; k.pw = a.pw;
  k.K = a.K; k.Kpad = a.Kpad; k.nkt = a.Kpad / bke;
  k.wrs = a.Kpad * es; k.wts = 128;
  k.ncls = a.ncls; k.cout_pad = a.cout_pad;
  k.M = a.M; k.Cout = a.Cout; k.tiles_n = 0; k.nblk = 0;
  k.ldres = a.ldres; k.act = a.act; k.out_f32 = a.out_f32;
  k.nseg = a.nseg;
  k.KH = a.KH; k.KW = a.KW; k.Cin = a.Cin; k.pp = k.Wp = k.Hv = k.patch_bytes = k.lds_bytes = 0; k.dbg = nullptr;
  for (int i = 0; i < 4; ++i) {
    k.seg_c0[i] = i < a.nseg ? a.seg[i].c0 : 1 << 30;
    k.seg_c1[i] = i < a.nseg ? a.seg[i].c1 : 1 << 30;
    k.seg_ld[i] = i < a.nseg ? a.seg[i].ld : 0;
    k.seg_ptr[i] = i < a.nseg ? (char*)a.seg[i].ptr : nullptr;
  }
  switch (a.dtype) {
    case BF16: return launch_typed<__bf16>(a, k, s);
    case F16: return launch_typed<_Float16>(a, k, s);
    case F32: return launch_typed<float>(a, k, s);
    case F16X2: return launch_typed<sf16>(a, k, s);
    case F16P: return launch_typed<pf16>(a, k, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace vnf
