// Stride-1 convolution with the input PATCH resident in LDS (gfx950 / MI355X).
//
// The LDS-DMA ring kernel (conv_igemm.hip) gathers every filter tap of every pixel from L1/L2:
// a 3x3 convolution pulls each input byte through the L2 -> LDS path nine times per N tile, and
// that path -- not the MFMA pipe -- bounds the big stem layers (rocprofv3 TCP/TCC counters:
// conv2d_4a sends 46 % of its 23 M L1 accesses on to L2).  Here a workgroup owns BM consecutive
// output pixels and loads the input rows they touch ONCE:
//
//   patch  = virtual input rows [v0, v1] x (W + 2*pw) pixels x Cin channels, pixel-major,
//            pixel pitch PP 16-byte slots (PP = Cin*ES/16 rounded up to == 2 mod 4, which makes
//            the 16 consecutive pixels x 4 k-chunks of one ds_read_b128 fragment hit 64
//            different banks without any swizzle); zero padding is materialised in the patch.
//            "Virtual" rows stack the images with their KH-1 vertical zero rows in between, so a
//            tile may cross image boundaries: row v = n*(H+2*ph) + (iy+ph).
//   weights: [BN rows][128 B] K tiles through the same LDS-DMA ring as conv_igemm.hip.
//
// The activation fragment of pixel p, tap (kh,kw), channel chunk cc sits at
//   patch + ((vb(p)-v0 + kh)*Wp + ox(p) + kw)*PP*16 + cc*16  =  pb[p] + ptab[k-chunk]
// -- one add per fragment read, the table is built from the layer's ktab.  K tiles, MFMA operand
// roles and the epilogue are those of conv_igemm.hip, so results are bit-identical to it.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "conv_device.h"

namespace vnf {

template <typename T, int BM, int BN, int WM, int WN, int S, bool DBG = false>
__global__ __launch_bounds__(WM* WN * 64) void conv_patch_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int CH = 16 / ES, BKE = 128 / ES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int NW = WM * WN, NT = NW * 64, RS = NT / 8;
  constexpr int BPM = (BN + RS - 1) / RS;  // weight DMA pieces per K tile (upper bound per wave)
  constexpr int STAGE = BN * 128;
  constexpr int EPI = patch_epi_bytes(BM, BN, WM);
  static_assert((NW == 4 || NW == 8) && S >= 3 && BPM <= 8, "4 or 8 waves, ring of >= 3 stages");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // instrumented build (tools/stamp_patch.py): cycle stamps of workgroup 700, 40 per wave
  int nstamp = 0;
  auto stamp = [&]() {
    if constexpr (DBG) {
      if (a.dbg && blockIdx.x == 700 && lane == 0 && nstamp < 40) a.dbg[wave * 40 + nstamp] = __builtin_readcyclecounter();
      ++nstamp;
    }
  };
  stamp();
  const int wm = wave / WN, wn = wave % WN;
  const int bid = xcd_remap(blockIdx.x, a.nblk);
  const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = a.nkt;
  const int m_last = min(m0 + BM, a.M) - 1;
  const int g0 = m0 / a.Wo, g1 = m_last / a.Wo;  // global output rows (n*Ho + ho) of this tile
  const int v0 = g0 + (g0 / a.Ho) * (a.KH - 1);
  const int v1 = g1 + (g1 / a.Ho) * (a.KH - 1) + a.KH - 1;
  const int nslots = (v1 - v0 + 1) * a.Wp * a.pp;

  const int ptab_bytes = ((nkt + 1) * 8 * 4 + 15) & ~15;  // one K tile of slack: the loop reads a tile ahead
  int* sP = reinterpret_cast<int*>(smem + S * STAGE);
  char* const zslot = smem + S * STAGE + ptab_bytes;
  const char* const patch = zslot + 16;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const unsigned lds_patch = lds0 + (unsigned)(S * STAGE + ptab_bytes + 16);

  // Planar split-f16: the patch is TWO pixel-major images, the hi planes of every 8-channel unit and -- one whole number
  // of DMA pieces further on -- the lo planes, each with one 16-byte slot per unit: the four lane groups of a fragment
  // read then take consecutive slots, as for the other dtypes (the bank argument above), instead of every second one.
  constexpr bool PL = is_planar<T>::value;
  const int npieces = (nslots + 63) >> 6;          // DMA pieces (1 KiB) per plane
  const int plane_bytes = npieces * 1024;
  // k-chunk -> byte offset inside the patch relative to the pixel's base (-1: K padding)
  // (its first -- usually only -- round of global loads is issued up here and consumed after the DMA burst below)
  const int4 e_first = tid < nkt * 8 ? a.ktab[tid] : int4{0, 0, 0, 0};
  auto fill_table = [&]() {
  for (int i = tid; i < nkt * 8; i += NT) {
    const int4 e = i == tid ? e_first : a.ktab[i];
    const int cc = (e.x - (e.y * a.W + e.z) * a.ldx) / CH;   // 16-byte chunk inside the pixel (planar: 2 * unit + plane)
    const int slot = PL ? (cc >> 1) : cc, plane = PL ? (cc & 1) : 0;
    sP[i] = e.w ? ((e.y * a.Wp + e.z) * a.pp + slot) * 16 + plane * plane_bytes : -1;
  }
  };
  if (tid < 4) reinterpret_cast<int*>(zslot)[tid] = 0;

  // the patch: one 1-KiB DMA piece per wave instruction, lane-linear in LDS
  {
    const int cpb = PL ? a.Cin / 8 : a.Cin / CH;
    for (int pc = wave; pc < (PL ? 2 : 1) * npieces; pc += NW) {
      const int plane = pc >= npieces ? 1 : 0;
      const int s = (pc - plane * npieces) * 64 + lane;
      const char* src = a.zero;
      if (s < nslots) {
        const int p = s / a.pp, c = s - p * a.pp;
        const int py = p / a.Wp, px = p - py * a.Wp;
        const int v = v0 + py, n = v / a.Hv;
        const int iy = v - n * a.Hv - a.ph, ix = px - a.pw;
        if (c < cpb && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          src = a.x + ((size_t)((n * a.H + iy) * a.W + ix) * a.ldx) * ES + (PL ? c * 32 + plane * 16 : c * 16);
      }
      glds16(src, __builtin_amdgcn_readfirstlane(lds_patch + (unsigned)pc * 1024u));
    }
  }

  const int lrow = tid >> 3, lcol = tid & 7;
  const int lchunk = lcol ^ (lrow & 7);
  const char* wsrc = a.w + (size_t)(n0 + lrow) * a.wrs + lchunk * 16;
  int np = 0;  // weight pieces this wave issues per K tile (wave-uniform)
#pragma unroll
  for (int p = 0; p < BPM; ++p)
    if (RS * p + 8 * wave < BN) ++np;

  auto issue = [&](int kt) {
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((kt % S) * STAGE) + (unsigned)(wave * 1024));
#pragma unroll
    for (int p = 0; p < BPM; ++p)
      if (RS * p + 8 * wave < BN)
        glds16(wsrc + (size_t)(RS * p) * a.wrs + (size_t)kt * a.wts, sbase + p * (NW * 1024));
  };
  auto wait_ring = [&]() {  // all but the (S-2)*np youngest pieces of this wave have landed
    switch (np) {
      case 1: wait_dma_and_barrier<(S - 2) * 1>(); break;
      case 2: wait_dma_and_barrier<(S - 2) * 2>(); break;
      case 3: wait_dma_and_barrier<(S - 2) * 3>(); break;
      case 4: wait_dma_and_barrier<(S - 2) * 4>(); break;
      case 5: wait_dma_and_barrier<(S - 2) * 5>(); break;
      case 6: wait_dma_and_barrier<(S - 2) * 6>(); break;
      case 7: wait_dma_and_barrier<(S - 2) * 7>(); break;
      case 8: wait_dma_and_barrier<(S - 2) * 8>(); break;
      default: wait_dma_and_barrier<0>(); break;
    }
  };

#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nkt) issue(t);
  fill_table();  // visible to every wave after the first K tile's barrier
  stamp();

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fgrp = lane >> 4;
  int pb[TM];  // byte offset of this lane's pixel (tap 0, chunk 0) in the patch, per MFMA pixel tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = min(m0 + wm * WTM + i * 16 + frow, m_last);
    const int g = m / a.Wo, ox = m - g * a.Wo;
    const int vb = g + (g / a.Ho) * (a.KH - 1);
    pb[i] = ((vb - v0) * a.Wp + ox) * a.pp * 16;
  }

  // software-pipelined over half K tiles (see conv_igemm.hip); the patch offsets of the next K
  // tile are fetched one tile ahead so no fragment read waits on a table read
  uint4 xf[2][TM], wf[2][TN];
  auto read_frags = [&](int kt, int ks, int t) {
    const char* sB = smem + (kt % S) * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[ks][i] = *reinterpret_cast<const uint4*>(patch + (t >= 0 ? pb[i] + t : -16));
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
  int t0 = 0, t1 = 0;
  auto sync_tile = [&](int kt) {
    // the patch pieces are older than every weight piece, so the first wait covers them too
    if (kt + S - 2 < nkt)
      wait_ring();
    else
      wait_dma_and_barrier<0>();
    if (kt + S - 1 < nkt) issue(kt + S - 1);
    if (kt == 0) { t0 = sP[fgrp]; t1 = sP[4 + fgrp]; }
  };
  if constexpr (is_planar<T>::value) {
    // planar split-f16 (see conv_igemm.hip): half 0 / 1 of a K tile = hi / lo plane; hi fragments double-buffered
    uint4 xh[2][TM], wh[2][TN];
    auto rd = [&](int kt, int ks, int t, uint4 (&x)[TM], uint4 (&w)[TN]) {
      const char* sB = smem + (kt % S) * STAGE;
#pragma unroll
      for (int i = 0; i < TM; ++i) x[i] = *reinterpret_cast<const uint4*>(patch + (t >= 0 ? pb[i] + t : -16));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 16 + frow;
        w[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
      }
    };
    auto step = [&](auto P, int kt) {
      constexpr int c = decltype(P)::value;
      sync_tile(kt);
      rd(kt, 0, t0, xh[c], wh[c]);
      if (kt > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_cross(acc[i][j], wh[c ^ 1][j], wf[1][j], xh[c ^ 1][i], xf[1][i]);
      }
      rd(kt, 1, t1, xf[1], wf[1]);
      t0 = sP[(kt + 1) * 8 + fgrp];  // one row of slack behind the table
      t1 = sP[(kt + 1) * 8 + 4 + fgrp];
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
    bool pend = false;
    for (int kt = 0; kt < nkt; ++kt) {
      stamp();
      sync_tile(kt);
      stamp();
      read_frags(kt, 0, t0);
      if (pend) mma(1);
      pend = BKE / 2 < a.K - kt * BKE;
      if (pend) read_frags(kt, 1, t1);
      t0 = sP[(kt + 1) * 8 + fgrp];  // one row of slack behind the table
      t1 = sP[(kt + 1) * 8 + 4 + fgrp];
      mma(0);
    }
    if (pend) mma(1);
  }
  stamp();
  __syncthreads();
  stamp();
  conv_epilogue<T, BM, BN, WM, WN, EPI>(a, acc, smem, m0, n0, a.lds_bytes);
  stamp();
}

// ===================================================================== host side
struct PatchCfg { int bm, bn, wm, wn, s; };
static const PatchCfg kPatch[] = {
    {256, 32, 8, 1, 3},  {256, 32, 4, 1, 3},  {128, 32, 4, 1, 3},  {256, 64, 4, 2, 3},  {128, 64, 4, 2, 3},
    {128, 64, 2, 2, 3},  {64, 64, 2, 2, 4},   {256, 192, 4, 2, 3}, {128, 192, 2, 4, 3}, {64, 192, 2, 4, 3},
    {128, 256, 2, 4, 3}, {256, 256, 4, 2, 3}, {128, 128, 2, 4, 3}, {256, 128, 4, 2, 3}, {64, 128, 2, 4, 3},
    {64, 128, 2, 2, 4},
    // deeper weight rings: the patch is resident, so ring stages are only BN x 128 B and more of them keep more bytes in flight
    {256, 64, 4, 2, 6},  {128, 64, 4, 2, 6},  {256, 128, 4, 2, 4}, {256, 32, 8, 1, 6},
    // <= 80 KiB of LDS: two workgroups per CU, so one's patch load / epilogue runs under the other's K loop
    {128, 96, 4, 1, 3},  {128, 96, 2, 2, 3},  {64, 192, 1, 4, 3},
    // 192-row tiles: conv2d_4a's 331,776 pixels are 1296 tiles of 256 = 5.06 rounds over 256 CUs (a sixth round for 16
    // workgroups), but 1728 tiles of 192 = 6.75 rounds of 3/4 the size
    {192, 192, 4, 2, 3},
};
constexpr int kNumPatch = (int)(sizeof(kPatch) / sizeof(kPatch[0]));

int patch_num_cfgs() { return kNumPatch; }

// most virtual input rows any BM-pixel tile of a (Ho x Wo, KH) layer touches
static int patch_rows_max(int Wo, int Ho, int KH, int BM) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int>, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_tuple(Wo, Ho, KH, BM);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int best = 0;
  const long long HoWo = (long long)Ho * Wo;
  for (long long t = 0; t <= HoWo; ++t) {  // tile starts repeat with period <= Ho*Wo tiles
    const long long m0 = t * BM, m1 = m0 + BM - 1;
    const long long g0 = m0 / Wo, g1 = m1 / Wo;
    const long long v0 = g0 + (g0 / Ho) * (KH - 1), v1 = g1 + (g1 / Ho) * (KH - 1) + KH - 1;
    if (v1 - v0 + 1 > best) best = (int)(v1 - v0 + 1);
  }
  cache[key] = best;
  return best;
}

struct PatchGeom { int pp, Wp, Hv, patch_bytes, lds; };

static bool patch_geom(const ConvArgs& a, const PatchCfg& c, PatchGeom& g) {
  const int es = dtype_size(a.dtype), ch = dtype_chan_align(a.dtype);   // slots per pixel: 16-byte chunks (planar: per plane)
  const int planes = a.dtype == F16P ? 2 : 1;
  if (a.sh != 1 || a.sw != 1 || a.Cin % ch || a.Ho != a.H + 2 * a.ph - a.KH + 1 || a.Wo != a.W + 2 * a.pw - a.KW + 1)
    return false;
  g.pp = a.Cin / ch;
  while (g.pp % 4 != 2) ++g.pp;
  g.Wp = a.W + 2 * a.pw;
  g.Hv = a.H + 2 * a.ph;
  const long long slots = (long long)patch_rows_max(a.Wo, a.Ho, a.KH, c.bm) * g.Wp * g.pp;
  if (slots * 16 * planes > 160 * 1024) return false;
  g.patch_bytes = (int)((slots * 16 + 1023) / 1024 * 1024) * planes;
  const int nkt = a.Kpad / (128 / es);
  int lds = c.s * c.bn * 128 + (((nkt + 1) * 8 * 4 + 15) & ~15) + 16 + g.patch_bytes;
  const int epi = patch_epi_bytes(c.bm, c.bn, c.wm);
  if (lds < epi) lds = epi;
  g.lds = lds;
  return lds <= 160 * 1024;
}

bool patch_cfg_ok(const ConvArgs& a, int pcfg) {
  if (pcfg < 0 || pcfg >= kNumPatch) return false;
  const PatchCfg& c = kPatch[pcfg];
  if (a.KH * a.KW == 1) return false;                  // no tap reuse: the ring kernel's job
  if (c.bn > 32 && a.Cout <= c.bn / 2) return false;   // more than half the tile would be padding
  if (((a.Cout + c.bn - 1) / c.bn) * c.bn > a.cout_pad) return false;
  if ((c.bn % 64) && a.Cout % c.bn) return false;
  PatchGeom g;
  return patch_geom(a, c, g);
}

template <typename T, int BM, int BN, int WM, int WN, int S>
static hipError_t launch_one(const KArgs& k, int lds, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_patch_kernel<T, BM, BN, WM, WN, S>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  KArgs kk = k;
  const int tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.Cout + BN - 1) / BN;
  kk.nblk = tiles_m * kk.tiles_n;
  hipLaunchKernelGGL((conv_patch_kernel<T, BM, BN, WM, WN, S>), dim3(kk.nblk), dim3(WM * WN * 64), lds, s, kk);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_patch_typed(int pcfg, const KArgs& k, int lds, hipStream_t s) {
  switch (pcfg) {
    case 0: return launch_one<T, 256, 32, 8, 1, 3>(k, lds, s);
    case 1: return launch_one<T, 256, 32, 4, 1, 3>(k, lds, s);
    case 2: return launch_one<T, 128, 32, 4, 1, 3>(k, lds, s);
    case 3: return launch_one<T, 256, 64, 4, 2, 3>(k, lds, s);
    case 4: return launch_one<T, 128, 64, 4, 2, 3>(k, lds, s);
    case 5: return launch_one<T, 128, 64, 2, 2, 3>(k, lds, s);
    case 6: return launch_one<T, 64, 64, 2, 2, 4>(k, lds, s);
    case 7: return launch_one<T, 256, 192, 4, 2, 3>(k, lds, s);
    case 8: return launch_one<T, 128, 192, 2, 4, 3>(k, lds, s);
    case 9: return launch_one<T, 64, 192, 2, 4, 3>(k, lds, s);
    case 10: return launch_one<T, 128, 256, 2, 4, 3>(k, lds, s);
    case 11: return launch_one<T, 256, 256, 4, 2, 3>(k, lds, s);
    case 12: return launch_one<T, 128, 128, 2, 4, 3>(k, lds, s);
    case 13: return launch_one<T, 256, 128, 4, 2, 3>(k, lds, s);
    case 14: return launch_one<T, 64, 128, 2, 4, 3>(k, lds, s);
    case 15: return launch_one<T, 64, 128, 2, 2, 4>(k, lds, s);
    case 16: return launch_one<T, 256, 64, 4, 2, 6>(k, lds, s);
    case 17: return launch_one<T, 128, 64, 4, 2, 6>(k, lds, s);
    case 18: return launch_one<T, 256, 128, 4, 2, 4>(k, lds, s);
    case 19: return launch_one<T, 256, 32, 8, 1, 6>(k, lds, s);
    case 20: return launch_one<T, 128, 96, 4, 1, 3>(k, lds, s);
    case 21: return launch_one<T, 128, 96, 2, 2, 3>(k, lds, s);
    case 22: return launch_one<T, 64, 192, 1, 4, 3>(k, lds, s);
    case 23: return launch_one<T, 192, 192, 4, 2, 3>(k, lds, s);
  }
  return hipErrorInvalidValue;
}

// Instrumented launch (tools/stamp_patch.py): VNF_PATCH_STAMP=<file> dumps cycle stamps of workgroup 700 of every bf16
// {256,192,4,2,3} launch with more than 700 workgroups: start, prologue issued, before / after the barrier of every K
// tile, K loop done, epilogue barrier, end.
static hipError_t launch_patch_stamped(const KArgs& k, int lds, hipStream_t s) {
  static long long* dbuf = nullptr;
  const int n = 8 * 40;
  if (!dbuf && hipMalloc((void**)&dbuf, n * 8) != hipSuccess) return hipErrorOutOfMemory;
  (void)hipMemsetAsync(dbuf, 0, n * 8, s);
  KArgs kk = k;
  kk.dbg = dbuf;
  kk.tiles_n = (k.Cout + 191) / 192;
  kk.nblk = ((k.M + 255) / 256) * kk.tiles_n;
  (void)hipFuncSetAttribute((const void*)conv_patch_kernel<__bf16, 256, 192, 4, 2, 3, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_patch_kernel<__bf16, 256, 192, 4, 2, 3, true>), dim3(kk.nblk), dim3(512), lds, s, kk);
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  static long long host[8 * 40];
  (void)hipMemcpy(host, dbuf, n * 8, hipMemcpyDeviceToHost);
  if (FILE* f = fopen(getenv("VNF_PATCH_STAMP"), "a")) {
    fprintf(f, "launch M=%d N=%d nkt=%d nblk=%d\n", k.M, k.Cout, k.nkt, kk.nblk);
    for (int w = 0; w < 8; ++w) {
      fprintf(f, "%d", w);
      for (int i = 0; i < 40; ++i) fprintf(f, " %lld", host[w * 40 + i] ? host[w * 40 + i] - host[0] : -1LL);
      fprintf(f, "\n");
    }
    fclose(f);
  }
  return hipSuccess;
}

hipError_t launch_patch(const ConvArgs& a, const KArgs& k, int pcfg, hipStream_t s) {
  if (!patch_cfg_ok(a, pcfg) || !k.zero) return hipErrorInvalidValue;
  PatchGeom g;
  patch_geom(a, kPatch[pcfg], g);
  KArgs kk = k;
  kk.KH = a.KH; kk.KW = a.KW; kk.Cin = a.Cin;
  kk.pp = g.pp; kk.Wp = g.Wp; kk.Hv = g.Hv; kk.patch_bytes = g.patch_bytes; kk.lds_bytes = g.lds;
  if (getenv("VNF_PATCH_STAMP") && pcfg == 7 && a.dtype == BF16 && ((k.M + 255) / 256) * ((k.Cout + 191) / 192) > 700)
    return launch_patch_stamped(kk, g.lds, s);
  switch (a.dtype) {
    case BF16: return launch_patch_typed<__bf16>(pcfg, kk, g.lds, s);
    case F16: return launch_patch_typed<_Float16>(pcfg, kk, g.lds, s);
    case F32: return launch_patch_typed<float>(pcfg, kk, g.lds, s);
    case F16X2: return launch_patch_typed<sf16>(pcfg, kk, g.lds, s);
    case F16P: return launch_patch_typed<pf16>(pcfg, kk, g.lds, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace vnf
