// Implicit-GEMM convolution for gfx950 (MI355X), NHWC activations, MFMA 16x16 tiles.
//
// GEMM view:  C[m][co] = sum_k A[m][k] * Wt[co][k]
//   m  = output pixel (n, ho, wo)          -> "pixel" axis, BM per workgroup
//   co = output channel                     -> "channel" axis, BN per workgroup
//   k  = (kh, kw, c) flattened, c fastest  -> walked in 128-byte K tiles
// A is never materialised: every 16-byte k-chunk (8 x 16-bit or 4 x f32 channels of one filter tap)
// is gathered straight from the NHWC input with a per-chunk offset table (ktab), zero-filled at
// padded borders.  Both operands are staged global -> registers -> LDS (double-buffered, one
// barrier per K tile; the next tile's global loads are in flight under the MFMAs of the current
// one), in 128-byte rows whose 16-byte chunks are XOR-swizzled with (row & 7) so that the
// ds_write_b128 of the loader and the ds_read_b128 of the fragment reads are both bank-conflict
// free on CDNA4's 64-bank LDS (checked exhaustively).
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
#include "kernels.h"

namespace vnf {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct KArgs {  // device-side copy of ConvArgs (POD)
  const char* x;
  const char* w;
  const float* bias;
  const int4* ktab;
  const char* res;
  const float* slope;
  int ldx, H, W, Ho, Wo, sh, sw, ph, pw;
  int K, Kpad, nkt;
  int ncls, cout_pad;
  int M, Cout, tiles_n, nblk;
  int ldres, act, out_f32;
  int nseg;
  int seg_c0[4], seg_c1[4], seg_ld[4];
  char* seg_ptr[4];
};

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x4_t& acc, const uint4& wf, const uint4& xf);

template <>
__device__ __forceinline__ void mma_chunk<__bf16>(f32x4_t& acc, const uint4& wf, const uint4& xf) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf),
                                                acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<_Float16>(f32x4_t& acc, const uint4& wf, const uint4& xf) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf), __builtin_bit_cast(f16x8_t, xf),
                                               acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x4_t& acc, const uint4& wf, const uint4& xf) {
  // lane group g holds k = 4g..4g+3 of this 16-deep block; MFMA j consumes element j of every
  // group, i.e. the k set {4g+j}.  Any consistent k permutation is a valid dot product.
  f32x4_t w4 = __builtin_bit_cast(f32x4_t, wf), x4 = __builtin_bit_cast(f32x4_t, xf);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[0], x4[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[1], x4[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[2], x4[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[3], x4[3], acc, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void load8(const char* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<__bf16>(const char* p, float (&v)[8]) {
  bf16x8_t r = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
template <>
__device__ __forceinline__ void load8<_Float16>(const char* p, float (&v)[8]) {
  f16x8_t r = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
template <>
__device__ __forceinline__ void load8<float>(const char* p, float (&v)[8]) {
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

template <typename T>
__device__ __forceinline__ void store8(char* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<__bf16>(char* p, const float (&v)[8]) {
  bf16x8_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
  *reinterpret_cast<bf16x8_t*>(p) = r;
}
template <>
__device__ __forceinline__ void store8<_Float16>(char* p, const float (&v)[8]) {
  f16x8_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (_Float16)v[i];
  *reinterpret_cast<f16x8_t*>(p) = r;
}
template <>
__device__ __forceinline__ void store8<float>(char* p, const float (&v)[8]) {
  f32x4_t a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4_t*>(p) = a;
  *reinterpret_cast<f32x4_t*>(p + 16) = b;
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int CH = 16 / ES;    // elements per 16-byte chunk
  constexpr int BKE = 128 / ES;  // elements per K tile
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(WM * WN == 4, "4 waves per workgroup");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware, bijective block -> tile map: blocks that share an XCD (same blockIdx % 8) walk
  // consecutive tiles, so the BN-column siblings of one pixel tile hit the same L2.
  int bid = blockIdx.x;
  {
    const int q = a.nblk >> 3, r = a.nblk & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = tid >> 3, lcol = tid & 7;
  int abase[AP], ahi[AP], awi[AP];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + lrow + 32 * p;
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
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (kt * BKE + ks * (BKE / 2) < a.K) {
        uint4 xf[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WTM + i * 16 + frow;
          xf[i] = *reinterpret_cast<const uint4*>(sA + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * WTN + j * 16 + frow;
          wf[j] = *reinterpret_cast<const uint4*>(sB + row * 128 + (((ks * 4 + fgrp) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[j], xf[i]);
      }
    }
    if (kt + 1 < a.nkt) lstore(st ^ 1);
    __syncthreads();
  }

  // ---- epilogue: fp32 accumulators -> LDS -> whole NHWC rows ------------------------------
  int sg = 0;
#pragma unroll
  for (int s = 1; s < 4; ++s)
    if (s < a.nseg && n0 >= a.seg_c0[s]) sg = s;
  char* const dptr = a.seg_ptr[sg];
  const int dld = a.seg_ld[sg], dc0 = a.seg_c0[sg];

  float* sC = reinterpret_cast<float*>(smem);
  constexpr int CST = BN + 4;  // floats per staged row (+16 B pad: conflict-free float4 writes)
  constexpr int CPR = BN / 8;  // 8-channel chunks per row
  for (int pass = 0; pass < WM; ++pass) {
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          *reinterpret_cast<f32x4_t*>(sC + (i * 16 + frow) * CST + wn * WTN + j * 16 + fgrp * 4) = acc[i][j];
    }
    __syncthreads();
    for (int idx = tid; idx < WTM * CPR; idx += 256) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int m = m0 + pass * WTM + r, c = n0 + cc * 8;
      if (m < a.M && c < a.Cout) {
        float v[8];
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(sC + r * CST + cc * 8);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(sC + r * CST + cc * 8 + 4);
        int cls = 0;
        if (a.ncls == 9) {
          const int rr = m % HoWo, ho = rr / a.Wo, wo = rr - ho * a.Wo;
          cls = (ho == 0 ? 0 : (ho == a.Ho - 1 ? 2 : 1)) * 3 + (wo == 0 ? 0 : (wo == a.Wo - 1 ? 2 : 1));
        }
        const float* bp = a.bias + (size_t)cls * a.cout_pad + c;
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bp), b1 = *reinterpret_cast<const f32x4_t*>(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + b0[e]; v[4 + e] = v1[e] + b1[e]; }
        if (a.res) {
          float rv[8];
          load8<T>(a.res + ((size_t)m * a.ldres + c) * ES, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (a.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (a.act == ACT_PRELU) {
          const float* sp = a.slope + c;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sp[e];
        }
        if (a.out_f32)
          store8<float>(dptr + ((size_t)m * dld + (c - dc0)) * 4, v);
        else
          store8<T>(dptr + ((size_t)m * dld + (c - dc0)) * ES, v);
      }
    }
    __syncthreads();
  }
}

template <typename T, int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const KArgs& k, hipStream_t s) {
  constexpr int stage = 2 * (BM + BN) * 128;
  constexpr int epi = (BM / WM) * (BN + 4) * 4;
  constexpr int lds = stage > epi ? stage : epi;
  KArgs kk = k;
  const int tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.Cout + BN - 1) / BN;
  kk.nblk = tiles_m * kk.tiles_n;
  hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN>), dim3(kk.nblk), dim3(256), lds, s, kk);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_typed(const ConvArgs& a, const KArgs& k, hipStream_t s) {
  // BN must divide every segment boundary so a tile maps to exactly one destination
  int bn = a.bn;
  if (bn == 0) {
    bn = 128;
    auto fits = [&](int b) {
      for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].c0 % b) return false;
      return true;
    };
    while (bn > 32 && !(fits(bn) && (a.Cout % bn == 0 || a.Cout > 2 * bn))) bn >>= 1;
    if (!fits(bn)) return hipErrorInvalidValue;
    // keep at least ~2 workgroups per CU in flight when the problem allows it
    auto blocks = [&](int bm_, int bn_) { return ((a.M + bm_ - 1) / bm_) * ((a.Cout + bn_ - 1) / bn_); };
    while (bn > 64 && blocks(128, bn) < 512) bn >>= 1;
  }
  int bm = a.bm;
  if (bm == 0) {
    bm = 128;
    if (bn <= 64 && ((a.M + 127) / 128) * ((a.Cout + bn - 1) / bn) < 512) bm = 64;
  }
  if (bm == 128 && bn == 128) return launch_cfg<T, 128, 128, 2, 2>(k, s);
  if (bm == 128 && bn == 64) return launch_cfg<T, 128, 64, 2, 2>(k, s);
  if (bm == 128 && bn == 32) return launch_cfg<T, 128, 32, 4, 1>(k, s);
  if (bm == 64 && bn == 64) return launch_cfg<T, 64, 64, 2, 2>(k, s);
  if (bm == 64 && bn == 32) return launch_cfg<T, 64, 32, 2, 2>(k, s);
  return hipErrorInvalidValue;
}

hipError_t launch_conv(const ConvArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int es = dtype_size(a.dtype);
  const int bke = 128 / es;
  if (a.Cout % 8 || a.Kpad % bke || a.nseg < 1 || a.nseg > 4) return hipErrorInvalidValue;
  KArgs k;
  k.x = (const char*)a.x; k.w = (const char*)a.w; k.bias = a.bias; k.ktab = a.ktab;
  k.res = (const char*)a.res; k.slope = a.slope;
  k.ldx = a.ldx; k.H = a.H; k.W = a.W; k.Ho = a.Ho; k.Wo = a.Wo;
  k.sh = a.sh; k.sw = a.sw; k.ph = a.ph; k.pw = a.pw;
  k.K = a.K; k.Kpad = a.Kpad; k.nkt = a.Kpad / bke;
  k.ncls = a.ncls; k.cout_pad = a.cout_pad;
  k.M = a.M; k.Cout = a.Cout; k.tiles_n = 0; k.nblk = 0;
  k.ldres = a.ldres; k.act = a.act; k.out_f32 = a.out_f32;
  k.nseg = a.nseg;
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
  }
  return hipErrorInvalidValue;
}

}  // namespace vnf
