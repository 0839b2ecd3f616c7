// Device-side building blocks shared by the implicit-GEMM convolution kernels (conv_igemm.hip: LDS-DMA ring
// kernel + register-staged fallback; conv_patch.hip: LDS-resident input patch kernel).
#pragma once
#include "kernels.h"
#include "split_f16.h"

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
  const char* zero;  // >= 16 zero bytes
  int ldx, H, W, Ho, Wo, sh, sw, ph, pw;
  int K, Kpad, nkt;
  int wrs, wts;  // weight image strides in bytes: between output channels, between K tiles
  int ncls, cout_pad;
  int M, Cout, tiles_n, nblk;
  int ldres, act, out_f32;
  int nseg;
  int seg_c0[4], seg_c1[4], seg_ld[4];
  char* seg_ptr[4];
  // patch kernel (conv_patch.hip) geometry: taps, channels per tap, LDS pixel pitch in 16-byte
  // slots, padded row width, virtual (vertically padded) image height, patch bytes
  int KH, KW, Cin, pp, Wp, Hv, patch_bytes;
  int lds_bytes;  // patch kernel: dynamic LDS of the launch (what the epilogue may stage into)
  long long* dbg;  // in-kernel stamp buffer of the instrumented build (tools), else null
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

template <>
__device__ __forceinline__ void mma_chunk<sf16>(f32x4_t& acc, const uint4& wf, const uint4& xf) {
  // each operand chunk is 4 k values as (hi, lo) halves: slots [h0 l0 h1 l1 h2 l2 h3 l3].  The first MFMA pairs
  // equal slots (sum hi*hi' + lo*lo'), the second pairs the weight chunk with the activation chunk's halves
  // swapped inside every dword (sum hi*lo' + lo*hi'): together the full product of the two split values.
  const uint4 xr = {(xf.x >> 16) | (xf.x << 16), (xf.y >> 16) | (xf.y << 16), (xf.z >> 16) | (xf.z << 16),
                    (xf.w >> 16) | (xf.w << 16)};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf), __builtin_bit_cast(f16x8_t, xf),
                                               acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf), __builtin_bit_cast(f16x8_t, xr),
                                               acc, 0, 0, 0);
}

// ---- planar split-f16 (pf16, split_f16.h): one K tile in LDS = [hi chunk of unit 0..3 | lo chunk of unit 0..3], so the
// fragment reads of "half" 0 / 1 of a K tile (chunk ks*4 + fgrp) return the hi / lo plane of the lane group's 8 k values
template <typename T> struct is_planar { static constexpr bool value = false; };
template <> struct is_planar<pf16> { static constexpr bool value = true; };

// channel (k) index of logical 16-byte chunk `lchunk` (0..7) inside a K tile, and the chunk's extra byte offset
// inside its 8-channel unit (planar: chunks 4..7 are the lo planes of units 0..3)
template <typename T> __device__ __forceinline__ int chunk_chan(int lchunk) {
  if constexpr (is_planar<T>::value) return (lchunk & 3) * 8;
  else return lchunk * (16 / (int)sizeof(T));
}
template <typename T> __device__ __forceinline__ int chunk_byte(int lchunk) {
  if constexpr (is_planar<T>::value) return (lchunk >> 2) * 16;
  else return 0;
}

__device__ __forceinline__ f32x4_t mfma_f16(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// planar split product of one K tile for one (weight tile, pixel tile): hi.hi' first, then the two cross terms
__device__ __forceinline__ void mma_hh(f32x4_t& acc, const uint4& wh, const uint4& xh) { acc = mfma_f16(wh, xh, acc); }
__device__ __forceinline__ void mma_cross(f32x4_t& acc, const uint4& wh, const uint4& wl, const uint4& xh, const uint4& xl) {
  acc = mfma_f16(wh, xl, acc);
  acc = mfma_f16(wl, xh, acc);
}
template <>
__device__ __forceinline__ void mma_chunk<pf16>(f32x4_t&, const uint4&, const uint4&) {}  // planar tiles go through mma_hh / mma_cross

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

template <>
__device__ __forceinline__ void load8<sf16>(const char* p, float (&v)[8]) {
  f16x8_t a = *reinterpret_cast<const f16x8_t*>(p), b = *reinterpret_cast<const f16x8_t*>(p + 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (float)a[2 * i] + (float)a[2 * i + 1];
    v[4 + i] = (float)b[2 * i] + (float)b[2 * i + 1];
  }
}

template <>
__device__ __forceinline__ void load8<pf16>(const char* p, float (&v)[8]) {
  f16x8_t h = *reinterpret_cast<const f16x8_t*>(p), l = *reinterpret_cast<const f16x8_t*>(p + 16);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i];
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

template <>
__device__ __forceinline__ void store8<sf16>(char* p, const float (&v)[8]) {
  f16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const sf16 s0(v[i]), s1(v[4 + i]);
    a[2 * i] = s0.hi; a[2 * i + 1] = s0.lo;
    b[2 * i] = s1.hi; b[2 * i + 1] = s1.lo;
  }
  *reinterpret_cast<f16x8_t*>(p) = a;
  *reinterpret_cast<f16x8_t*>(p + 16) = b;
}

template <>
__device__ __forceinline__ void store8<pf16>(char* p, const float (&v)[8]) {
  f16x8_t h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const sf16 s(v[i]);
    h[i] = s.hi; l[i] = s.lo;
  }
  *reinterpret_cast<f16x8_t*>(p) = h;
  *reinterpret_cast<f16x8_t*>(p + 16) = l;
}

// XCD-aware, bijective block -> tile map: blocks that share an XCD (same blockIdx % 8) walk
// consecutive tiles, so the BN-column siblings of one pixel tile hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- epilogue shared by the conv kernels: fp32 accumulators -> LDS -> whole NHWC rows ------------
// Fast form (the stem / reduction convolutions of the 2-byte plans: bias + ReLU / PReLU, no residual, no border-class bias,
// output in T): bias and activation are applied in the accumulator layout, the tile is staged ALREADY ROUNDED to T --
// half the LDS bytes, so the whole tile fits one pass and one barrier -- and every global load (the bias) is issued
// before the first store: vmcnt counts loads and stores together in issue order, so a load behind a store waits for
// the store's acknowledgement (the staged fp32 form paid that once per pass: conv2d_4a's epilogue was 17 k of its
// workgroups' 59 k cycles, in-kernel stamps).  Same fp32 sum, bias add, max and rounding as the general form, so the
// results are bit-identical to it.
template <typename T, int BM, int BN>
__device__ __forceinline__ bool conv_epilogue_is_fast(const KArgs& a, int avail) {
  if constexpr (sizeof(T) != 2)
    return false;
  else
    return !a.res && a.ncls == 1 && !a.out_f32 && BM * (BN * 2 + 16) <= avail;
}

template <typename T, int BM, int BN, int WM, int WN, int LDS_BYTES>
__device__ __forceinline__ void conv_epilogue(const KArgs& a, f32x4_t (&acc)[BM / WM / 16][BN / WN / 16], char* smem,
                                              int m0, int n0, int avail = LDS_BYTES) {
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tiles are whole 16x16 MFMA tiles");
  constexpr int ES = (int)sizeof(T);
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int NT = WM * WN * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, frow = lane & 15, fgrp = lane >> 4;
  const int HoWo = a.Ho * a.Wo;
  if constexpr (sizeof(T) == 2) {
    if (conv_epilogue_is_fast<T, BM, BN>(a, avail)) {
      constexpr int PITCH = BN * 2 + 16;  // +16 B: the 16 rows of one ds_write_b64 land on different banks
      typedef T tx4 __attribute__((ext_vector_type(4)));
      f32x4_t bias[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = n0 + wn * WTN + j * 16 + fgrp * 4;
        bias[j] = c < a.Cout ? *reinterpret_cast<const f32x4_t*>(a.bias + c) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      const bool relu = a.act == ACT_RELU, prelu = a.act == ACT_PRELU;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4_t slope = {0.f, 0.f, 0.f, 0.f};
        if (prelu) {
          const int c = n0 + wn * WTN + j * 16 + fgrp * 4;
          if (c < a.Cout) slope = *reinterpret_cast<const f32x4_t*>(a.slope + c);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          tx4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][e] + bias[j][e];
            if (relu) v = fmaxf(v, 0.f);
            if (prelu) v = v > 0.f ? v : v * slope[e];
            o[e] = (T)v;
          }
          *reinterpret_cast<tx4*>(smem + (wm * WTM + i * 16 + frow) * PITCH + (wn * WTN + j * 16 + fgrp * 4) * 2) = o;
        }
      }
      __syncthreads();
      constexpr int CPR8 = BN / 8;
      for (int idx = tid; idx < BM * CPR8; idx += NT) {
        const int r = idx / CPR8, cc = idx - r * CPR8;
        const int m = m0 + r, c = n0 + cc * 8;
        if (m < a.M && c < a.Cout) {
          int sg = 0;
#pragma unroll
          for (int s = 1; s < 4; ++s)
            if (s < a.nseg && c >= a.seg_c0[s]) sg = s;
          *reinterpret_cast<uint4*>(a.seg_ptr[sg] + ((size_t)m * a.seg_ld[sg] + (c - a.seg_c0[sg])) * 2) =
              *reinterpret_cast<const uint4*>(smem + r * PITCH + cc * 16);
        }
      }
      return;
    }
  }
  float* sC = reinterpret_cast<float*>(smem);
  constexpr int CST = BN + 4;  // floats per staged row (+16 B pad: conflict-free float4 writes)
  constexpr int CPR = BN / 8;  // 8-channel chunks per row
  // as many wave rows per pass as the staging LDS holds
  constexpr int EPASS = (BM * CST * 4 <= LDS_BYTES) ? 1 : WM;
  constexpr int RPP = BM / EPASS;
  static_assert(RPP * CST * 4 <= LDS_BYTES, "epilogue staging does not fit");
  // Thread map of the store phase.  When the chunk count per row divides the thread count every
  // thread owns ONE 8-channel column chunk (bias loaded once, residual loads hoisted above the
  // barrier); tiles whose width is not a power of two (96, 192 channels) use the general map.
  constexpr bool FIXEDCOL = (NT % CPR == 0) && ((RPP * CPR) % NT == 0);
  constexpr int NIT = (RPP * CPR + NT - 1) / NT, RSTEP = NT / CPR;
  if constexpr (FIXEDCOL) {
    const int cc = tid % CPR, r0 = tid / CPR;
    const int c = n0 + cc * 8;
    const bool cok = c < a.Cout;
    const int cl = cok ? c : 0;  // in-range column for the loads of masked threads
    // destination tensor of this thread's 8-channel chunk (segment boundaries are multiples of 8, so a
    // tile may span several destinations: concat-free routing happens per chunk, not per tile)
    int sg = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
      if (s < a.nseg && c >= a.seg_c0[s]) sg = s;
    char* const dptr = a.seg_ptr[sg];
    const int dld = a.seg_ld[sg], dc0 = a.seg_c0[sg];
    f32x4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    float slope[8];
    if (a.ncls == 1) {
      b0 = *reinterpret_cast<const f32x4_t*>(a.bias + cl);
      b1 = *reinterpret_cast<const f32x4_t*>(a.bias + cl + 4);
    }
    if (a.act == ACT_PRELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) slope[e] = a.slope[cl + e];
    }
    for (int pass = 0; pass < EPASS; ++pass) {
      float rv[NIT][8];
      if (a.res) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int m = min(m0 + pass * RPP + r0 + it * RSTEP, a.M - 1);
          load8<T>(a.res + ((size_t)m * a.ldres + cl) * ES, rv[it]);
        }
      }
      if ((wm * WTM) / RPP == pass) {
        const int rbase = wm * WTM - pass * RPP;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4_t*>(sC + (rbase + i * 16 + frow) * CST + wn * WTN + j * 16 + fgrp * 4) = acc[i][j];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int r = r0 + it * RSTEP;
        const int m = m0 + pass * RPP + r;
        float v[8];
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(sC + r * CST + cc * 8);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(sC + r * CST + cc * 8 + 4);
        if (a.ncls == 9) {
          const int mm = min(m, a.M - 1);
          const int rr = mm % HoWo, ho = rr / a.Wo, wo = rr - ho * a.Wo;
          const int cls = (ho == 0 ? 0 : (ho == a.Ho - 1 ? 2 : 1)) * 3 + (wo == 0 ? 0 : (wo == a.Wo - 1 ? 2 : 1));
          const float* bp = a.bias + (size_t)cls * a.cout_pad + cl;
          b0 = *reinterpret_cast<const f32x4_t*>(bp);
          b1 = *reinterpret_cast<const f32x4_t*>(bp + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + b0[e]; v[4 + e] = v1[e] + b1[e]; }
        if (a.res) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[it][e];
        }
        if (a.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (a.act == ACT_PRELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope[e];
        }
        if (m < a.M && cok) {
          if (a.out_f32)
            store8<float>(dptr + ((size_t)m * dld + (c - dc0)) * 4, v);
          else
            store8<T>(dptr + ((size_t)m * dld + (c - dc0)) * ES, v);
        }
      }
      __syncthreads();
    }
  } else {
    for (int pass = 0; pass < EPASS; ++pass) {
      if ((wm * WTM) / RPP == pass) {
        const int rbase = wm * WTM - pass * RPP;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4_t*>(sC + (rbase + i * 16 + frow) * CST + wn * WTN + j * 16 + fgrp * 4) = acc[i][j];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        const int r = idx / CPR, cc = idx - r * CPR;
        const int m = m0 + pass * RPP + r, c = n0 + cc * 8;
        if (idx < RPP * CPR && m < a.M && c < a.Cout) {
          int sg = 0;
#pragma unroll
          for (int s = 1; s < 4; ++s)
            if (s < a.nseg && c >= a.seg_c0[s]) sg = s;
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
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope[c + e];
          }
          char* const dptr = a.seg_ptr[sg];
          if (a.out_f32)
            store8<float>(dptr + ((size_t)m * a.seg_ld[sg] + (c - a.seg_c0[sg])) * 4, v);
          else
            store8<T>(dptr + ((size_t)m * a.seg_ld[sg] + (c - a.seg_c0[sg])) * ES, v);
        }
      }
      __syncthreads();
    }
  }
}

// One LDS-DMA piece: 64 lanes x 16 B -> lds_base + lane*16.  M0 carries the LDS base; it is
// compiler-reserved, so it is saved and restored inside the statement (cdna guide 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}

// Retire this wave's older DMA pieces (all but the N youngest), drain its own LDS reads of the
// stage about to be recycled, then meet the other waves.  One statement, so the compiler cannot
// move LDS accesses between the wait and the barrier.
template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- conv_patch.hip (LDS-resident input patch kernel; tile configurations follow the ring kernel's ids)
// epilogue staging bytes of a patch configuration: the whole tile when it fits 64 KiB, else one wave row per pass
constexpr int patch_epi_bytes(int bm, int bn, int wm) {
  return bm * (bn + 4) * 4 <= 64 * 1024 ? bm * (bn + 4) * 4 : (bm / wm) * (bn + 4) * 4;
}
int patch_num_cfgs();
bool patch_cfg_ok(const ConvArgs& a, int pcfg);
hipError_t launch_patch(const ConvArgs& a, const KArgs& k, int pcfg, hipStream_t s);

// ---- conv_ws.hip (wave-specialised big-tile kernel; configuration ids follow the patch kernel's)
int ws_num_cfgs();
bool ws_cfg_ok(const ConvArgs& a, int wcfg);
hipError_t launch_ws(const ConvArgs& a, const KArgs& k, int wcfg, hipStream_t s);

}  // namespace vnf
