// HBM-bound helper kernels around the convolution core: layout packing, pooling, L2 norm.
// All are coalesced streaming kernels (16-byte accesses along the NHWC channel axis).
#include <cstdlib>
#include <algorithm>
#include "kernels.h"
#include "split_f16.h"

namespace vnf {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

// planar split-f16 (split_f16.h pf16): 8 consecutive channels = [8 hi halves][8 lo halves], 32 bytes
__device__ __forceinline__ void store_unit_pf16(void* dst, const float (&v)[8]) {
  f16x8_t h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const sf16 s(v[i]);
    h[i] = s.hi; l[i] = s.lo;
  }
  reinterpret_cast<f16x8_t*>(dst)[0] = h;
  reinterpret_cast<f16x8_t*>(dst)[1] = l;
}
__device__ __forceinline__ void load_unit_pf16(const void* src, float (&v)[8]) {
  const f16x8_t h = reinterpret_cast<const f16x8_t*>(src)[0], l = reinterpret_cast<const f16x8_t*>(src)[1];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i];
}

// ---------------------------------------------------------------- NCHW (n,3,S,S) -> NHWC8
template <typename TI, typename TO>
__global__ void pack_input_kernel(const TI* __restrict__ x, TO* __restrict__ y, int n, int hw) {
  const size_t total = (size_t)n * hw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / hw, p = i - img * hw;
    const TI* src = x + img * 3 * (size_t)hw + p;
    if constexpr (sizeof(TO) == sizeof(pf16) && __is_same(TO, pf16)) {
      const float v[8] = {to_f(src[0]), to_f(src[hw]), to_f(src[2 * (size_t)hw]), 0.f, 0.f, 0.f, 0.f, 0.f};
      store_unit_pf16(y + i * 8, v);
    } else {
      TO o[8];
      o[0] = (TO)to_f(src[0]);
      o[1] = (TO)to_f(src[hw]);
      o[2] = (TO)to_f(src[2 * (size_t)hw]);
#pragma unroll
      for (int c = 3; c < 8; ++c) o[c] = (TO)0.f;
      TO* dst = y + i * 8;
      if constexpr (sizeof(TO) == 2) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
      } else {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
        *reinterpret_cast<uint4*>(dst + 4) = *reinterpret_cast<const uint4*>(o + 4);
      }
    }
  }
}

template <typename TI>
static hipError_t pack_dispatch_out(const void* x, void* out, int dtype, int n, int hw, hipStream_t s) {
  const size_t total = (size_t)n * hw;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (blocks == 0) return hipSuccess;
  switch (dtype) {
    case BF16: hipLaunchKernelGGL((pack_input_kernel<TI, __bf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (__bf16*)out, n, hw); break;
    case F16: hipLaunchKernelGGL((pack_input_kernel<TI, _Float16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (_Float16*)out, n, hw); break;
    case F32: hipLaunchKernelGGL((pack_input_kernel<TI, float>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (float*)out, n, hw); break;
    case F16X2: hipLaunchKernelGGL((pack_input_kernel<TI, sf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (sf16*)out, n, hw); break;
    case F16P: hipLaunchKernelGGL((pack_input_kernel<TI, pf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (pf16*)out, n, hw); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_pack_input(const void* x, int x_dtype, void* out, int dtype, int n, int hw, hipStream_t s) {
  switch (x_dtype) {
    case F32: return pack_dispatch_out<float>(x, out, dtype, n, hw, s);
    case BF16: return pack_dispatch_out<__bf16>(x, out, dtype, n, hw, s);
    case F16: return pack_dispatch_out<_Float16>(x, out, dtype, n, hw, s);
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------- IRv1 stem: NCHW input -> conv2d_1a, direct
// inception_resnet_v1.py:281 (BasicConv2d 3->32, 3x3 stride 2, folded BN, ReLU) straight from the caller's NCHW
// tensor: no NHWC8 staging pass (it wrote and re-read 105 MB per 256 images) and no 3->8 channel padding in an
// MFMA K of 72 that is 62 % zeros.  One thread per output pixel, 32 channels as 16 packed-FMA pairs in the
// reference's (c,kh,kw) order, fp32 weights [27][32] read as wave-uniform scalars, one 64/128-byte NHWC row out.
typedef float f2_t __attribute__((ext_vector_type(2)));

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) stem_conv1a_kernel(const TI* __restrict__ x, TO* __restrict__ y, int ldy, int n,
                                                          const float* __restrict__ wt) {
  constexpr int S = 160, SO = 79;
  const unsigned total = (unsigned)n * SO * SO;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned img = i / (SO * SO), p = i - img * (SO * SO);
  const int oy = (int)(p / SO), ox = (int)(p - (unsigned)oy * SO);
  const TI* src = x + (size_t)img * 3 * S * S + (size_t)(2 * oy) * S + 2 * ox;
  float in[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) in[(c * 3 + kh) * 3 + kw] = to_f(src[(size_t)c * S * S + kh * S + kw]);
  f2_t acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = f2_t{wt[27 * 32 + 2 * j], wt[27 * 32 + 2 * j + 1]};
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const f2_t v2 = {in[k], in[k]};
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = __builtin_elementwise_fma(v2, f2_t{wt[k * 32 + 2 * j], wt[k * 32 + 2 * j + 1]}, acc[j]);
  }
  if constexpr (__is_same(TO, pf16)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = fmaxf(acc[4 * u + j][0], 0.f); v[2 * j + 1] = fmaxf(acc[4 * u + j][1], 0.f); }
      store_unit_pf16(y + (size_t)i * ldy + u * 8, v);
    }
  } else {
    TO o[32];
#pragma unroll
    for (int j = 0; j < 16; ++j) { o[2 * j] = (TO)fmaxf(acc[j][0], 0.f); o[2 * j + 1] = (TO)fmaxf(acc[j][1], 0.f); }
    uint4* dst = reinterpret_cast<uint4*>(y + (size_t)i * ldy);
#pragma unroll
    for (int q = 0; q < (int)(32 * sizeof(TO) / 16); ++q) dst[q] = reinterpret_cast<const uint4*>(o)[q];
  }
}

// The same convolution on the f32 MFMA (v_mfma_f32_16x16x4_f32: the f32 vector rate, exact fmaf chain in k order, so
// bit-identical to the kernel above), for the 16-bit and planar split-f16 plans.  The VALU form reads its 27 x 32 weights
// as wave-uniform scalars at every tap -- 864 values do not fit the SGPR file, and the refetch, not the FMAs, set its
// time in the split-f16 plan (0.098 -> 0.07 ms; bf16: 0.062 -> 0.059, where the rest is index arithmetic and the two
// memory streams).  Here the weights are the A operand: lane (row r, k slot g) keeps
// w[k = 4t + g][channel 16 ct + r] for the 7 k-steps and both channel tiles in 14 VGPRs, loaded once; B = 16 output
// pixels, lane (pixel, g) fetching tap k = 4t + g of its pixel; D = channels 4g .. 4g+3 of the pixel per tile.  A lane-row
// swap between the two tiles gives every lane 8 channels = one 16-byte store (planar: one hi and one lo chunk).
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) stem_conv1a_mfma_kernel(const TI* __restrict__ x, TO* __restrict__ y, int ldy, int n,
                                                               const float* __restrict__ wt) {
  constexpr int S = 160, SO = 79, NPX = SO * SO, G = 4;   // G pixel groups of 16 per wave iteration
  const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  const unsigned total = (unsigned)n * NPX;
  const unsigned ngroups = (total + 15) / 16;
  const unsigned nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  float wa[2][7], bias[2][4];
  int koff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int k = 4 * t + g;
    wa[0][t] = k < 27 ? wt[k * 32 + col] : 0.f;
    wa[1][t] = k < 27 ? wt[k * 32 + 16 + col] : 0.f;
    koff[t] = k < 27 ? (k / 9) * S * S + ((k % 9) / 3) * S + k % 3 : 0;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[ct][e] = wt[27 * 32 + 16 * ct + 4 * g + e];
  const int cst = (g & 1) * 16 + (g >> 1) * 8;   // the 8 channels this lane stores after the swap
  for (unsigned gi = w0 * G; gi < ngroups; gi += nwaves * G) {
    float xb[G][7];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const unsigned i = min((gi + q) * 16 + col, total - 1);
      const unsigned img = i / NPX, p = i - img * NPX;
      const int oy = (int)(p / SO), ox = (int)(p - (unsigned)oy * SO);
      const TI* src = x + (size_t)img * 3 * S * S + (size_t)(2 * oy) * S + 2 * ox;
#pragma unroll
      for (int t = 0; t < 7; ++t) xb[q][t] = to_f(src[koff[t]]);
    }
#pragma unroll
    for (int q = 0; q < G; ++q) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 a0 = {bias[0][0], bias[0][1], bias[0][2], bias[0][3]}, a1 = {bias[1][0], bias[1][1], bias[1][2], bias[1][3]};
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][t], xb[q][t], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][t], xb[q][t], a1, 0, 0, 0);
      }
      const unsigned i = (gi + q) * 16 + col;
      const bool ok = gi + q < ngroups && i < total;
      if constexpr (__is_same(TO, pf16)) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 h0, l0, h1, l1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const sf16 s0(fmaxf(a0[e], 0.f)), s1(fmaxf(a1[e], 0.f));
          h0[e] = s0.hi; l0[e] = s0.lo;
          h1[e] = s1.hi; l1[e] = s1.lo;
        }
        const uint2 ph0 = __builtin_bit_cast(uint2, h0), ph1 = __builtin_bit_cast(uint2, h1);
        const uint2 pl0 = __builtin_bit_cast(uint2, l0), pl1 = __builtin_bit_cast(uint2, l1);
        const auto hx = __builtin_amdgcn_permlane16_swap(ph0.x, ph1.x, false, false);
        const auto hy = __builtin_amdgcn_permlane16_swap(ph0.y, ph1.y, false, false);
        const auto lx = __builtin_amdgcn_permlane16_swap(pl0.x, pl1.x, false, false);
        const auto ly = __builtin_amdgcn_permlane16_swap(pl0.y, pl1.y, false, false);
        if (ok) {
          char* d = reinterpret_cast<char*>(y) + ((size_t)i * ldy + cst) * 4;
          *reinterpret_cast<uint4*>(d) = uint4{hx[0], hy[0], hx[1], hy[1]};
          *reinterpret_cast<uint4*>(d + 16) = uint4{lx[0], ly[0], lx[1], ly[1]};
        }
      } else {
        typedef TO t4 __attribute__((ext_vector_type(4)));
        t4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] = (TO)fmaxf(a0[e], 0.f); o1[e] = (TO)fmaxf(a1[e], 0.f); }
        const uint2 p0 = __builtin_bit_cast(uint2, o0), p1 = __builtin_bit_cast(uint2, o1);
        const auto sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
        if (ok) *reinterpret_cast<uint4*>(y + (size_t)i * ldy + cst) = uint4{sx[0], sy[0], sx[1], sy[1]};
      }
    }
  }
}

template <typename TI>
static hipError_t stem_dispatch_out(const void* x, void* y, int ldy, int dtype, int n, const float* wt, hipStream_t s) {
  const char* e1a = getenv("VNF_STEM1A_MFMA");   // read per launch: the parity test flips it inside one process
  const bool mfma = !(e1a && atoi(e1a) == 0);
  if (mfma && n > 0 && (dtype == BF16 || dtype == F16 || dtype == F16P)) {
    const unsigned ngroups = ((unsigned)n * 79 * 79 + 15) / 16;
    const int blocks = (int)((ngroups + 15) / 16 < 2048 ? (ngroups + 15) / 16 : 2048);   // 4 waves x 4 groups per block round
    if (dtype == BF16) hipLaunchKernelGGL((stem_conv1a_mfma_kernel<TI, __bf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (__bf16*)y, ldy, n, wt);
    else if (dtype == F16) hipLaunchKernelGGL((stem_conv1a_mfma_kernel<TI, _Float16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (_Float16*)y, ldy, n, wt);
    else hipLaunchKernelGGL((stem_conv1a_mfma_kernel<TI, pf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (pf16*)y, ldy, n, wt);
    return hipGetLastError();
  }
  const unsigned total = (unsigned)n * 79 * 79;
  const int blocks = (int)((total + 255) / 256);
  if (blocks == 0) return hipSuccess;
  switch (dtype) {
    case BF16: hipLaunchKernelGGL((stem_conv1a_kernel<TI, __bf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (__bf16*)y, ldy, n, wt); break;
    case F16: hipLaunchKernelGGL((stem_conv1a_kernel<TI, _Float16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (_Float16*)y, ldy, n, wt); break;
    case F32: hipLaunchKernelGGL((stem_conv1a_kernel<TI, float>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (float*)y, ldy, n, wt); break;
    case F16X2: hipLaunchKernelGGL((stem_conv1a_kernel<TI, sf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (sf16*)y, ldy, n, wt); break;
    case F16P: hipLaunchKernelGGL((stem_conv1a_kernel<TI, pf16>), dim3(blocks), dim3(256), 0, s, (const TI*)x, (pf16*)y, ldy, n, wt); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_stem_conv1a(const void* x, int x_dtype, void* y, int ldy, int dtype, int n, const float* wt, hipStream_t s) {
  switch (x_dtype) {
    case F32: return stem_dispatch_out<float>(x, y, ldy, dtype, n, wt, s);
    case BF16: return stem_dispatch_out<__bf16>(x, y, ldy, dtype, n, wt, s);
    case F16: return stem_dispatch_out<_Float16>(x, y, ldy, dtype, n, wt, s);
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------- max pool 3x3 stride 2
template <typename T>
__global__ void maxpool3s2_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int H, int W,
                                  int C) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1, cc = C / CH;
  const unsigned total = (unsigned)n * Ho * Wo * cc;  // < 2^31, checked by the launcher: 32-bit index math
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned p = i / cc;
    const int c = (int)(i - p * cc) * CH;
    const unsigned q = p / Wo;
    const int wo = (int)(p - q * Wo);
    const unsigned img = q / Ho;
    const int ho = (int)(q - img * Ho);
    float m[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) m[e] = -3.4e38f;
    const T* xi = x + ((size_t)img * H * W + (size_t)(2 * ho) * W + 2 * wo) * ldx + c;
    T v[9][CH];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw)   // nine independent 16-byte loads in flight
        *reinterpret_cast<uint4*>(v[dh * 3 + dw]) = *reinterpret_cast<const uint4*>(xi + (size_t)(dh * W + dw) * ldx);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < CH; ++e) m[e] = fmaxf(m[e], (float)v[t][e]);
    T o[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) o[e] = (T)m[e];
    *reinterpret_cast<uint4*>(y + (size_t)p * ldy + c) = *reinterpret_cast<const uint4*>(o);
  }
}

// planar split-f16: one thread per (pixel, 8-channel unit); max of the recombined values, re-split (exact: the pair
// of the largest value is reproduced bit for bit)
__global__ void maxpool3s2_pf16_kernel(const pf16* __restrict__ x, int ldx, pf16* __restrict__ y, int ldy, int n, int H, int W, int C) {
  const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1, cc = C / 8;
  const unsigned total = (unsigned)n * Ho * Wo * cc;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned p = i / cc;
    const int c = (int)(i - p * cc) * 8;
    const unsigned q = p / Wo;
    const int wo = (int)(p - q * Wo);
    const unsigned img = q / Ho;
    const int ho = (int)(q - img * Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -3.4e38f;
    const pf16* xi = x + ((size_t)img * H * W + (size_t)(2 * ho) * W + 2 * wo) * ldx + c;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        float v[8];
        load_unit_pf16(xi + (size_t)(dh * W + dw) * ldx, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    store_unit_pf16(y + (size_t)p * ldy + c, m);
  }
}

hipError_t launch_maxpool3s2(const void* x, int ldx, void* y, int ldy, int dtype, int n, int H, int W, int C,
                             hipStream_t s) {
  const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
  const int ch = dtype_chan_align(dtype);
  if (C % ch) return hipErrorInvalidValue;
  const size_t total = (size_t)n * Ho * Wo * (C / ch);
  if (total == 0) return hipSuccess;
  if (total >= (1u << 31)) return hipErrorInvalidValue;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  switch (dtype) {
    case BF16: hipLaunchKernelGGL(maxpool3s2_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, ldx, (__bf16*)y, ldy, n, H, W, C); break;
    case F16: hipLaunchKernelGGL(maxpool3s2_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)x, ldx, (_Float16*)y, ldy, n, H, W, C); break;
    case F32: hipLaunchKernelGGL(maxpool3s2_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, n, H, W, C); break;
    case F16X2: hipLaunchKernelGGL(maxpool3s2_kernel<sf16>, dim3(blocks), dim3(256), 0, s, (const sf16*)x, ldx, (sf16*)y, ldy, n, H, W, C); break;
    case F16P: hipLaunchKernelGGL(maxpool3s2_pf16_kernel, dim3(blocks), dim3(256), 0, s, (const pf16*)x, ldx, (pf16*)y, ldy, n, H, W, C); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------- max pool k x k stride 2, ceil_mode=True
// (MTCNN R/O-Net pools, mtcnn.py:64,67,114,117,120): windows may hang over the border.
template <typename T>
__global__ void maxpool_ceil_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int n, int H, int W,
                                    int C, int k) {
  constexpr int CH = 16 / (int)sizeof(T);
  const int Ho = (H - k + 1) / 2 + 1, Wo = (W - k + 1) / 2 + 1, cc = C / CH;
  const unsigned total = (unsigned)n * Ho * Wo * cc;  // < 2^31, checked by the launcher: 32-bit index math
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned p = i / cc;
    const int c = (int)(i - p * cc) * CH;
    const unsigned q = p / Wo;
    const int wo = (int)(p - q * Wo);
    const unsigned img = q / Ho;
    const int ho = (int)(q - img * Ho);
    float m[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) m[e] = -INFINITY;
    const T* xi = x + (size_t)img * H * W * ldx + c;
    for (int dh = 0; dh < k; ++dh)
      for (int dw = 0; dw < k; ++dw) {
        const int yy = 2 * ho + dh, xx = 2 * wo + dw;
        if (yy < H && xx < W) {
          T v[CH];
          *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(xi + (unsigned)(yy * W + xx) * (size_t)ldx);
#pragma unroll
          for (int e = 0; e < CH; ++e) m[e] = fmaxf(m[e], (float)v[e]);
        }
      }
    T o[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) o[e] = (T)m[e];
    *reinterpret_cast<uint4*>(y + (size_t)p * ldy + c) = *reinterpret_cast<const uint4*>(o);
  }
}

hipError_t launch_maxpool_ceil(const void* x, int ldx, void* y, int ldy, int dtype, int n, int H, int W, int C, int k,
                               hipStream_t s) {
  const int Ho = (H - k + 1) / 2 + 1, Wo = (W - k + 1) / 2 + 1;
  const int ch = 16 / dtype_size(dtype);
  if (C % ch) return hipErrorInvalidValue;
  const size_t total = (size_t)n * Ho * Wo * (C / ch);
  if (total == 0) return hipSuccess;
  if (total >= (1u << 31)) return hipErrorInvalidValue;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  switch (dtype) {
    case BF16: hipLaunchKernelGGL(maxpool_ceil_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, ldx, (__bf16*)y, ldy, n, H, W, C, k); break;
    case F16: hipLaunchKernelGGL(maxpool_ceil_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)x, ldx, (_Float16*)y, ldy, n, H, W, C, k); break;
    case F32: hipLaunchKernelGGL(maxpool_ceil_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, n, H, W, C, k); break;
    case F16X2: hipLaunchKernelGGL(maxpool_ceil_kernel<sf16>, dim3(blocks), dim3(256), 0, s, (const sf16*)x, ldx, (sf16*)y, ldy, n, H, W, C, k); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------- global average pool
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int n, int HW, int C) {
  const size_t total = (size_t)n * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / C;
    const int c = (int)(i - img * C);
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += (float)x[(img * HW + p) * (size_t)ldx + c];
    y[i] = (T)(s / (float)HW);
  }
}

__global__ void avgpool_pf16_kernel(const pf16* __restrict__ x, int ldx, pf16* __restrict__ y, int n, int HW, int C) {
  const size_t total = (size_t)n * (C / 8);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / (C / 8);
    const int c = (int)(i - img * (C / 8)) * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < HW; ++p) {
      float v[8];
      load_unit_pf16(x + (img * HW + p) * (size_t)ldx + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = s[e] / (float)HW;
    store_unit_pf16(y + img * C + c, s);
  }
}

hipError_t launch_avgpool(const void* x, int ldx, void* y, int dtype, int n, int HW, int C, hipStream_t s) {
  const size_t total = (size_t)n * C;
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256);
  if (dtype == F16P) {
    if (C % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avgpool_pf16_kernel, dim3((int)((total / 8 + 255) / 256)), dim3(256), 0, s, (const pf16*)x, ldx, (pf16*)y, n, HW, C);
    return hipGetLastError();
  }
  switch (dtype) {
    case BF16: hipLaunchKernelGGL(avgpool_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, ldx, (__bf16*)y, n, HW, C); break;
    case F16: hipLaunchKernelGGL(avgpool_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)x, ldx, (_Float16*)y, n, HW, C); break;
    case F32: hipLaunchKernelGGL(avgpool_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, ldx, (float*)y, n, HW, C); break;
    case F16X2: hipLaunchKernelGGL(avgpool_kernel<sf16>, dim3(blocks), dim3(256), 0, s, (const sf16*)x, ldx, (sf16*)y, n, HW, C); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------- row-wise L2 normalisation
// one wave per row: F.normalize(x, p=2, dim=1) = x / max(||x||, 1e-12) (inception_resnet_v1.py:302)
__global__ void l2norm_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* xr = x + (size_t)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c] * xr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float d = fmaxf(sqrtf(s), 1e-12f);
  for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = xr[c] / d;
}

hipError_t launch_l2norm(const float* x, float* y, int n, int C, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(l2norm_kernel, dim3((n + 3) / 4), dim3(256), 0, s, x, y, n, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------- log_softmax + argmax + prob
// one wave per row of logits (mlp_model.py:14 log_softmax; demo_image.py:125-129 argmax, exp)
__global__ void logsoftmax_argmax_kernel(const float* __restrict__ logits, int ld, int C, int n,
                                         float* __restrict__ logp, int32_t* __restrict__ amax,
                                         float* __restrict__ prob) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* x = logits + (size_t)row * ld;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > m) { m = v; mi = c; }   // first occurrence within the lane's stride
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o);
    const int oi = __shfl_xor(mi, o);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float ls = logf(s);
  if (logp)
    for (int c = lane; c < C; c += 64) logp[(size_t)row * C + c] = (x[c] - m) - ls;
  if (lane == 0) {
    if (amax) amax[row] = mi;
    if (prob) prob[row] = expf(-ls);  // exp(logp[argmax]) with logp[argmax] = 0 - log(sum)
  }
}

hipError_t launch_logsoftmax_argmax(const float* logits, int ld, int C, int n, float* logp, int32_t* amax, float* prob,
                                    hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(logsoftmax_argmax_kernel, dim3((n + 3) / 4), dim3(256), 0, s, logits, ld, C, n, logp, amax, prob);
  return hipGetLastError();
}

// ---------------------------------------------------------------- depthwise 3x3 (MobileNetV1 blocks of RetinaFace)
// HBM-bound: one thread per (pixel, 4 channels), nine 16-byte loads, taps in (kh, kw) order as the reference's
// grouped conv sums them
// SPLIT: the tensor holds split-f16 (hi, lo) pairs in its 32-bit elements (the F16X2 plans) instead of fp32
template <bool SPLIT> __device__ __forceinline__ float ldv(float raw) { return SPLIT ? (float)__builtin_bit_cast(sf16, raw) : raw; }
template <bool SPLIT> __device__ __forceinline__ float stv(float v) { return SPLIT ? __builtin_bit_cast(float, sf16(v)) : v; }

template <bool SPLIT>
__global__ void dwconv3x3_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int H, int W, int C, int stride,
                                 int Ho, int Wo, const float* __restrict__ w9c, const float* __restrict__ bias, float slope) {
  const int c4 = C >> 2;
  const size_t total = (size_t)n * Ho * Wo * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    const size_t p = i / c4;
    const int wo = (int)(p % Wo);
    const size_t q = p / Wo;
    const int ho = (int)(q % Ho);
    const size_t img = q / Ho;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hi = ho * stride - 1 + kh, wi = wo * stride - 1 + kw;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + ((img * H + hi) * W + wi) * C + c);
          const f32x4_t ww = *reinterpret_cast<const f32x4_t*>(w9c + (kh * 3 + kw) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = acc[e] + ldv<SPLIT>(v[e]) * ww[e];
        }
      }
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(bias + c);
    f32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = acc[e] + b[e];
      o[e] = stv<SPLIT>(v > 0.f ? v : v * slope);
    }
    *reinterpret_cast<f32x4_t*>(y + p * C + c) = o;
  }
}

hipError_t launch_dwconv3x3(const float* x, float* y, int n, int H, int W, int C, int stride, const float* w9c,
                            const float* bias, float slope, bool split, hipStream_t s) {
  if (C % 4 || (stride != 1 && stride != 2)) return hipErrorInvalidValue;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const size_t total = (size_t)n * Ho * Wo * (C / 4);
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (split) hipLaunchKernelGGL(dwconv3x3_kernel<true>, dim3(blocks), dim3(256), 0, s, x, y, n, H, W, C, stride, Ho, Wo, w9c, bias, slope);
  else hipLaunchKernelGGL(dwconv3x3_kernel<false>, dim3(blocks), dim3(256), 0, s, x, y, n, H, W, C, stride, Ho, Wo, w9c, bias, slope);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// RetinaFace / MobileNetV1-0.25 early layers on the MFMA lane layout (fp32, exact v_mfma_f32_16x16x4_f32).
//
// The first layers of that network are pure data movement (a 1080p frame is 33 MB as NHWC4 fp32, conv0's output 133 MB,
// the 8->16 pointwise output 265 MB per 8 frames ...), with channel counts (3, 8, 16, 32) far below what the tiled
// implicit-GEMM kernel is built for.  Two kernels replace them:
//   * retina_stem_kernel: u8 RGB frame -> (x - mean) -> 3x3 stride-2 pad-1 conv 3->8 + folded BN + LeakyReLU, straight
//     from the frame bytes (no NHWC4 fp32 staging tensor).  One wave = 16 output pixels: B = the 27 taps of a pixel
//     (k = (kh*3+kw)*3 + c, lane group = k mod 4), A = weights (7 VGPRs), one MFMA per 4 k.
//   * dwpw_kernel<CIN,COUT>: depthwise 3x3 (+BN+LeakyReLU) and the pointwise 1x1 (+BN+LeakyReLU) that follows it in
//     conv_dw (components.py:30-40) in one pass: lane (pixel l&15, group l>>4) computes the depthwise outputs of
//     channels {4s + group}, which is exactly the B operand of k-step s of the pointwise GEMM; the depthwise tensor
//     never reaches memory.  Depthwise arithmetic (mul, add, tap order) is that of dwconv3x3_kernel.
struct RetinaStemW { const float* wa; const float* bias; float slope; };   // wa: [7][64] lane table, bias[8]

template <bool SPLIT>
__global__ void __launch_bounds__(256) retina_stem_kernel(const uint8_t* __restrict__ frames, int n, int H, int W, int Ho, int Wo,
                                                           RetinaStemW w, float* __restrict__ y) {
  // (tried and slower: four tiles in flight per wave, 0.15 ms per 8 frames; rows staged as aligned dwords in a
  // wave-private LDS strip, 0.19 ms; this form: 0.13 ms against 0.46 ms for the staging pass + plan convolution)
  const int lane = threadIdx.x & 63, lg = lane >> 4, lm = lane & 15;
  float wa[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) wa[s] = w.wa[s * 64 + lane];
  float b4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) b4[e] = lg < 2 ? w.bias[lg * 4 + e] : 0.f;
  const int tiles_w = (Wo + 15) >> 4;
  const long long ntile = (long long)n * Ho * tiles_w;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
  for (long long t = wave; t < ntile; t += nwave) {
    const int tw = (int)(t % tiles_w);
    const long long q = t / tiles_w;
    const int ho = (int)(q % Ho), img = (int)(q / Ho);
    const int wo = tw * 16 + lm;
    const uint8_t* fb = frames + (size_t)img * H * W * 3;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const int k = 4 * s + lg, tap = k / 3, c = k - tap * 3, kh = tap / 3, kw = tap - kh * 3;
      const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
      // clamped address + select on the value: the seven byte loads issue back to back (a predicated load is a branch
      // and a wait per tap)
      const bool ok = k < 27 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1), cc = min(c, 2);
      const float raw = (float)fb[((size_t)hc * W + wc) * 3 + cc] - (cc == 0 ? 104.f : (cc == 1 ? 117.f : 123.f));
      const float v = ok ? raw : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], v, acc, 0, 0, 0);
    }
    if (lg < 2 && wo < Wo) {
      f32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[e] + b4[e];
        o[e] = stv<SPLIT>(v > 0.f ? v : v * w.slope);
      }
      *reinterpret_cast<f32x4_t*>(y + (((size_t)img * Ho + ho) * Wo + wo) * 8 + lg * 4) = o;
    }
  }
}

hipError_t launch_retina_stem(const uint8_t* frames, int n, int H, int W, const float* wa, const float* bias, float slope, float* y,
                              bool split, hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long ntile = (long long)n * Ho * ((Wo + 15) / 16);
  if (ntile == 0) return hipSuccess;
  const int blocks = (int)std::min<long long>((ntile + 3) / 4, 8192);
  if (split) hipLaunchKernelGGL(retina_stem_kernel<true>, dim3(blocks), dim3(256), 0, s, frames, n, H, W, Ho, Wo, RetinaStemW{wa, bias, slope}, y);
  else hipLaunchKernelGGL(retina_stem_kernel<false>, dim3(blocks), dim3(256), 0, s, frames, n, H, W, Ho, Wo, RetinaStemW{wa, bias, slope}, y);
  return hipGetLastError();
}

struct DwPwW { const float *dw, *dbias, *pw, *pbias; float dslope, pslope; };   // dw [9][CIN], pw [COUT][CIN] (BN folded)

template <int CIN, int COUT, bool SPLIT>
__global__ void __launch_bounds__(256) dwpw_kernel(const float* __restrict__ x, int n, int H, int W, int stride, int Ho, int Wo, DwPwW w,
                                                    float* __restrict__ y) {
  // lane group g owns the CONTIGUOUS channels [g*KS, g*KS + KS): k-step s of the pointwise GEMM pairs channel g*KS + s of
  // every group (any consistent k permutation is a valid dot product), and a lane's depthwise inputs are vector loads
  constexpr int KS = CIN / 4, NT = COUT / 16, VL = KS < 4 ? KS : 4, NV = KS / VL;
  typedef float vec_t __attribute__((ext_vector_type(VL)));
  const int lane = threadIdx.x & 63, lg = lane >> 4, lm = lane & 15;
  float wd[9][KS], bd[KS], wp[NT][KS], bp[NT][4];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    bd[s] = w.dbias[lg * KS + s];
#pragma unroll
    for (int t = 0; t < 9; ++t) wd[t][s] = w.dw[t * CIN + lg * KS + s];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wp[nt][s] = w.pw[(size_t)(16 * nt + lm) * CIN + lg * KS + s];
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) bp[nt][e] = w.pbias[16 * nt + 4 * lg + e];
  const int tiles_w = (Wo + 15) >> 4;
  const long long ntile = (long long)n * Ho * tiles_w;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
  for (long long t = wave; t < ntile; t += nwave) {
    const int tw = (int)(t % tiles_w);
    const long long q = t / tiles_w;
    const int ho = (int)(q % Ho), img = (int)(q / Ho);
    const int wo = min(tw * 16 + lm, Wo - 1);          // tail lanes recompute the last pixel and do not store
    const float* xb = x + (size_t)img * H * W * CIN + lg * KS;
    // all 9 x NV loads first (clamped addresses, no predicate), then the arithmetic in dwconv3x3_kernel's order
    vec_t in[9][NV];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hi = ho * stride - 1 + kh, wi = wo * stride - 1 + kw;
        ok[kh * 3 + kw] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const float* px = xb + ((size_t)min(max(hi, 0), H - 1) * W + min(max(wi, 0), W - 1)) * CIN;
#pragma unroll
        for (int v = 0; v < NV; ++v) in[kh * 3 + kw][v] = *reinterpret_cast<const vec_t*>(px + v * VL);
      }
    float d[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) d[s] = 0.f;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float xv = ldv<SPLIT>(in[tp][s / VL][s % VL]);
        d[s] = ok[tp] ? d[s] + xv * wd[tp][s] : d[s];
      }
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float v = d[s] + bd[s];
      v = v > 0.f ? v : v * w.dslope;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[nt][s], v, acc[nt], 0, 0, 0);
    }
    if (tw * 16 + lm < Wo) {
      float* o = y + (((size_t)img * Ho + ho) * Wo + wo) * COUT + 4 * lg;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4_t r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[nt][e] + bp[nt][e];
          r[e] = stv<SPLIT>(v > 0.f ? v : v * w.pslope);
        }
        *reinterpret_cast<f32x4_t*>(o + 16 * nt) = r;
      }
    }
  }
}

bool dwpw_supported(int cin, int cout) {
  return (cin == 8 && cout == 16) || (cin == 16 && cout == 32) || (cin == 32 && cout == 32) || (cin == 32 && cout == 64);
}

hipError_t launch_dwpw(const float* x, float* y, int n, int H, int W, int cin, int cout, int stride, const float* dw, const float* dbias,
                       float dslope, const float* pw, const float* pbias, float pslope, bool split, hipStream_t s) {
  const int Ho = stride == 2 ? (H - 1) / 2 + 1 : H, Wo = stride == 2 ? (W - 1) / 2 + 1 : W;
  const long long ntile = (long long)n * Ho * ((Wo + 15) / 16);
  if (ntile == 0) return hipSuccess;
  const int blocks = (int)std::min<long long>((ntile + 3) / 4, 8192);
  const DwPwW w{dw, dbias, pw, pbias, dslope, pslope};
#define VNF_DWPW(CI, CO)                                                                                                    \
  do {                                                                                                                      \
    if (split) hipLaunchKernelGGL((dwpw_kernel<CI, CO, true>), dim3(blocks), dim3(256), 0, s, x, n, H, W, stride, Ho, Wo, w, y); \
    else hipLaunchKernelGGL((dwpw_kernel<CI, CO, false>), dim3(blocks), dim3(256), 0, s, x, n, H, W, stride, Ho, Wo, w, y);      \
  } while (0)
  if (cin == 8 && cout == 16) VNF_DWPW(8, 16);
  else if (cin == 16 && cout == 32) VNF_DWPW(16, 32);
  else if (cin == 32 && cout == 32) VNF_DWPW(32, 32);
  else if (cin == 32 && cout == 64) VNF_DWPW(32, 64);
  else return hipErrorInvalidValue;
#undef VNF_DWPW
  return hipGetLastError();
}

template <bool SPLIT>
__global__ void upsample_add_kernel(const float* __restrict__ x, int Hs, int Ws, float* __restrict__ y, int H, int W, int C,
                                    int n, float sh, float sw) {
  const int c4 = C >> 2;
  const size_t total = (size_t)n * H * W * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    const size_t p = i / c4;
    const int w = (int)(p % W);
    const size_t q = p / W;
    const int h = (int)(q % H);
    const size_t img = q / H;
    // ATen nearest: min(floor(dst * scale), in - 1) with scale = (float)in / out
    const int hs = min((int)floorf((float)h * sh), Hs - 1), ws = min((int)floorf((float)w * sw), Ws - 1);
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(y + p * C + c);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(x + ((img * Hs + hs) * Ws + ws) * C + c);
    f32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = stv<SPLIT>(ldv<SPLIT>(a[e]) + ldv<SPLIT>(b[e]));
    *reinterpret_cast<f32x4_t*>(y + p * C + c) = o;
  }
}

hipError_t launch_upsample_add(const float* x, int Hs, int Ws, float* y, int H, int W, int C, int n, bool split, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)n * H * W * (C / 4);
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (split)
    hipLaunchKernelGGL(upsample_add_kernel<true>, dim3(blocks), dim3(256), 0, s, x, Hs, Ws, y, H, W, C, n, (float)Hs / (float)H,
                       (float)Ws / (float)W);
  else
    hipLaunchKernelGGL(upsample_add_kernel<false>, dim3(blocks), dim3(256), 0, s, x, Hs, Ws, y, H, W, C, n, (float)Hs / (float)H,
                       (float)Ws / (float)W);
  return hipGetLastError();
}

// ---------------------------------------------------------------- NHWC slice -> NCHW fp32 (taps)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, int n, int HW, int C) {
  const size_t total = (size_t)n * HW * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const size_t q = i / HW;
    const int c = (int)(q % C);
    const size_t img = q / C;
    y[i] = (float)x[(img * HW + p) * (size_t)ldx + c];
  }
}

__global__ void nhwc_to_nchw_pf16_kernel(const pf16* __restrict__ x, int ldx, float* __restrict__ y, int n, int HW, int C) {
  const size_t total = (size_t)n * HW * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const size_t q = i / HW;
    const int c = (int)(q % C);
    const size_t img = q / C;
    // element c of the slice: unit c / 8 (32 bytes), hi half at 2 * (c % 8), lo half 16 bytes further (the slice starts
    // on a unit boundary)
    const _Float16* u = reinterpret_cast<const _Float16*>(x + (img * HW + p) * (size_t)ldx + (c & ~7));
    y[i] = (float)u[c & 7] + (float)u[8 + (c & 7)];
  }
}

hipError_t launch_nhwc_to_nchw_f32(const void* x, int ldx, int dtype, float* y, int n, int HW, int C, hipStream_t s) {
  const size_t total = (size_t)n * HW * C;
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == F16P) {
    hipLaunchKernelGGL(nhwc_to_nchw_pf16_kernel, dim3(blocks), dim3(256), 0, s, (const pf16*)x, ldx, y, n, HW, C);
    return hipGetLastError();
  }
  switch (dtype) {
    case BF16: hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, ldx, y, n, HW, C); break;
    case F16: hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)x, ldx, y, n, HW, C); break;
    case F32: hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, ldx, y, n, HW, C); break;
    case F16X2: hipLaunchKernelGGL(nhwc_to_nchw_kernel<sf16>, dim3(blocks), dim3(256), 0, s, (const sf16*)x, ldx, y, n, HW, C); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace vnf
