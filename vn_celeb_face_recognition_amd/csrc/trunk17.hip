// Persistent Block17 stack ("repeat_2": 10 x Block17, /root/reference/models/inception_resnet_v1.py:70-95,226-237)
// as ONE launch on gfx950: one workgroup (8 waves) = one 8x8x896 image, resident on its CU for the whole stack.
//
//   x -> [ reduce 1x1 896->(128|128) -> (1x7 128->128) -> (7x1 128->128) -> up 1x1 256->896, relu(x + up) ] x 10
//
// What stays where
//   * the residual trunk x (64 px x 896 ch) lives in REGISTERS as fp32 for all ten blocks, in the accumulator layout
//     of the block-output ("up") convolution: wave w owns the 16-channel tiles w, w+8, ..., w+48 x 64 pixels = 7 x 4
//     MFMA tiles = 112 VGPRs (tiles interleaved over the waves, so every LDS address of a lane is ONE per-lane base plus
//     compile-time offsets).  The up convolution accumulates straight into it (C-in = trunk), so the residual chain is never
//     rounded to 16 bits and never touches memory;
//   * every intermediate (the 16-bit copy of x the reduce conv consumes, branch outputs, the 256-channel concat) lives
//     in LDS in the K-tile image the MFMA B-fragment reads want ([k-tile][pixel row][128 B], 16-byte slots XOR-swizzled
//     by row & 7: conflict-free ds_read_b128, same image as conv_igemm.hip);
//   * weights are the only thing that streams: 1.34 MB per block.  Output channels are split over the 8 waves, so a
//     weight fragment is used by exactly ONE wave -- staging it in LDS would buy nothing.  The weights are repacked at
//     create time into one contiguous stream per wave in exactly the order the wave consumes its MFMA A-fragments
//     (1 KiB = 64 lanes x 16 B per fragment), and each wave runs its own register ring of 8 fragments in flight
//     (plain 16-byte global loads, counted waits left to the compiler): no barrier couples the waves inside a phase.
//
// Per block: 5 workgroup barriers (phase boundaries), 168 weight fragments and 672 MFMAs (16x16x32) per wave.
// The bound is the per-CU L2 -> register fill rate of the weight stream (13.4 MB per image-stack; every CU streams
// the same bytes at about the same time, so all but the first reader of a line hit L2), not HBM and not the MFMA pipe
// (0.88 GFLOP per image = ~90 us of matrix time per CU).
//
// Numerics: 16-bit MFMA operands (bf16 or f16) with fp32 accumulation, as the unfused plan; the trunk itself is fp32
// here (the unfused plan rounds it to 16 bits after every block), so this path is the more accurate of the two.
#include "conv_device.h"
#include "trunk17.h"

namespace vnf {

namespace {

constexpr int T17_PX = 64;
constexpr int KT_BYTES = 64 * 128;                 // one K tile (64 channels) of a 64-pixel image: [64 rows][128 B]
constexpr int OFF_XB = 0;                          // 14 K tiles: 16-bit copy of the trunk (phase 0/1)
constexpr int OFF_TB = 0;                          // 2 K tiles, aliases Xb (dead after phase 1): 1x7 output
constexpr int OFF_CATHI = 2 * KT_BYTES;            // 2 K tiles, aliases Xb: concat channels 128..255 (7x1 output)
constexpr int OFF_CATLO = 14 * KT_BYTES;           // 2 K tiles: concat channels 0..127 (branch0)
constexpr int OFF_TA = 16 * KT_BYTES;              // 2 K tiles: branch1.0 output
constexpr int OFF_ZERO = 18 * KT_BYTES;            // 256 zero bytes: out-of-image taps
constexpr int T17_LDS = 18 * KT_BYTES + 256;
constexpr int RING = 8;                            // weight fragments in flight per wave (divides T17_FRAGS)
static_assert(T17_FRAGS % RING == 0, "the ring index of a step must not depend on the block");

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
  static __device__ __forceinline__ f32x4_t run(const uint4& w, const uint4& x, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
  }
};
template <> struct Mma<_Float16> {
  static __device__ __forceinline__ f32x4_t run(const uint4& w, const uint4& x, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ uint2 pack4(const f32x4_t& v) {
  typedef T t4 __attribute__((ext_vector_type(4)));
  t4 r = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
  return __builtin_bit_cast(uint2, r);
}
template <typename T>
__device__ __forceinline__ f32x4_t unpack4(const uint2& u) {
  typedef T t4 __attribute__((ext_vector_type(4)));
  t4 r = __builtin_bit_cast(t4, u);
  return f32x4_t{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
}

}  // namespace

template <typename T>
__global__ __launch_bounds__(512, 2) void block17_trunk_kernel(const Trunk17Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow_c = lane & 15, fgrp_c = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);

  // ---- image -> LDS (the K-tile image of phase 1), by LDS-DMA: piece = 8 pixel rows x 128 B of one K tile
  {
    const char* src = (const char*)a.x + (size_t)img * T17_PX * a.ldx * 2;
    const int r = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int piece = wave * 14 + i;            // 112 pieces: [kt 0..13][row group 0..7]
      const int kt = piece >> 3, p = (piece & 7) * 8 + r;
      glds16(src + ((size_t)p * a.ldx + kt * 64 + ((slot ^ (p & 7)) << 3)) * 2, lds0 + OFF_XB + kt * KT_BYTES + (piece & 7) * 1024);
    }
    if (tid < 16) reinterpret_cast<uint4*>(smem + OFF_ZERO)[tid] = uint4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  // Per-lane LDS bases.  A wave's accumulator tiles cover channels 128*j + 16*wave + 4*fgrp + (0..3) of pixel 16*i + frow:
  // inside a [k-tile][64 rows][128 B] image that is wbase + j * 2 k-tiles + i * 2048 (the swizzle only involves
  // row & 7 = frow & 7 and the channel bits below 128).  B-fragment reads of pixel 16*i + frow, k-step ks of k-tile kt:
  // kt * KT_BYTES + i * 2048 + rb[ks & 1].
  const int cw = 16 * wave + 4 * fgrp_c;  // channel offset inside a 128-channel group
  const int wbase_c = (cw >> 6) * KT_BYTES + frow_c * 128 + ((((cw & 63) >> 3) ^ (frow_c & 7)) << 4) + (cw & 4) * 2;
  const int rb0_c = frow_c * 128 + ((fgrp_c ^ (frow_c & 7)) << 4), rb1_c = rb0_c ^ 64;

  // ---- trunk registers
  f32x4_t trunk[7][4];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      trunk[j][i] = unpack4<T>(*reinterpret_cast<const uint2*>(smem + OFF_XB + wbase_c + j * 2 * KT_BYTES + i * 2048));

  // ---- weight stream of this wave: fragment f of block b at wbase + (b * T17_FRAGS + f) * 64 + lane (uint4 units)
  const uint4* wp = reinterpret_cast<const uint4*>(a.wstream) + (size_t)wave * ((size_t)a.nblocks * T17_FRAGS + RING) * 64 + lane;
  uint4 wq[RING];
#pragma unroll
  for (int r = 0; r < RING; ++r) wq[r] = wp[r * 64];
  wp += RING * 64;  // next fragment to fetch

  for (int b = 0; b < a.nblocks; ++b) {
    const float* bias = a.bias + (size_t)b * T17_BIAS + cw;
    // Every LDS address below is invariant over the blocks, and the compiler would hoist a few hundred of them out of
    // this loop into registers it does not have (they spill).  Re-deriving them from values it cannot see through keeps
    // them where they are used: one or two v_add per fragment read, next to MFMAs.
    int wbase = wbase_c, rb0 = rb0_c, rb1 = rb1_c, frow = frow_c, fgrp = fgrp_c;
    asm volatile("" : "+v"(wbase), "+v"(rb0), "+v"(rb1), "+v"(frow), "+v"(fgrp));
    // ---------------------------------------------------------------- phase 0: trunk -> 16-bit K-tile image
    if (b > 0) {
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<uint2*>(smem + OFF_XB + wbase + j * 2 * KT_BYTES + i * 2048) = pack4<T>(trunk[j][i]);
      __syncthreads();
    }
    // ---------------------------------------------------------------- phase 1: reduce 1x1, 896 -> 256 (tiles w, w+8 per wave)
    {
      f32x4_t acc[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 128 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = bv;
      }
#pragma unroll
      for (int ks = 0; ks < 28; ++ks) {
        uint4 xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          xf[i] = *reinterpret_cast<const uint4*>(smem + OFF_XB + (ks >> 1) * KT_BYTES + i * 2048 + ((ks & 1) ? rb1 : rb0));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int s = ks * 2 + j;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = Mma<T>::run(wq[s % RING], xf[i], acc[j][i]);
          wq[s % RING] = *wp;
          wp += 64;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep one k-step's fragments live at a time (register budget)
      }
      // ReLU -> 16 bits -> branch0 half of the concat (channels 0..127, j = 0) / branch1.0 output (128..255, j = 1)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4_t v = acc[j][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          *reinterpret_cast<uint2*>(smem + (j == 0 ? OFF_CATLO : OFF_TA) + wbase + i * 2048) = pack4<T>(v);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- phases 2, 3: (1,7) then (7,1), 128 -> 128 (tile w per wave)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int src_off = ph == 0 ? OFF_TA : OFF_TB;
      const int dst_off = ph == 0 ? OFF_TB : OFF_CATHI;
      f32x4_t acc[4];
      {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 256 + 128 * ph);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = bv;
      }
#pragma unroll
      for (int ks = 0; ks < 28; ++ks) {
        const int tap = ks >> 2;                    // k = tap * 128 + channel
        const int kt = (ks >> 1) & 1, kh = ks & 1;  // K tile and half of the 128-channel source image
        uint4 xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int off;
          bool ok;
          if (ph == 0) {   // pixel (y, x + tap - 3): the row's swizzle changes with x
            const int xx = (frow & 7) + tap - 3;
            ok = (unsigned)xx < 8u;
            off = src_off + kt * KT_BYTES + i * 2048 + (frow + tap - 3) * 128 + (((kh * 4 + fgrp) ^ (xx & 7)) << 4);
          } else {         // pixel (y + tap - 3, x): same swizzle, rows 8 apart
            const int yy = 2 * i + (frow >> 3) + tap - 3;
            ok = (unsigned)yy < 8u;
            off = src_off + kt * KT_BYTES + i * 2048 + (tap - 3) * 1024 + (kh ? rb1 : rb0);
          }
          xf[i] = *reinterpret_cast<const uint4*>(smem + (ok ? off : OFF_ZERO));
        }
        const int s = 56 + 28 * ph + ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = Mma<T>::run(wq[s % RING], xf[i], acc[i]);
        wq[s % RING] = *wp;
        wp += 64;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4_t v = acc[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        *reinterpret_cast<uint2*>(smem + dst_off + wbase + i * 2048) = pack4<T>(v);
      }
      __syncthreads();
    }
    // ---------------------------------------------------------------- phase 4: up 1x1, 256 -> 896, accumulated into the trunk
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int base = (ks < 4 ? OFF_CATLO : OFF_CATHI) + ((ks >> 1) & 1) * KT_BYTES;
      uint4 xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const uint4*>(smem + base + i * 2048 + ((ks & 1) ? rb1 : rb0));
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int s = 112 + ks * 7 + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) trunk[j][i] = Mma<T>::run(wq[s % RING], xf[i], trunk[j][i]);
        wq[s % RING] = *wp;
        wp += 64;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // relu(x + conv*scale + bias*scale): scale is folded into the weights and the bias
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 512 + 128 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) trunk[j][i][e] = fmaxf(trunk[j][i][e] + bv[e], 0.f);
    }
    __syncthreads();  // every wave is done reading the concat: the next block's phase 0 may overwrite it
  }

  // ---- trunk -> 16 bits -> LDS image -> whole NHWC rows of the output tensor
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint2*>(smem + OFF_XB + wbase_c + j * 2 * KT_BYTES + i * 2048) = pack4<T>(trunk[j][i]);
  __syncthreads();
  {
    char* dst = (char*)a.y + (size_t)img * T17_PX * a.ldy * 2;
#pragma unroll
    for (int it = 0; it < 14; ++it) {
      const int idx = it * 512 + tid;               // 7168 16-byte chunks: [pixel][112 chunks]
      const int p = idx / 112, ch = idx - p * 112;  // chunk ch = channels 8*ch .. 8*ch+7
      const uint4 v = *reinterpret_cast<const uint4*>(smem + OFF_XB + (ch >> 3) * KT_BYTES + p * 128 + (((ch & 7) ^ (p & 7)) << 4));
      *reinterpret_cast<uint4*>(dst + ((size_t)p * a.ldy + ch * 8) * 2) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight stream
// Fragment order of one wave within one block (must match the kernel's consumption order):
//   reduce : for ks in 0..27, j in 0..1   rows 128j+16w .. +15,  k = 32ks .. +31      (56)
//   1x7    : for ks in 0..27              rows 16w .. +15,       k = 32ks .. +31      (28)
//   7x1    : for ks in 0..27              rows 16w .. +15,       k = 32ks .. +31      (28)
//   up     : for ks in 0..7, j in 0..6    rows 128j+16w .. +15,  k = 32ks .. +31      (56)
// Lane l of a fragment holds the 8 k values k0 + 8*(l>>4) .. +7 of row r0 + (l&15) (the MFMA A-operand map).
__global__ void trunk17_repack_kernel(Trunk17Pack p, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t frag = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t per_wave = (size_t)p.nblocks * T17_FRAGS + RING;
  if (frag >= 8 * per_wave) return;
  const int wave = (int)(frag / per_wave);
  const size_t f = frag - (size_t)wave * per_wave;
  uint4 v = {0u, 0u, 0u, 0u};
  if (f < (size_t)p.nblocks * T17_FRAGS) {
    const int b = (int)(f / T17_FRAGS), s = (int)(f - (size_t)b * T17_FRAGS);
    int conv, r0, k0;
    if (s < 56) { conv = 0; r0 = 128 * (s & 1) + 16 * wave; k0 = 32 * (s >> 1); }
    else if (s < 84) { conv = 1; r0 = 16 * wave; k0 = 32 * (s - 56); }
    else if (s < 112) { conv = 2; r0 = 16 * wave; k0 = 32 * (s - 84); }
    else { conv = 3; r0 = 128 * ((s - 112) % 7) + 16 * wave; k0 = 32 * ((s - 112) / 7); }
    const char* w = (const char*)p.w[b][conv];
    v = *reinterpret_cast<const uint4*>(w + ((size_t)(r0 + (lane & 15)) * p.kpad[conv] + k0 + 8 * (lane >> 4)) * 2);
  }
  out[frag * 64 + lane] = v;
}

size_t trunk17_stream_bytes(int nblocks) { return (size_t)8 * ((size_t)nblocks * T17_FRAGS + RING) * 1024; }

hipError_t trunk17_repack(const Trunk17Pack& p, void* out, hipStream_t s) {
  const size_t frags = (size_t)8 * ((size_t)p.nblocks * T17_FRAGS + RING);
  hipLaunchKernelGGL(trunk17_repack_kernel, dim3((unsigned)((frags + 3) / 4)), dim3(256), 0, s, p, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_trunk17(const Trunk17Args& a, int dtype, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)block17_trunk_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, T17_LDS);
    (void)hipFuncSetAttribute((const void*)block17_trunk_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, T17_LDS);
    (void)hipGetLastError();
    attr_done = true;
  }
  if (dtype == BF16)
    hipLaunchKernelGGL(block17_trunk_kernel<__bf16>, dim3(a.n), dim3(512), T17_LDS, s, a);
  else if (dtype == F16)
    hipLaunchKernelGGL(block17_trunk_kernel<_Float16>, dim3(a.n), dim3(512), T17_LDS, s, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace vnf
