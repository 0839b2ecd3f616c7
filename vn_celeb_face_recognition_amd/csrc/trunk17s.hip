// Persistent Block17 stack ("repeat_2": 10 x Block17, /root/reference/models/inception_resnet_v1.py:70-95,226-237) in
// PLANAR SPLIT-F16 (the encoders' f16x2 dtype, split_f16.h): the in-gate (<= 1e-4) twin of trunk17.hip.
//
// Same shape as the 16-bit kernel -- one workgroup (8 waves) per 8x8x896 image, resident for the whole stack; the
// residual trunk in fp32 REGISTERS (112 VGPRs per wave) for all ten blocks; weights streamed per wave in fragment
// order with a register ring -- but every operand is an (hi, lo) pair of f16 planes and every product three MFMAs
//     W_hi . X_hi  +  W_hi . X_lo  +  W_lo . X_hi           (v_mfma_f32_16x16x32_f16, fp32 accumulate)
// so the activations cost twice the LDS of the 16-bit kernel and no longer fit at once.  A "tile image" here is
// [64 pixel rows][128 B] = 32 channels: per row the hi planes of the four 8-channel units (64 B), then their lo planes
// (16-byte slots XOR-swizzled by row & 7, as everywhere): one 32-deep k-step reads slot fgrp (hi) and 4 + fgrp (lo).
// The 160 KiB of LDS hold exactly 20 images; per block:
//
//   write X(ch 0..639)  -> images 0..19 |B| reduce k-steps 0..19 |B| write X(ch 640..895) -> 0..7, zero page -> 16 |B|
//   reduce k-steps 20..27, ReLU -> branch0 half of the concat (8..11) and branch1.0 (12..15)               |B|
//   (1,7): 12..15 -> 0..3  |B|  (7,1): 0..3 -> 4..7 (other half of the concat)  |B|  up: 8..11, 4..7 -> trunk     |B|
//
// (7 workgroup barriers per block; the zero page for out-of-image taps is rewritten in every block because the first
// pass needs all 20 images).  Weight stream: 336 fragments of 1 KiB per wave per block (hi and lo fragment of each
// (channel tile, k-step)), 26.9 MB per workgroup for the ten blocks.
//
// Numerics: fp32-class (~22 significant bits per operand); the trunk is never rounded below fp32 between blocks.
#include <type_traits>

#include "conv_device.h"
#include "trunk17.h"

namespace vnf {

namespace {

constexpr int S17_PX = 64;
constexpr int IMG = 64 * 128;                      // one tile image: 32 channels of the 64 pixels, hi | lo
constexpr int OFF_X1 = 0;                          // pass 1: channels 0..639, 20 images
constexpr int OFF_X2 = 0;                          // pass 2: channels 640..895, 8 images
constexpr int OFF_TB = 0;                          // 4 images: (1,7) output
constexpr int OFF_CATHI = 4 * IMG;                 // 4 images: concat channels 128..255 ((7,1) output)
constexpr int OFF_CATLO = 8 * IMG;                 // 4 images: concat channels 0..127 (branch0)
constexpr int OFF_TA = 12 * IMG;                   // 4 images: branch1.0 output
constexpr int OFF_ZERO = 16 * IMG;                 // 256 zero bytes: out-of-image taps (dead space after pass 1)
constexpr int S17_LDS = 20 * IMG;
constexpr int S17_FRAGS = 2 * T17_FRAGS;           // 336: 112 reduce + 56 (1x7) + 56 (7x1) + 112 up
#ifndef T17S_RING
#define T17S_RING 8
#endif
#ifndef T17S_P1
#define T17S_P1 0
#endif
constexpr int RING = T17S_RING;                    // fragments in flight per wave: (hi, lo) pairs
static_assert(S17_FRAGS % RING == 0, "the ring index of a step must not depend on the block");

__device__ __forceinline__ void split4(const f32x4_t& v, uint2& hi, uint2& lo) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const sf16 s(v[e]);
    h[e] = s.hi; l[e] = s.lo;
  }
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}
__device__ __forceinline__ f32x4_t join4(const uint2& hi, const uint2& lo) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 h = __builtin_bit_cast(h4, hi), l = __builtin_bit_cast(h4, lo);
  return f32x4_t{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
}

}  // namespace

__global__ __launch_bounds__(512, 2) void block17_trunk_split_kernel(const Trunk17Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow_c = lane & 15, fgrp_c = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const char* const xsrc = (const char*)a.x + (size_t)img * S17_PX * a.ldx * 4;

  // image -> LDS by LDS-DMA: piece = 8 pixel rows x 128 B of one tile image; logical slot q of a row = hi (q < 4) or lo
  // plane of 8-channel unit q & 3 of the image's 32 channels
  auto dma_in = [&](int t0, int nimg, int lds_off) {
    const int r = lane >> 3, slot = lane & 7;
    for (int piece = wave; piece < nimg * 8; piece += 8) {
      const int t = piece >> 3, p = (piece & 7) * 8 + r;
      const int q = slot ^ (p & 7);
      glds16(xsrc + (size_t)p * a.ldx * 4 + ((t0 + t) * 4 + (q & 3)) * 32 + (q >> 2) * 16,
             __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(lds_off + t * IMG + (piece & 7) * 1024)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // Per-lane LDS bases.  A wave's accumulator tiles cover channels 128*j + 16*wave + 4*fgrp + (0..3) of pixel 16*i + frow:
  // inside the [image][64 rows][128 B] layout that is wbase + 4*j images + i * 2048 for the hi halves (8 bytes), the lo
  // halves 64 bytes further (slot ^ 4).  B-fragment reads of pixel 16*i + frow, k-step (= image) s: s * IMG + i * 2048 +
  // rb0 (hi) / rb1 (lo).
  const int cw = 16 * wave + 4 * fgrp_c;  // channel offset inside a 128-channel group
  const int wbase_c = (cw >> 5) * IMG + frow_c * 128 + ((((cw & 31) >> 3) ^ (frow_c & 7)) << 4) + (cw & 4) * 2;
  const int rb0_c = frow_c * 128 + ((fgrp_c ^ (frow_c & 7)) << 4), rb1_c = rb0_c ^ 64;
  // every other term of an address below is a multiple of 128, so the lo plane's address is (base ^ 64) + the same terms:
  // the compiler folds those terms into the instructions' immediate offsets instead of keeping one register per address
  const int wlo_c = wbase_c ^ 64;

  // ---- the image -> trunk registers, through LDS in the two passes (whole 128-byte lines from memory)
  f32x4_t trunk[7][4];
  dma_in(0, 20, OFF_X1);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = OFF_X1 + j * 4 * IMG + i * 2048;
      trunk[j][i] = join4(*reinterpret_cast<const uint2*>(smem + wbase_c + c), *reinterpret_cast<const uint2*>(smem + wlo_c + c));
    }
  __syncthreads();
  dma_in(20, 8, OFF_X2);
  __syncthreads();
#pragma unroll
  for (int j = 5; j < 7; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = OFF_X2 + (j - 5) * 4 * IMG + i * 2048;
      trunk[j][i] = join4(*reinterpret_cast<const uint2*>(smem + wbase_c + c), *reinterpret_cast<const uint2*>(smem + wlo_c + c));
    }
  __syncthreads();

  // ---- weight stream of this wave: fragment f of block b at (b * S17_FRAGS + f) * 64 + lane (uint4 units)
  const uint4* wp = reinterpret_cast<const uint4*>(a.wstream) + (size_t)wave * ((size_t)a.nblocks * S17_FRAGS + RING) * 64 + lane;
  uint4 wq[RING];
#pragma unroll
  for (int r = 0; r < RING; ++r) wq[r] = wp[r * 64];
  wp += RING * 64;  // next fragment to fetch

  for (int b = 0; b < a.nblocks; ++b) {
    const float* bias = a.bias + (size_t)b * T17_BIAS + cw;
    // keep the (block-invariant) LDS addresses from being hoisted into registers that do not exist (trunk17.hip)
    int wbase = wbase_c, wlo = wlo_c, rb0 = rb0_c, rb1 = rb1_c, frow = frow_c, fgrp = fgrp_c;
    asm volatile("" : "+v"(wbase), "+v"(wlo), "+v"(rb0), "+v"(rb1), "+v"(frow), "+v"(fgrp));
    // ------------------------------------------------------------ pass 1 image: trunk channels 0..639 as (hi, lo) planes
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 hi, lo;
        split4(trunk[j][i], hi, lo);
        const int c = OFF_X1 + j * 4 * IMG + i * 2048;
        *reinterpret_cast<uint2*>(smem + wbase + c) = hi;
        *reinterpret_cast<uint2*>(smem + wlo + c) = lo;
        if (i & 1) __builtin_amdgcn_sched_barrier(0);   // split two tiles at a time: the halves are temporaries
      }
    __syncthreads();
    // ------------------------------------------------------------ reduce 1x1, 896 -> 256 (tiles w, w+8 per wave), two passes
    f32x4_t acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 128 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = bv;
    }
    // One k-step = 4 weight fragments ((hi, lo) of tiles w and w+8) against the four pixel tiles, pixel tile by pixel tile:
    // only two pixel tiles' fragments are live at a time (this phase holds 32 accumulators on top of the trunk).
    auto reduce_steps = [&](auto KS0, auto KS1, int base) {
#if T17S_P1 == 1
#pragma unroll
      for (int ks = decltype(KS0)::value; ks < decltype(KS1)::value; ++ks) {
        uint4 xh[4], xl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[i] = *reinterpret_cast<const uint4*>(smem + base + (ks - decltype(KS0)::value) * IMG + i * 2048 + rb0);
          xl[i] = *reinterpret_cast<const uint4*>(smem + base + (ks - decltype(KS0)::value) * IMG + i * 2048 + rb1);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int s = (ks * 2 + j) * 2;   // fragment pair (hi, lo) of (k-step, tile j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = mfma_f16(wq[s % RING], xh[i], acc[j][i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = mfma_f16(wq[s % RING], xl[i], acc[j][i]);
          wq[s % RING] = *wp;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = mfma_f16(wq[(s + 1) % RING], xh[i], acc[j][i]);
          wq[(s + 1) % RING] = wp[64];
          wp += 128;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
#endif
#pragma unroll
      for (int ks = decltype(KS0)::value; ks < decltype(KS1)::value; ++ks) {
        const int xo = base + (ks - decltype(KS0)::value) * IMG;
        const int s = ks * 4;   // fragments s .. s+3: tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo
        uint4 xh[2], xl[2];
        xh[0] = *reinterpret_cast<const uint4*>(smem + xo + rb0);
        xl[0] = *reinterpret_cast<const uint4*>(smem + xo + rb1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < 3) {
            xh[(i + 1) & 1] = *reinterpret_cast<const uint4*>(smem + xo + (i + 1) * 2048 + rb0);
            xl[(i + 1) & 1] = *reinterpret_cast<const uint4*>(smem + xo + (i + 1) * 2048 + rb1);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[j][i] = mfma_f16(wq[(s + 2 * j) % RING], xh[i & 1], acc[j][i]);
            acc[j][i] = mfma_f16(wq[(s + 2 * j) % RING], xl[i & 1], acc[j][i]);
            acc[j][i] = mfma_f16(wq[(s + 2 * j + 1) % RING], xh[i & 1], acc[j][i]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) wq[(s + r) % RING] = wp[r * 64];
        wp += 256;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#ifndef T17S_NO_P1
    reduce_steps(std::integral_constant<int, 0>{}, std::integral_constant<int, 20>{}, OFF_X1);
#endif
    __syncthreads();   // pass 1 read by every wave: its images may be overwritten
    // ------------------------------------------------------------ pass 2 image: channels 640..895 (+ the zero page)
    // (the asm keeps the compiler from splitting these eight tiles before pass 1 and carrying 32 more registers through it)
    asm volatile("" : "+v"(trunk[5][0]), "+v"(trunk[5][1]), "+v"(trunk[5][2]), "+v"(trunk[5][3]), "+v"(trunk[6][0]), "+v"(trunk[6][1]),
                 "+v"(trunk[6][2]), "+v"(trunk[6][3]));
#pragma unroll
    for (int j = 5; j < 7; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 hi, lo;
        split4(trunk[j][i], hi, lo);
        const int c = OFF_X2 + (j - 5) * 4 * IMG + i * 2048;
        *reinterpret_cast<uint2*>(smem + wbase + c) = hi;
        *reinterpret_cast<uint2*>(smem + wlo + c) = lo;
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
    if (tid < 16) reinterpret_cast<uint4*>(smem + OFF_ZERO)[tid] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
#ifndef T17S_NO_P1
    reduce_steps(std::integral_constant<int, 20>{}, std::integral_constant<int, 28>{}, OFF_X2);
#endif
    // ReLU -> (hi, lo) -> branch0 half of the concat (channels 0..127, j = 0) / branch1.0 output (j = 1): images 8..15,
    // which nobody has read since the barrier after pass 1
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4_t v = acc[j][i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        uint2 hi, lo;
        split4(v, hi, lo);
        const int c = (j == 0 ? OFF_CATLO : OFF_TA) + i * 2048;
        *reinterpret_cast<uint2*>(smem + wbase + c) = hi;
        *reinterpret_cast<uint2*>(smem + wlo + c) = lo;
      }
    __syncthreads();
    // ------------------------------------------------------------ (1,7) then (7,1), 128 -> 128 (tile w per wave)
#ifndef T17S_NO_P23
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int src_off = ph == 0 ? OFF_TA : OFF_TB;
      const int dst_off = ph == 0 ? OFF_TB : OFF_CATHI;
      f32x4_t pacc[4];
      {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 256 + 128 * ph);
#pragma unroll
        for (int i = 0; i < 4; ++i) pacc[i] = bv;
      }
#pragma unroll
      for (int ks = 0; ks < 28; ++ks) {
        const int tap = ks >> 2, t = ks & 3;        // k = tap * 128 + channel; image t of the 128-channel source
        // (addresses are re-derived per k-step from values the compiler cannot see through: hoisted, the 56 of a phase
        // would be carried in registers next to the trunk)
        asm volatile("" : "+v"(frow), "+v"(fgrp), "+v"(rb0), "+v"(rb1));
        // per tap one hi and one lo base; the image, the pixel tile and (7,1)'s row shift are immediate offsets
        uint4 xh[4], xl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = src_off + t * IMG + i * 2048;
          if (ph == 0) {   // pixel (y, x + tap - 3): the row's swizzle changes with x
            const int xx = (frow & 7) + tap - 3;
            const bool ok = (unsigned)xx < 8u;
            const int bh = (frow + tap - 3) * 128 + ((fgrp ^ (xx & 7)) << 4), bl = bh ^ 64;
            xh[i] = *reinterpret_cast<const uint4*>(smem + (ok ? bh + c : OFF_ZERO));
            xl[i] = *reinterpret_cast<const uint4*>(smem + (ok ? bl + c : OFF_ZERO));
          } else {         // pixel (y + tap - 3, x): same swizzle, rows 8 apart
            const int yy = 2 * i + (frow >> 3) + tap - 3;
            const bool ok = (unsigned)yy < 8u;
            xh[i] = *reinterpret_cast<const uint4*>(smem + (ok ? rb0 + c + (tap - 3) * 1024 : OFF_ZERO));
            xl[i] = *reinterpret_cast<const uint4*>(smem + (ok ? rb1 + c + (tap - 3) * 1024 : OFF_ZERO));
          }
        }
        const int s = 112 + 56 * ph + 2 * ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) pacc[i] = mfma_f16(wq[s % RING], xh[i], pacc[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pacc[i] = mfma_f16(wq[s % RING], xl[i], pacc[i]);
        wq[s % RING] = *wp;
#pragma unroll
        for (int i = 0; i < 4; ++i) pacc[i] = mfma_f16(wq[(s + 1) % RING], xh[i], pacc[i]);
        wq[(s + 1) % RING] = wp[64];
        wp += 128;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4_t v = pacc[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        uint2 hi, lo;
        split4(v, hi, lo);
        const int c = dst_off + i * 2048;
        *reinterpret_cast<uint2*>(smem + wbase + c) = hi;
        *reinterpret_cast<uint2*>(smem + wlo + c) = lo;
      }
      __syncthreads();
    }
#endif
    // ------------------------------------------------------------ up 1x1, 256 -> 896, accumulated into the trunk
#ifndef T17S_NO_P4
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int base = (ks < 4 ? OFF_CATLO : OFF_CATHI) + (ks & 3) * IMG;
      uint4 xh[4], xl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[i] = *reinterpret_cast<const uint4*>(smem + base + i * 2048 + rb0);
        xl[i] = *reinterpret_cast<const uint4*>(smem + base + i * 2048 + rb1);
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int s = 224 + (ks * 7 + j) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) trunk[j][i] = mfma_f16(wq[s % RING], xh[i], trunk[j][i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) trunk[j][i] = mfma_f16(wq[s % RING], xl[i], trunk[j][i]);
        wq[s % RING] = *wp;
#pragma unroll
        for (int i = 0; i < 4; ++i) trunk[j][i] = mfma_f16(wq[(s + 1) % RING], xh[i], trunk[j][i]);
        wq[(s + 1) % RING] = wp[64];
        wp += 128;
        __builtin_amdgcn_sched_barrier(0);   // the refill loads stay behind the tile that frees their registers
      }
    }
#endif
    // relu(x + conv*scale + bias*scale): scale is folded into the weights and the bias
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 512 + 128 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) trunk[j][i][e] = fmaxf(trunk[j][i][e] + bv[e], 0.f);
    }
    __syncthreads();  // every wave is done reading the concat: the next block's pass 1 may overwrite it
  }

  // ---- trunk -> (hi, lo) planes -> LDS images -> whole NHWC rows of the output tensor, in the two passes
  char* const dst = (char*)a.y + (size_t)img * S17_PX * a.ldy * 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int j0 = pass == 0 ? 0 : 5, j1 = pass == 0 ? 5 : 7;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j >= j0 && j < j1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint2 hi, lo;
          split4(trunk[j][i], hi, lo);
          const int c = (j - j0) * 4 * IMG + i * 2048;
          *reinterpret_cast<uint2*>(smem + wbase_c + c) = hi;
          *reinterpret_cast<uint2*>(smem + wlo_c + c) = lo;
        }
      }
    __syncthreads();
    const int units = (j1 - j0) * 16;                // 8-channel units per pixel in this pass
    const int chunks = S17_PX * units * 2;           // 16-byte chunks: [pixel][unit][plane]
    for (int idx = tid; idx < chunks; idx += 512) {
      const int p = idx / (units * 2), c = idx - p * (units * 2);
      const int u = c >> 1, plane = c & 1;
      const int q = (u & 3) + 4 * plane;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + (u >> 2) * IMG + p * 128 + ((q ^ (p & 7)) << 4));
      *reinterpret_cast<uint4*>(dst + (size_t)p * a.ldy * 4 + (j0 * 16 + u) * 32 + plane * 16) = v;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- weight stream
// Fragment order of one wave within one block (must match the kernel's consumption order); every (tile, k-step) is a
// pair: the hi-plane fragment, then the lo-plane fragment
//   reduce : for ks in 0..27, j in 0..1   rows 128j+16w .. +15,  k = 32ks .. +31      (112)
//   1x7    : for ks in 0..27              rows 16w .. +15,       k = 32ks .. +31      (56)
//   7x1    : for ks in 0..27              rows 16w .. +15,       k = 32ks .. +31      (56)
//   up     : for ks in 0..7, j in 0..6    rows 128j+16w .. +15,  k = 32ks .. +31      (112)
// Lane l of a fragment holds the 8 k values k0 + 8*(l>>4) .. +7 of row r0 + (l&15) (the MFMA A-operand map); the packed
// engine weights keep a K tile of 32 k values as [32 hi halves][32 lo halves] (engine.cpp convert_to F16P).
__global__ void trunk17s_repack_kernel(Trunk17Pack p, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t frag = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t per_wave = (size_t)p.nblocks * S17_FRAGS + RING;
  if (frag >= 8 * per_wave) return;
  const int wave = (int)(frag / per_wave);
  const size_t f = frag - (size_t)wave * per_wave;
  uint4 v = {0u, 0u, 0u, 0u};
  if (f < (size_t)p.nblocks * S17_FRAGS) {
    const int b = (int)(f / S17_FRAGS), sp = (int)(f - (size_t)b * S17_FRAGS);
    const int s = sp >> 1, plane = sp & 1;
    int conv, r0, k0;
    if (s < 56) { conv = 0; r0 = 128 * (s & 1) + 16 * wave; k0 = 32 * (s >> 1); }
    else if (s < 84) { conv = 1; r0 = 16 * wave; k0 = 32 * (s - 56); }
    else if (s < 112) { conv = 2; r0 = 16 * wave; k0 = 32 * (s - 84); }
    else { conv = 3; r0 = 128 * ((s - 112) % 7) + 16 * wave; k0 = 32 * ((s - 112) / 7); }
    const char* w = (const char*)p.w[b][conv];
    v = *reinterpret_cast<const uint4*>(w + ((size_t)(r0 + (lane & 15)) * p.kpad[conv] + k0) * 4 + plane * 64 + (lane >> 4) * 16);
  }
  out[frag * 64 + lane] = v;
}

size_t trunk17s_stream_bytes(int nblocks) { return (size_t)8 * ((size_t)nblocks * S17_FRAGS + RING) * 1024; }

hipError_t trunk17s_repack(const Trunk17Pack& p, void* out, hipStream_t s) {
  const size_t frags = (size_t)8 * ((size_t)p.nblocks * S17_FRAGS + RING);
  hipLaunchKernelGGL(trunk17s_repack_kernel, dim3((unsigned)((frags + 3) / 4)), dim3(256), 0, s, p, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_trunk17s(const Trunk17Args& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  static const hipError_t attr = hipFuncSetAttribute((const void*)block17_trunk_split_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, S17_LDS);
  (void)attr;
  (void)hipGetLastError();
  hipLaunchKernelGGL(block17_trunk_split_kernel, dim3(a.n), dim3(512), S17_LDS, s, a);
  return hipGetLastError();
}

}  // namespace vnf
