// repeat_1 -- the five Block35 of /root/reference/models/inception_resnet_v1.py:36-67, 226-232 -- as ONE launch, one
// workgroup (8 waves) per 17x17x256 image, with the residual stream x resident in REGISTERS across the five blocks.
//
// The one-launch-per-block kernel (block35.hip) moves x through the CU three times per block (reduce operand by LDS-DMA,
// residual re-read, y store: 444 KB per image and block) and runs its phases one after the other, so a block takes
// ~65 k cycles for ~11 k cycles of MFMA work.  Here x is loaded once and stored once:
//
//   * x lives as T-rounded values in the MFMA accumulator layout: lane (frow, fgrp) of the wave that owns pixel tile i
//     holds channels 16j + 4 fgrp .. +3 of pixel 16 i + frow for every 16-channel tile j -- 16 x 2 registers per pixel
//     tile, 96 registers for a wave's three tiles.  (The plan's activations are T-rounded between layers anyway, so this
//     IS the tensor the five-launch plan writes to memory.)
//   * a 32-channel k-step of such a tile becomes an MFMA B fragment (lane holds channels 8 fgrp .. +7) with four
//     cross-row swaps (v_permlane32_swap + v_permlane16_swap on both registers), no LDS: the reduce 1x1 reads x from
//     registers, and the three branch outputs b0 / b1 / b2 -- each produced and consumed by the same wave for the same
//     pixels -- go from accumulators straight into the up-projection's fragments.  Only the two 3x3 sources (t1 -> t2b,
//     t2) are pixel images in LDS, because a 3x3 tap reads other waves' pixels.
//   * the up-projection's epilogue is (acc + bias) + x -> ReLU -> T, written back into the x registers.
// Weights: block35_repack's per-block image (MFMA A-fragment order) by linear LDS-DMA, every transfer issued a phase or
// more before its first use.  Same K order, operand roles, fp32 sums and roundings as block35.hip and the unfused plan:
// bit-identical results.
//
// LDS (136 KiB):
//   [0, 49152)         W1 (reduce, 48 fragments) during phase A -- prefetched under the previous block's phase E; from
//                      the end of phase A its first 37888 bytes are the two pixel images ([296 rows][64 B], 16-byte
//                      chunks XOR-swizzled by ((-(row>>2)) & 3)): image 0 = t1, then t2b; image 1 = t2
//   [49152, 98304)     W5 (up, 48 fragments), loaded under phases A-D
//   [98304, 135168)    two 3x3 weight buffers (18 fragments each): W2 | W3 prefetched under phase E, W4 under phase C
//   [135168, 139264)   the blocks' 448 fp32 biases, double-buffered by block parity
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "block35.h"
#include "conv_device.h"

namespace vnf {

namespace {

constexpr int NPX = 289, IMW = 17;
constexpr int ROWS = 296;            // image rows kept in LDS (289 pixels, padding rows zero)
constexpr int ZROW = 295;            // an always-zero padding row: source of out-of-image taps
constexpr int IMG_BYTES = ROWS * 64; // 18944
constexpr int OFF_W1 = 0;
constexpr int OFF_W5 = 48 * 1024;
constexpr int OFF_W33 = 96 * 1024;
constexpr int W33_BYTES = 18 * 1024;
constexpr int OFF_BIAS = OFF_W33 + 2 * W33_BYTES;  // 135168
constexpr int T35_LDS = OFF_BIAS + 2 * 2048;       // 139264
static_assert(2 * IMG_BYTES <= 48 * 1024, "the two images live inside the W1 region");

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
  const t4 r = __builtin_bit_cast(t4, u);
  return f32x4_t{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
}

// Two neighbouring 16-channel tiles of one pixel tile in accumulator layout (lane row q holds channel quad q of each)
// -> the MFMA B fragment of that 32-channel k-step (lane row r holds channels 8r .. 8r+7):
//   permlane32_swap: lane rows 2,3 of t0 <-> rows 0,1 of t1;  permlane16_swap: odd rows of t0 <-> even rows of t1
__device__ __forceinline__ uint4 quads_to_frag(const uint2& t0, const uint2& t1) {
  const auto ax = __builtin_amdgcn_permlane32_swap(t0.x, t1.x, false, false);
  const auto ay = __builtin_amdgcn_permlane32_swap(t0.y, t1.y, false, false);
  const auto bx = __builtin_amdgcn_permlane16_swap(ax[0], ax[1], false, false);
  const auto by = __builtin_amdgcn_permlane16_swap(ay[0], ay[1], false, false);
  return uint4{bx[0], by[0], bx[1], by[1]};
}

__device__ __forceinline__ int img_chunk(int q, int chunk) { return q * 64 + ((chunk ^ ((0 - (q >> 2)) & 3)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace

// NT = pixel tiles of the wave (wave, wave + 8, wave + 16: three for waves 0-2, two for the rest) as a compile-time
// constant: with a run-time tile count around the third tile its registers are conditionally defined everywhere and the
// allocator spills ~800 bytes
template <typename T, int NT, bool DBG>
__device__ __forceinline__ void block35_stack_body(const Block35StackArgs& a, char* smem, const int wave) {
  const int tid = threadIdx.x, lane = tid & 63;
  // instrumented build (VNF_T35_STAMP): cycles per phase segment summed over the blocks, in scalar registers (the
  // segment ids follow the stamp() calls in program order: 0 = prologue, 1 = first wait, then per block 12 segments)
  long long tsum[16];
  long long tlast = 0;
  int nstamp = 0;
  if constexpr (DBG) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tsum[i] = 0;
    tlast = __builtin_readcyclecounter();
  }
  auto stamp = [&](int seg) {
    if constexpr (DBG) {
      const long long t = __builtin_readcyclecounter();
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i == seg) tsum[i] += t - tlast;
      tlast = t;
    }
  };
  const int frow = lane & 15, fgrp = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const char* __restrict__ xg = (const char*)a.x + (size_t)img * NPX * a.ldx * 2;
  char* __restrict__ yg = (char*)a.y + (size_t)img * NPX * a.ldy * 2;

  // linear LDS-DMA copy of `total` 1-KiB pieces, CNT per wave (ids past the end repeat the last piece, so every wave
  // issues the same number of DMAs and the counted waits are wave-independent)
  auto copy_lin = [&](const char* src, int lds_off, int total, auto cnt_tag) {
    constexpr int CNT = decltype(cnt_tag)::value;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const int id = min(wave + 8 * i, total - 1);
      glds16(src + (size_t)id * 1024 + lane * 16, lds0 + lds_off + id * 1024);
    }
  };
  using C1 = std::integral_constant<int, 1>;
  using C3 = std::integral_constant<int, 3>;
  using C5 = std::integral_constant<int, 5>;
  using C6 = std::integral_constant<int, 6>;
  auto prefetch_front = [&](int b) {  // bias, W1, W2 | W3 of block b: 12 DMAs per wave
    const char* wi = (const char*)a.wimg + (size_t)b * B35_WIMG_BYTES;
    copy_lin(wi + B35_BIASOFF, OFF_BIAS + (b & 1) * 2048, 2, C1{});
    copy_lin(wi + B35_W1, OFF_W1, 48, C6{});
    copy_lin(wi + B35_W2, OFF_W33, 36, C5{});
  };

  // ---- x -> registers (accumulator layout): 16-byte loads of 8 channels, then the lane-row swap that splits them into
  // the channel quads of two neighbouring 16-channel tiles (the inverse -- it is an involution -- of the y store below)
  uint2 xr[NT][16];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    {
      const int p = min(16 * (wave + 8 * i) + frow, NPX - 1);
#pragma unroll
      for (int jp = 0; jp < 8; ++jp) {
        const int c = (2 * jp + (fgrp & 1)) * 16 + (fgrp >> 1) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(xg + ((size_t)p * a.ldx + c) * 2);
        const auto sx = __builtin_amdgcn_permlane16_swap(v.x, v.z, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(v.y, v.w, false, false);
        xr[i][2 * jp] = uint2{sx[0], sy[0]};
        xr[i][2 * jp + 1] = uint2{sx[1], sy[1]};
      }
    }
  prefetch_front(0);
  copy_lin((const char*)a.wimg + B35_W5, OFF_W5, 48, C6{});
  stamp(0);
  wait_vm<0>();
  __syncthreads();
  stamp(1);

  const int lane_k = lane;
#pragma unroll 1
  for (int b = 0; b < a.nblocks; ++b) {
    // Every LDS address below is invariant over the blocks, and the compiler would hoist a few hundred of them out of
    // this loop into registers it does not have (they spill).  Re-deriving them from a value it cannot see through
    // keeps them where they are used.
    int lane = lane_k;
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fgrp = lane >> 4;
    const char* wi = (const char*)a.wimg + (size_t)b * B35_WIMG_BYTES;
    const float* bias = reinterpret_cast<const float*>(smem + OFF_BIAS + (b & 1) * 2048);
    uint4 cf[3][NT];  // concat fragments of the up projection: [k-step = b0, b1, b2][pixel tile]

    // ================================================================= phase A: reduce 1x1, 256 -> 96, x from registers
    // two passes of 48 output channels: 36 accumulator registers beside the 96 of x
    {
      uint2 qa[NT][6];  // the reduce outputs (bias, ReLU, T-rounded) per pixel tile and 16-channel tile
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        f32x4_t acc[3][NT];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // software-pipelined by hand: the weight fragments and the lane-row swaps of k-step ks+1 are issued before the
        // MFMAs of k-step ks (with two waves per SIMD nothing else hides the LDS latency and the swaps' VALU time)
        uint4 xf[2][NT], wf[2][3];
        auto fetch = [&](auto KS, auto C) {
          constexpr int ks = decltype(KS)::value, c = decltype(C)::value;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            wf[c][j] = *reinterpret_cast<const uint4*>(smem + OFF_W1 + (ks * 6 + 3 * jh + j) * 1024 + lane * 16);
#pragma unroll
          for (int i = 0; i < NT; ++i) xf[c][i] = quads_to_frag(xr[i][2 * ks], xr[i][2 * ks + 1]);
        };
        auto step = [&](auto KS) {
          constexpr int ks = decltype(KS)::value, c = ks & 1;
          if constexpr (ks + 1 < 8) fetch(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, c ^ 1>{});
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[j][i] = Mma<T>::run(wf[c][j], xf[c][i], acc[j][i]);
          __builtin_amdgcn_sched_barrier(0);
        };
        fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
#pragma unroll
        for (int i = 0; i < NT; ++i)
          {
            const int p = 16 * (wave + 8 * i) + frow;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 16 * (3 * jh + j) + 4 * fgrp);
              f32x4_t v = acc[j][i];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = p < NPX ? fmaxf(v[e] + bv[e], 0.f) : 0.f;
              qa[i][3 * jh + j] = pack4<T>(v);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      stamp(2);
      __syncthreads();  // W1 is dead: the images (which overlay it) may be written
      stamp(3);
      // b0 (channel tiles 0,1) stays in registers as the first concat fragment; t1 (2,3) -> image 0, t2 (4,5) -> image 1
#pragma unroll
      for (int i = 0; i < NT; ++i)
        {
          const int p = 16 * (wave + 8 * i) + frow;
          cf[0][i] = quads_to_frag(qa[i][0], qa[i][1]);
          if (p < ROWS) {
#pragma unroll
            for (int j = 2; j < 6; ++j)
              *reinterpret_cast<uint2*>(smem + ((j >> 1) - 1) * IMG_BYTES + img_chunk(p, 2 * (j & 1) + (fgrp >> 1)) + (fgrp & 1) * 8) = qa[i][j];
          }
        }
      stamp(4);
      __syncthreads();
      stamp(5);
    }

    // pixel coordinates of this lane's rows (3x3 taps); derived here, not above phase A, whose registers are all taken
    int py[NT], px[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int p = 16 * (wave + 8 * i) + frow;
      py[i] = p < NPX ? p / IMW : -100;
      px[i] = p - (p / IMW) * IMW;
    }
    // ================================================================= phases B, C, D: 3x3 pad 1, 32 -> 32
    //   B: image 0 (t1) -> b1 (registers), W2 | C: image 1 (t2) -> image 0 (t2b), W3 | D: image 0 -> b2 (registers), W4
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
      const int src = (ph == 1 ? 1 : 0) * IMG_BYTES;
      const int wb = OFF_W33 + (ph == 1 ? W33_BYTES : 0);
      f32x4_t acc[2][NT];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        uint4 xf[NT], wf[2];
#pragma unroll
        for (int i = 0; i < NT; ++i)
          {
            const int q = 16 * (wave + 8 * i) + frow + dy * IMW + dx;
            const bool ok = (unsigned)(py[i] + dy) < (unsigned)IMW && (unsigned)(px[i] + dx) < (unsigned)IMW;
            xf[i] = *reinterpret_cast<const uint4*>(smem + src + (ok ? img_chunk(q, fgrp) : ZROW * 64));
          }
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const uint4*>(smem + wb + (tap * 2 + j) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i)
            acc[j][i] = Mma<T>::run(wf[j], xf[i], acc[j][i]);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
        {
          const int p = 16 * (wave + 8 * i) + frow;
          uint2 q[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 96 + 32 * ph + 16 * j + 4 * fgrp);
            f32x4_t v = acc[j][i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = p < NPX ? fmaxf(v[e] + bv[e], 0.f) : 0.f;
            q[j] = pack4<T>(v);
          }
          if (ph == 1) {
            // t2b overwrites t1 (image 0): its last readers finished before phase B's closing barrier
            if (p < ROWS) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
                *reinterpret_cast<uint2*>(smem + img_chunk(p, 2 * j + (fgrp >> 1)) + (fgrp & 1) * 8) = q[j];
            }
          } else {
            cf[ph == 0 ? 1 : 2][i] = quads_to_frag(q[0], q[1]);
          }
        }
      stamp(6);
      if (ph == 1) wait_vm<0>();  // W4 (issued after B) and W5 have landed before the barrier that opens D / E
      __syncthreads();
      stamp(7);
      if (ph == 0) copy_lin(wi + B35_W4, OFF_W33, 18, C3{});  // W2 is dead
    }
    // images and both 3x3 buffers are dead: the next block's bias, W1, W2 | W3 travel under phase E
    // (the last block fetches its own again: no branch in the loop)
    prefetch_front(min(b + 1, a.nblocks - 1));

    // ================================================================= phase E: up 1x1, 96 -> 256, + x, ReLU -> x
    // eight groups of 32 output channels (24 accumulator registers beside x's 96 and the 36 of the concat fragments)
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4_t acc[2][NT];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[jj][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        uint4 wf[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) wf[jj] = *reinterpret_cast<const uint4*>(smem + OFF_W5 + (ks * 16 + 2 * g + jj) * 1024 + lane * 16);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < NT; ++i)
            acc[jj][i] = Mma<T>::run(wf[jj], cf[ks][i], acc[jj][i]);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
        {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            // (acc + bias) + x, in the unfused epilogue's order
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 192 + 32 * g + 16 * jj + 4 * fgrp);
            const f32x4_t xv = unpack4<T>(xr[i][2 * g + jj]);
            f32x4_t v = acc[jj][i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] + bv[e]) + xv[e], 0.f);
            xr[i][2 * g + jj] = pack4<T>(v);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(8);
    wait_vm<0>();     // the next block's bias, W1, W2 | W3 (this wave's pieces) have landed
    __syncthreads();  // ... everybody's; and W5 is dead
    stamp(9);
    copy_lin((const char*)a.wimg + (size_t)min(b + 1, a.nblocks - 1) * B35_WIMG_BYTES + B35_W5, OFF_W5, 48, C6{});
  }
  wait_vm<0>();  // the redundant last prefetch: no DMA may be in flight into a workgroup's LDS when it ends

  // ---- optional tail: mixed_6a.branch1.0 (1x1, 256 -> 192) on the stack's output, x still in registers: the reduce
  // phase's machinery once more (fragments by lane-row swaps, weights in LDS in fragment order), three passes of four
  // channel tiles, register epilogue with 16-byte stores.  Same K order, bias add, ReLU and rounding as the plan's
  // convolution kernel: bit-identical to it.
  if (a.wtail) {
    using C13 = std::integral_constant<int, 13>;
    __syncthreads();   // every wave's redundant last prefetch has landed (wait_vm<0> above): the regions may be refilled
    copy_lin((const char*)a.wtail, 0, 97, C13{});   // 96 fragments over [0, 96 KiB), the biases behind them
    wait_vm<0>();
    __syncthreads();
    int lane_t = lane_k;
    asm volatile("" : "+v"(lane_t));
    const int frow_t = lane_t & 15, fgrp_t = lane_t >> 4;
    const float* tbias = reinterpret_cast<const float*>(smem + 96 * 1024);
    char* __restrict__ yt = (char*)a.ytail + (size_t)img * NPX * a.ldyt * 2;
#pragma unroll
    for (int jh = 0; jh < 3; ++jh) {
      f32x4_t acc[4][NT];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // software-pipelined like phase A: k-step ks+1's fragments are fetched before k-step ks's MFMAs
      uint4 xf[2][NT], wf[2][4];
      auto fetch = [&](auto KS, auto C) {
        constexpr int ks = decltype(KS)::value, c = decltype(C)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[c][j] = *reinterpret_cast<const uint4*>(smem + (ks * 12 + 4 * jh + j) * 1024 + lane_t * 16);
#pragma unroll
        for (int i = 0; i < NT; ++i) xf[c][i] = quads_to_frag(xr[i][2 * ks], xr[i][2 * ks + 1]);
      };
      auto step = [&](auto KS) {
        constexpr int ks = decltype(KS)::value, c = ks & 1;
        if constexpr (ks + 1 < 8) fetch(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, c ^ 1>{});
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[j][i] = Mma<T>::run(wf[c][j], xf[c][i], acc[j][i]);
        __builtin_amdgcn_sched_barrier(0);
      };
      fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int t0 = 4 * jh + 2 * jp;   // channel tiles t0, t0 + 1 are exchanged between the lane rows
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(tbias + 16 * t0 + 4 * fgrp_t);
        const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(tbias + 16 * t0 + 16 + 4 * fgrp_t);
        const int c = (t0 + (fgrp_t & 1)) * 16 + (fgrp_t >> 1) * 8;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          f32x4_t v0 = acc[2 * jp][i], v1 = acc[2 * jp + 1][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e] + b0[e], 0.f); v1[e] = fmaxf(v1[e] + b1[e], 0.f); }
          const uint2 p0 = pack4<T>(v0), p1 = pack4<T>(v1);
          const auto sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
          const auto sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
          const int p = 16 * (wave + 8 * i) + frow_t;
          if (p < NPX) *reinterpret_cast<uint4*>(yt + ((size_t)p * a.ldyt + c) * 2) = uint4{sx[0], sy[0], sx[1], sy[1]};
        }
      }
    }
  }
  if constexpr (DBG) {
    if (a.dbg && blockIdx.x == 100 && (threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a.dbg[wave * 64 + i] = tsum[i];
    }
  }
  // ---- x registers -> y: the lane-row swap gives every lane 8 consecutive channels = one 16-byte store
  // (addresses from an opaque lane id again: computed up front they would sit in registers -- spilled -- all kernel long)
  int lane_y = lane_k;
  asm volatile("" : "+v"(lane_y));
  const int frow_y = lane_y & 15, fgrp_y = lane_y >> 4;
#pragma unroll
  for (int i = 0; i < NT; ++i)
    {
      const int p = 16 * (wave + 8 * i) + frow_y;
#pragma unroll
      for (int jp = 0; jp < 8; ++jp) {
        const int c = (2 * jp + (fgrp_y & 1)) * 16 + (fgrp_y >> 1) * 8;
        const uint2 p0 = xr[i][2 * jp], p1 = xr[i][2 * jp + 1];
        const auto sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
        if (p < NPX) *reinterpret_cast<uint4*>(yg + ((size_t)p * a.ldy + c) * 2) = uint4{sx[0], sy[0], sx[1], sy[1]};
      }
    }
}

template <typename T, bool DBG = false>
__global__ __launch_bounds__(512, 2) void block35_stack_kernel(const Block35StackArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 3)
    block35_stack_body<T, 3, DBG>(a, smem, wave);
  else
    block35_stack_body<T, 2, DBG>(a, smem, wave);
}

// Instrumented launch: VNF_T35_STAMP=<file> appends the stamps of workgroup 100 (bf16, n > 100): per wave
//   start | x loaded, weights issued | first barrier | per block: A done, barrier, images written, barrier, then for
//   B, C, D: phase done, barrier; E done, barrier
static hipError_t launch_stack_stamped(const Block35StackArgs& a, hipStream_t s) {
  static long long* dbuf = nullptr;
  const int n = 8 * 64;
  if (!dbuf && hipMalloc((void**)&dbuf, n * 8) != hipSuccess) return hipErrorOutOfMemory;
  (void)hipMemsetAsync(dbuf, 0, n * 8, s);
  Block35StackArgs aa = a;
  aa.dbg = dbuf;
  (void)hipFuncSetAttribute((const void*)block35_stack_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T35_LDS);
  hipLaunchKernelGGL((block35_stack_kernel<__bf16, true>), dim3(a.n), dim3(512), T35_LDS, s, aa);
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  static long long host[8 * 64];
  (void)hipMemcpy(host, dbuf, n * 8, hipMemcpyDeviceToHost);
  if (FILE* f = fopen(getenv("VNF_T35_STAMP"), "a")) {
    fprintf(f, "launch n=%d nblocks=%d\n", a.n, a.nblocks);
    for (int w = 0; w < 8; ++w) {
      fprintf(f, "%d", w);
      for (int i = 0; i < 10; ++i) fprintf(f, " %lld", host[w * 64 + i]);
      fprintf(f, "\n");
    }
    fclose(f);
  }
  return hipSuccess;
}

// mixed_6a.branch1.0's packed engine weights [192 rows][kpad] -> 96 MFMA A-fragments (f = ks * 12 + j: rows 16 j .. + 15,
// k = 32 ks + 8 (lane >> 4) .. + 7) and a last KiB with the 192 fp32 biases
__global__ void block35_tail_repack_kernel(const char* __restrict__ w, int kpad, const float* __restrict__ bias, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (f > 96) return;
  uint4 v = {0u, 0u, 0u, 0u};
  if (f == 96) {
    if (lane * 4 < 192) v = reinterpret_cast<const uint4*>(bias)[lane];
  } else {
    const int ks = f / 12, j = f % 12;
    v = *reinterpret_cast<const uint4*>(w + ((size_t)(16 * j + (lane & 15)) * kpad + 32 * ks + 8 * (lane >> 4)) * 2);
  }
  out[(size_t)f * 64 + lane] = v;
}

hipError_t block35_tail_repack(const void* w, int kpad, const float* bias, void* out, hipStream_t s) {
  hipLaunchKernelGGL(block35_tail_repack_kernel, dim3(25), dim3(256), 0, s, (const char*)w, kpad, bias, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_block35_stack(const Block35StackArgs& a, int dtype, hipStream_t s) {
  if (a.n <= 0 || a.nblocks <= 0) return hipSuccess;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)block35_stack_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, T35_LDS);
    (void)hipFuncSetAttribute((const void*)block35_stack_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, T35_LDS);
    (void)hipGetLastError();
    attr_done = true;
  }
  if (dtype == BF16 && a.n > 100 && getenv("VNF_T35_STAMP")) return launch_stack_stamped(a, s);
  if (dtype == BF16)
    hipLaunchKernelGGL(block35_stack_kernel<__bf16>, dim3(a.n), dim3(512), T35_LDS, s, a);
  else if (dtype == F16)
    hipLaunchKernelGGL(block35_stack_kernel<_Float16>, dim3(a.n), dim3(512), T35_LDS, s, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace vnf
