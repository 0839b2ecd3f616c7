// One Block35 (/root/reference/models/inception_resnet_v1.py:36-67) per launch in PLANAR SPLIT-F16 (the encoders' f16x2
// dtype, split_f16.h): the in-gate twin of block35.hip.  One workgroup (8 waves) per 17x17x256 image:
//
//   x -> reduce 1x1 256->(32|32|32) -> { b0 ; b1 = 3x3(t1) ; b2 = 3x3(3x3(t2)) } -> up 1x1 96->256 -> y = relu(x + up)
//
// Every operand is an (hi, lo) pair of f16 planes and every product three MFMAs (W_hi.X_hi + W_hi.X_lo + W_lo.X_hi), so
// activations and weights take twice the LDS of the 16-bit kernel.  What changes against block35.hip:
//   * an image / x K tile is [296 pixel rows][128 B] = 32 channels: per row the hi planes of the four 8-channel units,
//     then their lo planes (16-byte slots XOR-swizzled by row & 7: conflict-free for any 16 consecutive rows, which is
//     what a 3x3 tap reads);
//   * the reduce weights no longer fit beside the x ring: each ring stage carries its k-step of W1 (12 KiB) behind its
//     x tile (8 k-steps of 32 channels, two stages);
//   * the three branch outputs b0, b1, b2 never touch LDS: an accumulator quad is 4 channels of a pixel, two 16-channel
//     tiles side by side are the 8 k values a lane contributes to a 32-deep MFMA step -- in a k ORDER of our choosing,
//     as long as the weights use the same one (block35s_repack_kernel permutes W5's k accordingly).  So the up
//     convolution's B operand is built from registers; only t1, t2 and t2b live in LDS (two image slots);
//   * W5 (96 KiB) lands in the space the images and the 3x3 weights leave behind during phase D.
// Work split by PIXELS as in block35.hip: 19 tiles of 16 pixels, wave w owns tiles w, w+8, w+16.
//
// LDS map (bytes):  I0 [0, 37888)  I1 [37888, 75776)  WA [75776, 112640)  WB [112640, 149504)  bias [149504, 151552)
//   phase A: ring stage s at s * 50176 (x tile 37888 + W1 k-step 12288), W2 prefetched into WB
//   then  B: I0 (t1), WB (W2) -> b1 regs | C: I1 (t2), WA (W3) -> I0 (t2b) | D: I0, WB (W4) -> b2 regs, W5 k-steps 0,1 land
//   in [37888, 103424) | E: W5 k-step 2 -> [103424, 136192); per-wave fp32 staging of the epilogue in [0, 18432)
#include <type_traits>

#include "block35.h"
#include "conv_device.h"

namespace vnf {

namespace {

constexpr int NPX = 289, IMW = 17;
constexpr int ROWS = 296;                         // rows kept per image: 289 pixels rounded up to whole 8-row DMA pieces
constexpr int ZROW = 295;                         // a padding row, always zero: source of out-of-image taps
constexpr int IMGB = ROWS * 128;                  // 37888: one 32-channel image / one x K tile
constexpr int W1STEP = 12 * 1024;                 // one k-step of W1: 6 tiles x (hi, lo) fragments
constexpr int STAGE = IMGB + W1STEP;              // 50176
constexpr int OFF_I0 = 0, OFF_I1 = IMGB, OFF_WA = 2 * IMGB, OFF_WB = OFF_WA + 36 * 1024, OFF_BIAS = OFF_WB + 36 * 1024;
constexpr int OFF_W5A = IMGB;                     // W5 k-steps 0, 1 (64 KiB) during phase D
constexpr int OFF_W5B = OFF_W5A + 64 * 1024;      // W5 k-step 2 (32 KiB) after phase D
constexpr int S35_LDS = 160 * 1024;
static_assert(2 * STAGE <= OFF_WB && OFF_BIAS + 2048 <= S35_LDS && OFF_W5B + 32 * 1024 <= OFF_BIAS, "LDS map");
constexpr int STG_PITCH = 144;                    // fp32 staging row: 32 channels + 16 B pad
constexpr int STG_WAVE = 16 * STG_PITCH;          // 2304 B per wave

// weight image of one block (1-KiB fragments, every (tile, k-step) as a (hi, lo) pair):
//   W1: k-step major, 8 x 12 | W2, W3, W4: 36 each (tap major, 2 tiles) | W5: k-step major, 3 x 32 (16 tiles) | 2 KiB biases
constexpr int S35_W1 = 0, S35_W2 = 96, S35_W3 = 132, S35_W4 = 168, S35_W5 = 204, S35_FRAGS = 300;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void split4h(const f32x4_t& v, uint2& hi, uint2& lo) {
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

}  // namespace

__global__ __launch_bounds__(512, 2) void block35_split_kernel(const Block35Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const char* __restrict__ xg = (const char*)a.x + (size_t)img * NPX * a.ldx * 4;
  char* __restrict__ yg = (char*)a.y + (size_t)img * NPX * a.ldy * 4;
  const char* wimg = (const char*)a.wimg;
  const int nt = wave < 3 ? 3 : 2;  // pixel tiles of this wave: wave, wave + 8, wave + 16

  // linear LDS-DMA copy of `total` 1-KiB pieces: this wave takes pieces wave, wave+8, ... (CNT of them; ids past the end
  // repeat the last piece, so every wave issues the same number of DMAs and the counted waits are wave-independent)
  auto copy_lin = [&](const char* src, int lds_off, int total, auto cnt_tag) {
    constexpr int CNT = decltype(cnt_tag)::value;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const int id = min(wave + 8 * i, total - 1);  // wave-uniform
      glds16(src + (size_t)id * 1024 + lane * 16, lds0 + lds_off + id * 1024);
    }
  };
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C4 = std::integral_constant<int, 4>;
  using C5 = std::integral_constant<int, 5>;
  using C8 = std::integral_constant<int, 8>;
  // k-step kt of x (channels 32kt .. 32kt+31) and of W1 -> ring stage: 37 pieces of 8 pixel rows x 128 B (5 per wave) +
  // 12 weight pieces (2 per wave): 7 DMAs per wave
  auto issue_stage = [&](int kt, int stage) {
    const int r = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int id = min(wave + 8 * i, 36);
      const int p = id * 8 + r;
      const int q = slot ^ (p & 7);      // logical slot: hi (q < 4) / lo plane of unit q & 3
      const char* src = p < NPX ? xg + (size_t)p * a.ldx * 4 + (kt * 4 + (q & 3)) * 32 + (q >> 2) * 16 : (const char*)a.zero;
      glds16(src, lds0 + stage * STAGE + id * 1024);
    }
    copy_lin(wimg + (size_t)(S35_W1 + kt * 12) * 1024, stage * STAGE + IMGB, 12, C2{});
  };

  copy_lin(wimg + (size_t)S35_FRAGS * 1024, OFF_BIAS, 2, C1{});  // oldest DMA: landed whenever anything else has
  issue_stage(0, 0);
  issue_stage(1, 1);
  copy_lin(wimg + (size_t)S35_W2 * 1024, OFF_WB, 36, C5{});

  const float* bias = reinterpret_cast<const float*>(smem + OFF_BIAS);
  // ================================================================= phase A: reduce 1x1, 256 -> 96
  f32x4_t accA[6][3];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) accA[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int rbx0 = (16 * wave + frow) * 128 + ((fgrp ^ (frow & 7)) << 4), rbx1 = rbx0 ^ 64;  // + i * 16384 per pixel tile
#pragma unroll 1
  for (int kt = 0; kt < 8; ++kt) {
    // stage kt has landed once only the younger DMAs are outstanding: kt 0, 1: the other stage + W2; 2..6: one stage
    if (kt < 2) wait_vm<12>(); else if (kt < 7) wait_vm<7>(); else wait_vm<0>();
    __syncthreads();
    const char* st = smem + (kt & 1) * STAGE;
    uint4 xh[3], xl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
        xh[i] = *reinterpret_cast<const uint4*>(st + rbx0 + i * 16384);
        xl[i] = *reinterpret_cast<const uint4*>(st + rbx1 + i * 16384);
      }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const uint4 wh = *reinterpret_cast<const uint4*>(st + IMGB + (2 * j) * 1024 + lane * 16);
      const uint4 wl = *reinterpret_cast<const uint4*>(st + IMGB + (2 * j + 1) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) accA[j][i] = mfma_f16(wh, xh[i], accA[j][i]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) accA[j][i] = mfma_f16(wh, xl[i], accA[j][i]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) accA[j][i] = mfma_f16(wl, xh[i], accA[j][i]);
    }
    if (kt < 6) {
      __syncthreads();  // every wave is done with this stage
      issue_stage(kt + 2, kt & 1);
    }
  }
  __syncthreads();  // the ring is dead: images may be written, W3 may land
  copy_lin(wimg + (size_t)S35_W3 * 1024, OFF_WA, 36, C5{});

  // concat fragments of the up convolution, in registers: cfh/cfl[src][pixel tile], k order within a 32-deep step:
  // lane group g, element e -> channel 4g + e (e < 4) | 16 + 4g + (e - 4) of the 32-channel branch output
  uint4 cfh[3][3], cfl[3][3];
  auto to_frag = [&](const f32x4_t& t0, const f32x4_t& t1, uint4& fh, uint4& fl) {
    uint2 h0, l0, h1, l1;
    split4h(t0, h0, l0);
    split4h(t1, h1, l1);
    fh = uint4{h0.x, h0.y, h1.x, h1.y};
    fl = uint4{l0.x, l0.y, l1.x, l1.y};
  };
  // bias + ReLU (zero on the padding pixels) of an accumulator quad
  auto act = [&](const f32x4_t& acc, int boff, int p) {
    const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + boff + 4 * fgrp);
    f32x4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p < NPX ? fmaxf(acc[e] + bv[e], 0.f) : 0.f;
    return v;
  };
  // accumulator quad (channels 16*jt + 4*fgrp .. +3 of the 32-channel image, pixel p) -> image: hi 8 bytes, lo 64 B on
  auto to_image = [&](const f32x4_t& v, int image_off, int jt, int p) {
    uint2 hi, lo;
    split4h(v, hi, lo);
    const int o = image_off + p * 128 + (((2 * jt + (fgrp >> 1)) ^ (p & 7)) << 4) + (fgrp & 1) * 8;
    *reinterpret_cast<uint2*>(smem + o) = hi;
    *reinterpret_cast<uint2*>(smem + (o ^ 64)) = lo;
  };
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i < nt) {
      const int p = 16 * (wave + 8 * i) + frow;
      to_frag(act(accA[0][i], 0, p), act(accA[1][i], 16, p), cfh[0][i], cfl[0][i]);      // b0
      if (p < ROWS) {
        to_image(act(accA[2][i], 32, p), OFF_I0, 0, p);                                   // t1
        to_image(act(accA[3][i], 48, p), OFF_I0, 1, p);
        to_image(act(accA[4][i], 64, p), OFF_I1, 0, p);                                   // t2
        to_image(act(accA[5][i], 80, p), OFF_I1, 1, p);
      }
    }
  __syncthreads();

  // pixel coordinates of this lane's rows (3x3 taps)
  int py[3], px[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = 16 * (wave + 8 * i) + frow;
    py[i] = p < NPX ? p / IMW : -100;
    px[i] = p - (p / IMW) * IMW;
  }
  // ================================================================= phases B, C, D: 3x3 pad 1, 32 -> 32
  //   B: I0 (t1) -> b1 (registers), W2 in WB | C: I1 (t2) -> I0 (t2b), W3 in WA | D: I0 (t2b) -> b2 (registers), W4 in WB
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
    const int src = ph == 1 ? OFF_I1 : OFF_I0;
    const int wb = ph == 1 ? OFF_WA : OFF_WB;
    f32x4_t acc[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      uint4 xh[3], xl[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) {
          const int q = 16 * (wave + 8 * i) + frow + dy * IMW + dx;
          const bool ok = (unsigned)(py[i] + dy) < (unsigned)IMW && (unsigned)(px[i] + dx) < (unsigned)IMW;
          const int o = ok ? q * 128 + ((fgrp ^ (q & 7)) << 4) : ZROW * 128;   // the zero row: all 128 bytes are zero
          xh[i] = *reinterpret_cast<const uint4*>(smem + src + o);
          xl[i] = *reinterpret_cast<const uint4*>(smem + src + (o ^ 64));
        }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint4 wh = *reinterpret_cast<const uint4*>(smem + wb + ((tap * 2 + j) * 2) * 1024 + lane * 16);
        const uint4 wl = *reinterpret_cast<const uint4*>(smem + wb + ((tap * 2 + j) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[j][i] = mfma_f16(wh, xh[i], acc[j][i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[j][i] = mfma_f16(wh, xl[i], acc[j][i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[j][i] = mfma_f16(wl, xh[i], acc[j][i]);
      }
    }
    if (ph == 1) {
      // t2b -> I0: its last readers (phase B) finished before B's closing barrier
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) {
          const int p = 16 * (wave + 8 * i) + frow;
          if (p < ROWS) {
            to_image(act(acc[0][i], 96 + 32, p), OFF_I0, 0, p);
            to_image(act(acc[1][i], 96 + 32 + 16, p), OFF_I0, 1, p);
          }
        }
    } else {
      const int s = ph == 0 ? 1 : 2;   // b1 / b2
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) {
          const int p = 16 * (wave + 8 * i) + frow;
          to_frag(act(acc[0][i], 96 + 32 * ph, p), act(acc[1][i], 96 + 32 * ph + 16, p), cfh[s][i], cfl[s][i]);
        }
    }
    wait_vm<0>();      // B: W3 | C: W4 | D: W5 k-steps 0, 1 -- each issued a whole phase ago
    __syncthreads();
    if (ph == 0) copy_lin(wimg + (size_t)S35_W4 * 1024, OFF_WB, 36, C5{});          // W2 is dead
    if (ph == 1) copy_lin(wimg + (size_t)S35_W5 * 1024, OFF_W5A, 64, C8{});         // I1 and W3 are dead
    if (ph == 2) copy_lin(wimg + (size_t)(S35_W5 + 64) * 1024, OFF_W5B, 32, C4{});  // W4 is dead
  }
  wait_vm<0>();
  __syncthreads();

  // ================================================================= phase E: up 1x1, 96 -> 256, + x, ReLU
  char* const stg = smem + wave * STG_WAVE;
#pragma unroll 1
  for (int g = 0; g < 8; ++g) {       // 32 output channels (two tiles) per round
    // residual of this round's 32 channels first: they travel under the MFMAs.  Lane -> (pixel lane >> 2, 8-channel
    // unit lane & 3) of a pixel tile: one 32-byte unit (hi plane, lo plane) per lane, 128 contiguous bytes per pixel.
    uint4 xrh[3], xrl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
        // unconditional load from a clamped row (rows past the image are never stored)
        const int p = min(16 * (wave + 8 * i) + (lane >> 2), NPX - 1);
        const char* s = xg + (size_t)p * a.ldx * 4 + (4 * g + (lane & 3)) * 32;
        xrh[i] = *reinterpret_cast<const uint4*>(s);
        xrl[i] = *reinterpret_cast<const uint4*>(s + 16);
      }
    f32x4_t acc[2][3];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[jj][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int f = (ks * 16 + 2 * g + jj) * 2;
        const int wo = (ks < 2 ? OFF_W5A : OFF_W5B - 64 * 1024) + f * 1024 + lane * 16;
        const uint4 wh = *reinterpret_cast<const uint4*>(smem + wo);
        const uint4 wl = *reinterpret_cast<const uint4*>(smem + wo + 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[jj][i] = mfma_f16(wh, cfh[ks][i], acc[jj][i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[jj][i] = mfma_f16(wh, cfl[ks][i], acc[jj][i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[jj][i] = mfma_f16(wl, cfh[ks][i], acc[jj][i]);
      }
    }
    // one wait for all residual units here, while no y store of this round is in flight (block35.hip)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt)
        asm volatile("" : "+v"(xrh[i].x), "+v"(xrh[i].y), "+v"(xrh[i].z), "+v"(xrh[i].w), "+v"(xrl[i].x), "+v"(xrl[i].y), "+v"(xrl[i].z),
                     "+v"(xrl[i].w));
    // per pixel tile: 16 px x 32 channels of fp32 through this wave's staging rows, then y = relu((acc + bias) + x) as
    // whole 32-byte units
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 192 + 32 * g + 16 * jj + 4 * fgrp);
          f32x4_t v = acc[jj][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
          *reinterpret_cast<f32x4_t*>(stg + frow * STG_PITCH + (16 * jj + 4 * fgrp) * 4) = v;
        }
        const int pr = lane >> 2, u = lane & 3;
        const int p = 16 * (wave + 8 * i) + pr;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(stg + pr * STG_PITCH + u * 32);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(stg + pr * STG_PITCH + u * 32 + 16);
        const f16x8_t rh = __builtin_bit_cast(f16x8_t, xrh[i]), rl = __builtin_bit_cast(f16x8_t, xrl[i]);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = fmaxf(v0[e] + ((float)rh[e] + (float)rl[e]), 0.f);
          o[4 + e] = fmaxf(v1[e] + ((float)rh[4 + e] + (float)rl[4 + e]), 0.f);
        }
        if (p < NPX) store8<pf16>(yg + (size_t)p * a.ldy * 4 + (4 * g + u) * 32, o);
      }
  }
}

// ---------------------------------------------------------------------------------------------- weight image
// Per block S35_FRAGS fragments of 1 KiB in MFMA A-fragment order, each (16-channel tile, 32-deep k-step) as a (hi, lo)
// pair, then 2 KiB of biases.  Lane l of a fragment holds 8 k values of output channel r0 + (l & 15):
//   W1 (reduce), W2..W4 (3x3): k = k0 + 8*(l>>4) .. +7                                   (natural order)
//   W5 (up): element e of lane group g = l>>4 is k = k0 + 4g + e (e < 4) | k0 + 16 + 4g + (e - 4): the order in which
//            the kernel's register-built B fragments hold a 32-channel branch output
// The packed engine weights keep a K tile of 32 k values as [32 hi halves][32 lo halves] (engine.cpp convert_to F16P).
__global__ void block35s_repack_kernel(Block35Pack p, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int fp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (fp >= S35_FRAGS + 2) return;
  if (fp >= S35_FRAGS) {   // the two trailing KiB: 448 fp32 biases (+ zero padding)
    const int q = (fp - S35_FRAGS) * 64 + lane;  // 16-byte chunk index
    uint4 v = {0u, 0u, 0u, 0u};
    if (q * 4 < B35_BIAS) v = reinterpret_cast<const uint4*>(p.bias)[q];
    out[(size_t)fp * 64 + lane] = v;
    return;
  }
  const int f = fp >> 1, plane = fp & 1;
  int conv, r0, k0;
  if (f < 48) { conv = 0; r0 = 16 * (f % 6); k0 = 32 * (f / 6); }
  else if (f < 102) { const int c = (f - 48) / 18, q = (f - 48) % 18; conv = 1 + c; r0 = 16 * (q & 1); k0 = 32 * (q >> 1); }
  else { conv = 4; r0 = 16 * ((f - 102) & 15); k0 = 32 * ((f - 102) >> 4); }
  const _Float16* w = (const _Float16*)p.w[conv];
  // K tile of 32 k values of a row: 64 halves = [32 hi][32 lo]; k0 is a multiple of 32
  const _Float16* tile = w + ((size_t)(r0 + (lane & 15)) * p.kpad[conv] + k0) * 2 + plane * 32;
  const int g = lane >> 4;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = conv == 4 ? (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4)) : 8 * g + e;
    v[e] = tile[kk];
  }
  out[(size_t)fp * 64 + lane] = __builtin_bit_cast(uint4, v);
}

hipError_t block35s_repack(const Block35Pack& p, void* out, hipStream_t s) {
  hipLaunchKernelGGL(block35s_repack_kernel, dim3((S35_FRAGS + 2 + 3) / 4), dim3(256), 0, s, p, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_block35s(const Block35Args& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  static const hipError_t attr = hipFuncSetAttribute((const void*)block35_split_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, S35_LDS);
  (void)attr;
  (void)hipGetLastError();
  hipLaunchKernelGGL(block35_split_kernel, dim3(a.n), dim3(512), S35_LDS, s, a);
  return hipGetLastError();
}

}  // namespace vnf
