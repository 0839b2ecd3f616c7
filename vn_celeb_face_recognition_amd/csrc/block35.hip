// One Block35 (/root/reference/models/inception_resnet_v1.py:36-67) per launch, one workgroup (8 waves) per 17x17x256
// image, every intermediate in LDS:
//
//   x -> reduce 1x1 256->(32|32|32) -> { b0 ; b1 = 3x3(t1) ; b2 = 3x3(3x3(t2)) } -> up 1x1 96->256 -> y = relu(x + up)
//
// The unfused plan spends five launches per block on GEMMs of N = 32 (the three 3x3 branches run at 3-4 % of the MFMA
// peak: one 10-20 us latency chain each); here the whole block is one launch whose only HBM/L2 traffic is x (read twice:
// once as the reduce operand, once as the residual) and y.
//
// Work split: by PIXELS.  The image is 19 tiles of 16 pixels (304 rows, the last 15 are padding; the 7 of them held in LDS are kept at zero); wave w
// owns tiles w, w+8, w+16.  Every wave therefore needs every weight, so weights live in LDS -- in MFMA A-fragment order
// (1 KiB per (k-step, 16-channel tile): lane l reads its 16 bytes at fragment*1024 + l*16, conflict-free, and the
// create-time image is copied by linear LDS-DMA) -- while activations live in row-major pixel images the 3x3 taps can
// address: [304 rows][64 B] per 32-channel image, 16-byte chunks XOR-swizzled by ((-(row>>2)) & 3).
//
// LDS (160 KiB exactly):
//   [0, 75776)        phase A: two stages of the x K-tile ring ([296 rows][128 B] each, slot ^ (row&7) swizzle, filled by
//                     LDS-DMA with the swizzle on the source address); afterwards the four 32-channel images
//                     b0 | t1 -> t2b | t2 -> b2 | b1
//   [75776, 124928)   W1 (reduce, 48 fragments) during phase A, then W5 (up, 48 fragments), prefetched under phases B-D
//   [124928, 161792)  two 3x3 weight buffers (18 fragments each); in phase E the per-wave fp32 staging of the epilogue
//   [161792, 163840)  the block's 448 fp32 biases (they arrive by LDS-DMA with the weights: a plain global load beside
//                     in-flight DMAs would make the compiler's own vmcnt waits drain the whole DMA queue)
//
// Phases and barriers: A (4 K tiles of x through the ring) | B: b1 = 3x3(t1) | C: t2b = 3x3(t2) | D: b2 = 3x3(t2b) |
// E: up + residual + ReLU, staged per wave through LDS so x is read and y written in whole 128-byte rows.
#include <type_traits>

#include "block35.h"
#include "conv_device.h"

namespace vnf {

namespace {

constexpr int NPX = 289, IMW = 17;
constexpr int ROWS = 296;                         // image rows kept in LDS: 289 pixels rounded up to whole 8-row DMA pieces
constexpr int ZROW = 295;                         // a padding row, always zero: source of out-of-image taps
constexpr int XR_STAGE = ROWS * 128;              // 37888
constexpr int IMG_BYTES = ROWS * 64;              // 18944
constexpr int OFF_W = 2 * XR_STAGE;               // 75776
constexpr int OFF_W33 = OFF_W + 48 * 1024;        // 124928
constexpr int W33_BYTES = 18 * 1024;
constexpr int OFF_BIAS = OFF_W33 + 2 * W33_BYTES; // 161792: 448 fp32 biases (2 KiB)
constexpr int B35_LDS = OFF_BIAS + 2048;          // 163840
static_assert(B35_LDS == 160 * 1024 && 4 * IMG_BYTES == 2 * XR_STAGE, "LDS map");
// Pixel tile 18 spans rows 288..303: its lanes of rows >= 296 read past their image / ring stage (still inside the
// workgroup's LDS).  What they read only reaches the MFMA columns of those padding pixels, which are never stored.
constexpr int STG_PITCH = 272;                    // fp32 staging row: 64 channels + 16 B pad
constexpr int STG_WAVE = 16 * STG_PITCH;          // 4352 B per wave

template <typename T> struct Mma35;
template <> struct Mma35<__bf16> {
  static __device__ __forceinline__ f32x4_t run(const uint4& w, const uint4& x, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
  }
};
template <> struct Mma35<_Float16> {
  static __device__ __forceinline__ f32x4_t run(const uint4& w, const uint4& x, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ uint2 pack4_35(const f32x4_t& v) {
  typedef T t4 __attribute__((ext_vector_type(4)));
  t4 r = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
  return __builtin_bit_cast(uint2, r);
}

// byte offset of the 16-byte chunk `chunk` (0..3) of pixel row q inside a 32-channel image
__device__ __forceinline__ int img_chunk(int q, int chunk) { return q * 64 + ((chunk ^ ((0 - (q >> 2)) & 3)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace

template <typename T>
__global__ __launch_bounds__(512, 2) void block35_kernel(const Block35Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  // x and y are different tensors (the plan ping-pongs the block buffers): without __restrict__ every y store is
  // followed by a vmcnt(0) before the next x load
  const char* __restrict__ xg = (const char*)a.x + (size_t)img * NPX * a.ldx * 2;
  char* __restrict__ yg = (char*)a.y + (size_t)img * NPX * a.ldy * 2;
  const char* wimg = (const char*)a.wimg;
  const int nt = wave < 3 ? 3 : 2;  // pixel tiles of this wave: wave, wave + 8, wave + 16

  // linear LDS-DMA copy of `total` 1-KiB pieces: this wave takes pieces wave, wave+8, ... (CNT of them; ids past the
  // end repeat the last piece -- same bytes to the same place -- so every wave issues the same number of DMAs and the
  // counted waits below are wave-independent)
  auto copy_lin = [&](const char* src, int lds_off, int total, auto cnt_tag) {
    constexpr int CNT = decltype(cnt_tag)::value;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const int id = min(wave + 8 * i, total - 1);  // wave-uniform
      glds16(src + (size_t)id * 1024 + lane * 16, lds0 + lds_off + id * 1024);
    }
  };
  // K tile kt of x -> ring stage: 37 pieces of 8 pixel rows x 128 B, 5 per wave
  auto issue_x = [&](int kt, int stage) {
    const int r = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int id = min(wave + 8 * i, 36);
      const int p = id * 8 + r;
      const char* src = p < NPX ? xg + ((size_t)p * a.ldx + kt * 64 + ((slot ^ (p & 7)) << 3)) * 2 : (const char*)a.zero;
      glds16(src, lds0 + stage * XR_STAGE + id * 1024);
    }
  };
  using C3 = std::integral_constant<int, 3>;
  using C5 = std::integral_constant<int, 5>;
  using C6 = std::integral_constant<int, 6>;

  using C1 = std::integral_constant<int, 1>;
  copy_lin(wimg + B35_BIASOFF, OFF_BIAS, 2, C1{});  // oldest DMA: landed whenever anything else has
  copy_lin(wimg + B35_W1, OFF_W, 48, C6{});
  issue_x(0, 0);
  issue_x(1, 1);
  copy_lin(wimg + B35_W2, OFF_W33, 36, C5{});  // W2 and W3 are contiguous in the image

  const float* bias = reinterpret_cast<const float*>(smem + OFF_BIAS);
  // ================================================================= phase A: reduce 1x1, 256 -> 96
  f32x4_t accA[6][3];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) accA[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int rbx0 = (16 * wave + frow) * 128 + ((fgrp ^ (frow & 7)) << 4), rbx1 = rbx0 ^ 64;  // + i * 16384 per pixel tile
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    // tile kt (and W1) have landed once at most the younger DMAs are outstanding: X1+W23 / W23+X2 / X3 / none
    if (kt == 0) wait_vm<10>(); else if (kt == 1) wait_vm<10>(); else if (kt == 2) wait_vm<5>(); else wait_vm<0>();
    __syncthreads();
    const char* xr = smem + (kt & 1) * XR_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 xf[3], wf[6];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) xf[i] = *reinterpret_cast<const uint4*>(xr + (ks ? rbx1 : rbx0) + i * 16384);
#pragma unroll
      for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const uint4*>(smem + OFF_W + ((kt * 2 + ks) * 6 + j) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) accA[j][i] = Mma35<T>::run(wf[j], xf[i], accA[j][i]);
    }
    if (kt < 2) {
      __syncthreads();  // every wave is done with this stage
      issue_x(kt + 2, kt & 1);
    }
  }
  __syncthreads();  // the ring and W1 are dead: images may be written, W5 may land
  copy_lin(wimg + B35_W5, OFF_W, 48, C6{});
  // ReLU -> 16 bits -> images 0 (b0), 1 (t1), 2 (t2); padding rows stay zero
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
        const int p = 16 * (wave + 8 * i) + frow;
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 16 * j + 4 * fgrp);
        f32x4_t v = accA[j][i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p < NPX ? fmaxf(v[e] + bv[e], 0.f) : 0.f;
        if (p < ROWS)
          *reinterpret_cast<uint2*>(smem + (j >> 1) * IMG_BYTES + img_chunk(p, 2 * (j & 1) + (fgrp >> 1)) + (fgrp & 1) * 8) = pack4_35<T>(v);
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
  //   B: image 1 (t1) -> image 3 (b1), W2 | C: image 2 (t2) -> image 1 (t2b), W3 | D: image 1 -> image 2 (b2), W4
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
    const int src = (ph == 0 ? 1 : ph == 1 ? 2 : 1) * IMG_BYTES;
    const int dst = (ph == 0 ? 3 : ph == 1 ? 1 : 2) * IMG_BYTES;
    const int wb = OFF_W33 + (ph == 1 ? W33_BYTES : 0);
    f32x4_t acc[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      uint4 xf[3], wf[2];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) {
          const int q = 16 * (wave + 8 * i) + frow + dy * IMW + dx;
          const bool ok = (unsigned)(py[i] + dy) < (unsigned)IMW && (unsigned)(px[i] + dx) < (unsigned)IMW;
          xf[i] = *reinterpret_cast<const uint4*>(smem + src + (ok ? img_chunk(q, fgrp) : ZROW * 64));  // a zero row
        }
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const uint4*>(smem + wb + (tap * 2 + j) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[j][i] = Mma35<T>::run(wf[j], xf[i], acc[j][i]);
    }
    // no barrier needed before the writes: each phase writes an image nobody reads in this phase (B: 3, C: 1 -- whose
    // last readers finished before B's closing barrier -- D: 2, read last in C)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nt) {
          const int p = 16 * (wave + 8 * i) + frow;
          const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 96 + 32 * ph + 16 * j + 4 * fgrp);
          f32x4_t v = acc[j][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p < NPX ? fmaxf(v[e] + bv[e], 0.f) : 0.f;
          if (p < ROWS) *reinterpret_cast<uint2*>(smem + dst + img_chunk(p, 2 * j + (fgrp >> 1)) + (fgrp & 1) * 8) = pack4_35<T>(v);
        }
    if (ph == 1) wait_vm<0>();  // W4 (issued after B) and W5 have landed before the barrier that opens D / E
    __syncthreads();
    if (ph == 0) copy_lin(wimg + B35_W4, OFF_W33, 18, C3{});  // W2 is dead
  }

  // ================================================================= phase E: up 1x1, 96 -> 256, + x, ReLU
  uint4 cf[3][3];  // concat fragments: [k-step = image b0, b1, b2][pixel tile]
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
        const int p = 16 * (wave + 8 * i) + frow;
        const int im = ks == 0 ? 0 : ks == 1 ? 3 : 2;
        cf[ks][i] = *reinterpret_cast<const uint4*>(smem + im * IMG_BYTES + img_chunk(p, fgrp));
      }
  float* stg = reinterpret_cast<float*>(smem + OFF_W33 + wave * STG_WAVE);
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    // residual chunks of this group's 64 channels first: they travel under the MFMAs (lane = 16-byte chunk id & 63 of
    // a 16 px x 8 chunk tile, two chunks per lane and pixel tile)
    uint4 xres[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          // unconditional load from a clamped row (a select between a load and a constant would put every load in its
          // own branch with its own vmcnt(0)); rows past the image are never stored
          const int id = lane + 64 * r, p = min(16 * (wave + 8 * i) + (id >> 3), NPX - 1);
          xres[i][r] = *reinterpret_cast<const uint4*>(xg + ((size_t)p * a.ldx + 64 * g + 8 * (id & 7)) * 2);
        }
      }
    f32x4_t acc[4][3];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[jj][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      uint4 wf[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) wf[jj] = *reinterpret_cast<const uint4*>(smem + OFF_W + (ks * 16 + 4 * g + jj) * 1024 + lane * 16);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nt) acc[jj][i] = Mma35<T>::run(wf[jj], cf[ks][i], acc[jj][i]);
    }
    // One wait for all residual chunks HERE, while no y store of this group is in flight: vmcnt counts loads and stores
    // together and the compiler cannot count a load past a younger store, so a first use further down would become a
    // vmcnt(0) behind every preceding store (a full store round trip per 16 pixels).
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
#pragma unroll
        for (int r = 0; r < 2; ++r) asm volatile("" : "+v"(xres[i][r].x), "+v"(xres[i][r].y), "+v"(xres[i][r].z), "+v"(xres[i][r].w));
      }
    // per pixel tile: 16 px x 64 channels of fp32 through this wave's staging rows, then whole 128-byte rows:
    // y = relu(x + acc), 8 lanes per pixel row
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nt) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          // (acc + bias) + x, in the unfused epilogue's order: the block is bit-identical to the five-launch plan
          const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + 192 + 64 * g + 16 * jj + 4 * fgrp);
          f32x4_t v = acc[jj][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
          *reinterpret_cast<f32x4_t*>(reinterpret_cast<char*>(stg) + frow * STG_PITCH + (16 * jj + 4 * fgrp) * 4) = v;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int id = lane + 64 * r, pr = id >> 3, cc = id & 7;
          const int p = 16 * (wave + 8 * i) + pr;
          const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<char*>(stg) + pr * STG_PITCH + cc * 32);
          const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<char*>(stg) + pr * STG_PITCH + cc * 32 + 16);
          typedef T t8 __attribute__((ext_vector_type(8)));
          const t8 xr = __builtin_bit_cast(t8, xres[i][r]);
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = fmaxf(v0[e] + (float)xr[e], 0.f); o[4 + e] = fmaxf(v1[e] + (float)xr[4 + e], 0.f); }
          if (p < NPX) store8<T>(yg + ((size_t)p * a.ldy + 64 * g + 8 * cc) * 2, o);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------- weight image
// Per block: W1 | W2 | W3 | W4 | W5 in MFMA A-fragment order (lane l of fragment f holds k = k0 + 8*(l>>4) .. +7 of row
// r0 + (l&15)):  W1: f = ks*6 + j  (r0 = 16j, k0 = 32ks)   W2..4: f = tap*2 + j  (k0 = 32 tap)   W5: f = ks*16 + j.
__global__ void block35_repack_kernel(Block35Pack p, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (f >= B35_FRAGS + 2) return;
  if (f >= B35_FRAGS) {   // the two trailing KiB: 448 fp32 biases (+ zero padding)
    const int q = (f - B35_FRAGS) * 64 + lane;  // 16-byte chunk index
    uint4 v = {0u, 0u, 0u, 0u};
    if (q * 4 < B35_BIAS) v = reinterpret_cast<const uint4*>(p.bias)[q];
    out[(size_t)f * 64 + lane] = v;
    return;
  }
  int conv, r0, k0;
  if (f < 48) { conv = 0; r0 = 16 * (f % 6); k0 = 32 * (f / 6); }
  else if (f < 102) { const int c = (f - 48) / 18, q = (f - 48) % 18; conv = 1 + c; r0 = 16 * (q & 1); k0 = 32 * (q >> 1); }
  else { conv = 4; r0 = 16 * ((f - 102) & 15); k0 = 32 * ((f - 102) >> 4); }
  const char* w = (const char*)p.w[conv];
  out[(size_t)f * 64 + lane] = *reinterpret_cast<const uint4*>(w + ((size_t)(r0 + (lane & 15)) * p.kpad[conv] + k0 + 8 * (lane >> 4)) * 2);
}

hipError_t block35_repack(const Block35Pack& p, void* out, hipStream_t s) {
  hipLaunchKernelGGL(block35_repack_kernel, dim3((B35_FRAGS + 2 + 3) / 4), dim3(256), 0, s, p, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_block35(const Block35Args& a, int dtype, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)block35_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, B35_LDS);
    (void)hipFuncSetAttribute((const void*)block35_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, B35_LDS);
    (void)hipGetLastError();
    attr_done = true;
  }
  if (dtype == BF16)
    hipLaunchKernelGGL(block35_kernel<__bf16>, dim3(a.n), dim3(512), B35_LDS, s, a);
  else if (dtype == F16)
    hipLaunchKernelGGL(block35_kernel<_Float16>, dim3(a.n), dim3(512), B35_LDS, s, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace vnf
