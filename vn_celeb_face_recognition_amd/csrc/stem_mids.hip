// conv2d_2a -> conv2d_2b -> maxpool_3a -> conv2d_3b of the InceptionResnetV1 stem (/root/reference/models/
// inception_resnet_v1.py:221-224, 282-286) as ONE launch in PLANAR SPLIT-F16 (the encoders' f16x2 dtype, split_f16.h):
// the in-gate twin of stem_mid.hip.  One workgroup (8 waves) per image walks the image top to bottom, one row per step,
// one workgroup barrier per step, every intermediate row in LDS.
//
// Operands are (hi, lo) pairs of f16 planes, products three MFMAs each, so a row costs twice the LDS of the 16-bit
// kernel; what makes it fit:
//   * the 2b rows are never stored: the 2b waves keep the running VERTICAL maximum of the pooling window in fp32
//     registers (a wave owns the same channels of the same pixels in every row) and emit one fp32 row per pooled row;
//     the pooling waves only take the horizontal maximum of that row;
//   * 2b's zero rows above and below the image are skipped filter rows (wave-uniform), not a zero row in LDS;
//   * the input ring is 6 rows (3 in use + 3 in flight).
// A row of 32 channels is [pixel][128 B]: hi planes of the four 8-channel units, then the lo planes, 16-byte slots
// XOR-swizzled by pixel & 7 (conflict-free for any 16 consecutive pixels = one tap's fragment read).
//
// Software pipeline, step s:
//   DMA    : 1a row s+5 -> input ring (10 pieces of 8 px x 128 B, waves 6 and 7)
//   2a     : output row s    from input rows s..s+2                 -> A2 ring (4 rows, one zero pixel either side)
//   2b     : output row s-2  from A2 rows s-3..s-1 (waves 0..3)     -> running vertical max; on even rows >= 2 the
//            maximum of rows 2p..2p+2 goes to the VM row (fp32) as pooled row p = (s-4)/2
//   pool   : step s+1: horizontal max of the VM row -> (hi, lo) planes -> P row
//   3b     : step s+2: conv2d_3b (1x1, 64 -> 80) of the P row -> global
// Wave roles (weights in REGISTERS: 18 fragments per 16-channel tile):
//   waves 0-3: 2b, channel tile = wave, all 5 pixel tiles of the row
//   waves 4-7: 2a, channel tile = wave & 1, pixel tiles {0,1,2} (waves 4,5) or {3,4} (waves 6,7); pooling; conv2d_3b
#include <type_traits>

#include "conv_device.h"
#include "stem_mid.h"

namespace vnf {

namespace {

constexpr int W1A = 79, W2 = 77, WP = 38;
constexpr int ROWB = 84 * 128;         // a 32-channel row: 84 pixel slots (taps of tile 4 stay inside) x 128 B
constexpr int IN_RING = 6, A2_RING = 4, AHEAD = 5;
constexpr int VM_ROW = 80 * 256;       // vertical-max row: 80 px x 64 channels fp32
constexpr int P_ROW = 48 * 256;        // pooled row as (hi, lo) planes: [px][2 k-tiles x 128 B], 3 pixel tiles
constexpr int OFF_IN = 0, OFF_A2 = OFF_IN + IN_RING * ROWB, OFF_VM = OFF_A2 + A2_RING * ROWB, OFF_P = OFF_VM + VM_ROW;
constexpr int SMS_LDS = OFF_P + P_ROW;
static_assert(SMS_LDS <= 160 * 1024, "LDS map");

__device__ __forceinline__ void split4q(const f32x4_t& v, uint2& hi, uint2& lo) {
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

__global__ __launch_bounds__(512, 2) void stem_mid_split_kernel(const StemMidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const char* __restrict__ xg = (const char*)a.x + (size_t)img * W1A * W1A * a.ldx * 4;
  char* __restrict__ yg = (char*)a.y + (size_t)img * WP * WP * a.ldy * 4;

  // zero the A2 ring (its padding columns stay zero for the whole kernel) and the P row (pixels 38..47 stay zero)
  for (int i = tid; i < (A2_RING * ROWB + P_ROW) / 16; i += 512) {
    const int off = i * 16 < A2_RING * ROWB ? OFF_A2 + i * 16 : OFF_P + (i * 16 - A2_RING * ROWB);
    *reinterpret_cast<uint4*>(smem + off) = uint4{0u, 0u, 0u, 0u};
  }

  // 1a row r -> input ring slot r % 6: 10 pieces of 8 px x 128 B; lane -> (px = lane >> 3, physical slot = lane & 7), the
  // swizzle and the hi / lo plane selection ride on the source address.  Waves 6 and 7 issue 5 pieces each.
  auto issue_row = [&](int r) {
    if (wave < 6) return;
    const int rr = min(r, W1A - 1);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int id = (wave - 6) * 5 + i;
      const int p = id * 8 + (lane >> 3), slot = lane & 7;
      const int pc = min(p, W1A - 1);   // pixel 79: inside the padding of the ring row, any finite data
      const int q = slot ^ (p & 7);     // logical slot: hi (q < 4) / lo plane of unit q & 3
      glds16(xg + (size_t)(rr * W1A + pc) * a.ldx * 4 + (q & 3) * 32 + (q >> 2) * 16, lds0 + OFF_IN + (r % IN_RING) * ROWB + id * 1024);
    }
  };
#pragma unroll
  for (int r = 0; r < AHEAD; ++r) issue_row(r);

  // weights: this wave's 18 A-fragments ((hi, lo) per tap) of its channel tile
  const int is2b = wave < 4;
  const int wtile = is2b ? 2 + wave : (wave & 1);   // image order: 2a tiles 0,1 then 2b tiles 0..3
  uint4 wh[9], wl[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wh[t] = reinterpret_cast<const uint4*>(a.wfrag)[(size_t)((wtile * 9 + t) * 2) * 64 + lane];
    wl[t] = reinterpret_cast<const uint4*>(a.wfrag)[(size_t)((wtile * 9 + t) * 2 + 1) * 64 + lane];
  }
  const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(a.bias + (is2b ? 32 + 16 * wave : 16 * (wave & 1)) + 4 * fgrp);
  // conv2d_3b on the pooled rows (waves 4..7): wave 4+j owns output-channel tile j for the three pixel tiles, and the
  // fifth tile (channels 64..79) is shared: wave 4+i takes its pixel tile i.  (hi, lo) fragments of the two 32-deep
  // steps, read from the packed engine weights ([row][K tile: 32 hi | 32 lo])
  // (one register array serves both roles -- a wave is either a 2b wave or a 2a wave for the whole kernel, but the
  // register allocator cannot know: aux[0..7] = conv2d_3b fragments [tile u][k-step][hi, lo] on waves 4..7, aux[0..4] =
  // the running vertical maximum of the pooling window on waves 0..3)
  uint4 aux[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) aux[i] = uint4{0u, 0u, 0u, 0u};
  f32x4_t b3[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  if (!is2b) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ct = u == 0 ? wave - 4 : 4;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char* wp = (const char*)a.w3b + ((size_t)(16 * ct + frow) * a.k3b_pad + 32 * ks) * 4 + fgrp * 16;
        aux[(u * 2 + ks) * 2] = *reinterpret_cast<const uint4*>(wp);
        aux[(u * 2 + ks) * 2 + 1] = *reinterpret_cast<const uint4*>(wp + 64);
      }
      b3[u] = *reinterpret_cast<const f32x4_t*>(a.b3b + 16 * ct + 4 * fgrp);
    }
  }
  // pixel tiles of this wave: 2b waves 0..4; 2a waves 4,5: 0..2, waves 6,7: 3..4
  const int pt0 = is2b ? 0 : (wave < 6 ? 0 : 3), npt = is2b ? 5 : (wave < 6 ? 3 : 2);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Per-lane LDS offsets, computed ONCE (stem_mid.hip): fragment address = row base (wave-uniform) + per-lane offset of
  // the tap's column shift + pixel tile * 2048; the lo plane sits at the same address ^ 64.
  int rd[3];            // read offset of pixel 16*pt0 + frow + k, k = 0..2 (2a: k = dx; 2b: k = dx + 1 with its +1 column pad)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int px = 16 * pt0 + frow + k;
    rd[k] = px * 128 + ((fgrp ^ (px & 7)) << 4);
  }
  const int c2a = 16 * (wave & 1) + 4 * fgrp;                                  // 2a output channel of this lane's quad
  const int x2a = 16 * pt0 + frow + 1;                                         // A2 column of this lane's pixel (tile 0)
  const int st2a = x2a * 128 + (((c2a >> 3) ^ (x2a & 7)) << 4) + (c2a & 4) * 2;   // + i * 2048
  // VM row (fp32): [px][16 chunks of 16 B], chunk XORed with px & 15 -- unswizzled the 16 pixels of a store would sit
  // 256 B apart on one bank
  const int stvm = frow * 256 + (((4 * wave + fgrp) ^ frow) << 4);              // + i * 4096 (2b waves: wave < 4)


  // one output row of a 3x3 convolution for NPT pixel tiles, filter rows [dy0, dy1); fragments are fetched one filter
  // row (3 taps x NPT x (hi, lo)) at a time -- the partner wave on the SIMD covers the read latency
  auto conv_row = [&](auto npt_tag, auto addr, int dy0, int dy1, f32x4_t* acc) {
    constexpr int NPT = decltype(npt_tag)::value;
#pragma unroll
    for (int i = 0; i < NPT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (dy < dy0 || dy >= dy1) continue;   // wave-uniform: 2b's zero rows above / below the image
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        uint4 xh[NPT], xl[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
          const int o = addr(dy, dx, i);
          xh[i] = *reinterpret_cast<const uint4*>(smem + o);
          xl[i] = *reinterpret_cast<const uint4*>(smem + (o ^ 64));
        }
        const int t = dy * 3 + dx;
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = mfma_f16(wh[t], xh[i], acc[i]);
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = mfma_f16(wh[t], xl[i], acc[i]);
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = mfma_f16(wl[t], xh[i], acc[i]);
      }
    }
  };
  using N2 = std::integral_constant<int, 2>;
  using N3 = std::integral_constant<int, 3>;
  using N5 = std::integral_constant<int, 5>;

  for (int s = 0; s < 81; ++s) {
    if (is2b) {
      const int b = s - 2;   // 2b output row
      if (b >= 0 && b < W2) {
        f32x4_t acc[5];
        int rb[3];   // wave-uniform row bases of the three filter rows
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rb[dy] = OFF_A2 + ((b + dy - 1 + A2_RING) % A2_RING) * ROWB;
        // A2 column of pixel x + dx - 1 is x + dx (pixels live at columns 1..77): offsets rd[0..2]
        conv_row(N5{}, [&](int dy, int dx, int i) { return rb[dy] + rd[dx] + i * 2048; }, b == 0 ? 1 : 0, b == W2 - 1 ? 2 : 3, acc);
        const bool emit = (b & 1) == 0 && b >= 2;   // row b closes the window of pooled row (b - 2) / 2 and opens the next
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          f32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][e] + bias[e], 0.f);
          const f32x4_t vmi = __builtin_bit_cast(f32x4_t, aux[i]);   // zeros before row 0: <= every ReLU output
          f32x4_t m;
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(vmi[e], v[e]);
          if (emit) {
            *reinterpret_cast<f32x4_t*>(smem + OFF_VM + stvm + i * 4096) = m;
            aux[i] = __builtin_bit_cast(uint4, v);
          } else {
            aux[i] = __builtin_bit_cast(uint4, m);
          }
        }
      }
    } else {
      const int r2 = s;      // 2a output row
      if (r2 < W2) {
        f32x4_t acc[3];
        int rb[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rb[dy] = OFF_IN + ((r2 + dy) % IN_RING) * ROWB;
        auto addr = [&](int dy, int dx, int i) { return rb[dy] + rd[dx] + i * 2048; };
        if (npt == 3) conv_row(N3{}, addr, 0, 3, acc); else conv_row(N2{}, addr, 0, 3, acc);
        char* dst = smem + OFF_A2 + (r2 % A2_RING) * ROWB;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < npt) {
            const int x = 16 * (pt0 + i) + frow;
            f32x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][e] + bias[e], 0.f);
            uint2 hi, lo;
            split4q(v, hi, lo);
            if (x < W2) {
              *reinterpret_cast<uint2*>(dst + st2a + i * 2048) = hi;
              *reinterpret_cast<uint2*>(dst + ((st2a + i * 2048) ^ 64)) = lo;
            }
          }
      }
      if (s >= 5 && ((s - 5) & 1) == 0 && s <= 79) {
        // horizontal max of the VM row written in the previous step (pooled row p): 38 px x 8 units of 8 channels = 304
        // items over the 256 lanes of waves 4..7 -> (hi, lo) planes in the P row
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int id = (wave - 4) * 64 + lane + 256 * it;
          if (id < WP * 8) {
            const int ox = id >> 3, u = id & 7;
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = 0.f;   // <= every candidate: ReLU outputs
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int px = 2 * ox + dx;
              const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(smem + OFF_VM + px * 256 + (((2 * u) ^ (px & 15)) << 4));
              const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(smem + OFF_VM + px * 256 + (((2 * u + 1) ^ (px & 15)) << 4));
#pragma unroll
              for (int e = 0; e < 4; ++e) { m[e] = fmaxf(m[e], v0[e]); m[4 + e] = fmaxf(m[4 + e], v1[e]); }
            }
            f16x8_t h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const sf16 sp(m[e]);
              h[e] = sp.hi; l[e] = sp.lo;
            }
            const int o = OFF_P + ox * 256 + (u >> 2) * 128 + (((u & 3) ^ (ox & 7)) << 4);
            *reinterpret_cast<f16x8_t*>(smem + o) = h;
            *reinterpret_cast<f16x8_t*>(smem + (o ^ 64)) = l;
          }
        }
      }
      if (s >= 6 && ((s - 6) & 1) == 0) {
        // conv2d_3b of pooled row p (in the P row since the previous step)
        const int p = (s - 6) >> 1;
        auto emit3 = [&](int ct, int pt, const f32x4_t& acc, const f32x4_t& bb) {
          const int px = 16 * pt + frow;
          if (px < WP) {
            f32x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[e] + bb[e], 0.f);
            uint2 hi, lo;
            split4q(v, hi, lo);
            char* d = yg + (size_t)(p * WP + px) * a.ldy * 4 + (2 * ct + (fgrp >> 1)) * 32 + (fgrp & 1) * 8;
            *reinterpret_cast<uint2*>(d) = hi;
            *reinterpret_cast<uint2*>(d + 16) = lo;
          }
        };
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
          const int px = 16 * pt + frow;   // < 48: inside the (zero-padded) P row
          uint4 xh[2], xl[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int o = OFF_P + px * 256 + ks * 128 + ((fgrp ^ (px & 7)) << 4);
            xh[ks] = *reinterpret_cast<const uint4*>(smem + o);
            xl[ks] = *reinterpret_cast<const uint4*>(smem + (o ^ 64));
          }
          f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            acc = mfma_f16(aux[ks * 2], xh[ks], acc);
            acc = mfma_f16(aux[ks * 2], xl[ks], acc);
            acc = mfma_f16(aux[ks * 2 + 1], xh[ks], acc);
          }
          emit3(wave - 4, pt, acc, b3[0]);
          if (pt == wave - 4) {       // wave-uniform: the shared fifth channel tile, this wave's pixel tile
            f32x4_t acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              acc4 = mfma_f16(aux[(2 + ks) * 2], xh[ks], acc4);
              acc4 = mfma_f16(aux[(2 + ks) * 2], xl[ks], acc4);
              acc4 = mfma_f16(aux[(2 + ks) * 2 + 1], xh[ks], acc4);
            }
            emit3(4, pt, acc4, b3[1]);
          }
        }
      }
    }
    // Row s+5 goes out now (its slot held row s-1, last read in the previous step).  Row r is first read at step r-2,
    // i.e. 3 steps after its issue; a DMA wave issues 5 pieces per step, so "all but the 10 youngest memory operations
    // complete" at the end of each step retires every piece at least 2 steps old (the y stores of conv2d_3b in the window
    // only make the wait stricter).
    if (s + AHEAD < W1A) issue_row(s + AHEAD);
    if (wave >= 6) {
      // once the last row has gone out the allowance shrinks with it, so the image's last rows are covered as well
      const int j = s + AHEAD - (W1A - 1);   // steps since the last issue
      if (j <= 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (j == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- weight fragments
// 6 tiles x 9 taps x (hi, lo) of 1 KiB in MFMA A-fragment order: tiles 0,1 = conv2d_2a channels 0..31, tiles 2..5 =
// conv2d_2b channels 0..63; lane l of fragment (tile, tap, plane) holds k = 32*tap + 8*(l>>4) .. +7 of output channel
// 16*tile' + (l&15).  Packed engine weights: K tile of 32 k values = [32 hi halves][32 lo halves].
__global__ void stem_mids_repack_kernel(StemMidPack p, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int fp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (fp >= 108) return;
  const int f = fp >> 1, plane = fp & 1;
  const int tile = f / 9, tap = f % 9;
  const int conv = tile < 2 ? 0 : 1, r0 = 16 * (tile < 2 ? tile : tile - 2);
  const char* w = (const char*)p.w[conv];
  out[(size_t)fp * 64 + lane] =
      *reinterpret_cast<const uint4*>(w + ((size_t)(r0 + (lane & 15)) * p.kpad[conv] + 32 * tap) * 4 + plane * 64 + (lane >> 4) * 16);
}

hipError_t stem_mids_repack(const StemMidPack& p, void* out, hipStream_t s) {
  hipLaunchKernelGGL(stem_mids_repack_kernel, dim3(27), dim3(256), 0, s, p, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_stem_mids(const StemMidArgs& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  if (!a.w3b || !a.b3b) return hipErrorInvalidValue;   // the split kernel always carries conv2d_3b
  static const hipError_t attr = hipFuncSetAttribute((const void*)stem_mid_split_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMS_LDS);
  (void)attr;
  (void)hipGetLastError();
  hipLaunchKernelGGL(stem_mid_split_kernel, dim3(a.n), dim3(512), SMS_LDS, s, a);
  return hipGetLastError();
}

}  // namespace vnf
