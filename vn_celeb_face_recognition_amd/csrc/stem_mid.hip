// conv2d_2a -> conv2d_2b -> maxpool_3a of the InceptionResnetV1 stem (/root/reference/models/inception_resnet_v1.py:
// 221-224, 282-285) as ONE launch: one workgroup (8 waves) per image walks the image top to bottom, one row per step,
// with every intermediate row in LDS.
//
//   conv2d_1a output (79x79x32, NHWC, from stem_conv1a_kernel)
//     -> 2a: 3x3 valid, 32->32, BN+ReLU  (77x77x32)
//     -> 2b: 3x3 pad 1,  32->64, BN+ReLU (77x77x64)
//     -> maxpool 3x3 stride 2            (38x38x64)  -> global
//
// Unfused these three are HBM/L2-bound launches moving 0.73 GB per 256 images (2a: 102 MB in + 97 MB out, 2b: 97 MB
// in + 194 MB out, pool: 194 MB in + 47 MB out) at 250-330 TFLOP/s; fused, the only traffic is the 102 MB in and the
// 47 MB out, and the bound is the MFMA pipe (84 GFLOP per 256 images).
//
// Software pipeline over rows, one workgroup barrier per step s:
//   DMA    : 1a row s+9 -> input ring (12 rows)                             (LDS-DMA, 5 x 1 KiB pieces per row; the rows come
//            from HBM, a round trip is several steps long: 9 rows of run-ahead, counted wait 5 steps behind the issue)
//   2a     : output row s     from input rows s..s+2       -> A2 ring (4 rows, stored with one zero pixel either side)
//   2b     : output row s-2   from A2 rows s-3..s-1        -> B2 ring (6 rows)   (rows -1 and 77 are a zero row)
//   pool   : output row (s-5)/2 on odd s from B2 rows s-5..s-3 -> global, 16-byte stores
// Wave roles (weights live in REGISTERS, 9 MFMA A-fragments per 16-channel tile, loaded once).  Every wave owns TWO
// channel tiles for its pixel tiles, so one B fragment read from LDS feeds two MFMAs: with one channel tile per wave
// (one ds_read_b128 per MFMA, 325 KiB of fragment reads per step) the kernel ran at the LDS read rate, not the MFMA's.
//   waves 0-3: 2b, channel tiles 2*(wave>>1), +1; pixel tiles {0,1,2} (even waves: 54 MFMAs per step) or {3,4} (odd: 36);
//              they also pool (packed 16-bit integer max: the rows are ReLU outputs)
//   waves 4-7: 2a, both channel tiles; pixel tiles {2} (wave 4), {0,1} (wave 5), {3} (wave 6), {4} (wave 7) -- 18 or 36
//              MFMAs, so that each SIMD's pair (w, w+4) carries 72 / 72 / 72 / 54; they also run conv2d_3b, waves 6,7
//              issue the DMA
// Rounding points and summation order are those of the unfused plan (16-bit rows after each conv, fp32 accumulate over
// (kh,kw,c) in order, bias added after the sum), so the result is bit-identical to it.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_device.h"
#include "stem_mid.h"

namespace vnf {

namespace {

constexpr int W1A = 79, W2 = 77, WP = 38;
constexpr int IN_ROW = 84 * 64;        // 1a row: 79 px x 32 ch (64 B), padded to 84 px so tile 4's taps stay inside
constexpr int A2_ROW = 84 * 64;        // 2a row: col 0 and col 78 are zero padding of 2b, pixels at 1..77
constexpr int B2_ROW = 80 * 128;       // 2b row: 77 px x 64 ch (128 B), 5 pixel tiles
constexpr int IN_RING = 12, A2_RING = 4, B2_RING = 6;
constexpr int AHEAD = 9;               // rows the input DMA runs ahead of conv2d_2a (HBM latency is several steps)
constexpr int OFF_IN = 0, OFF_A2 = OFF_IN + IN_RING * IN_ROW, OFF_B2 = OFF_A2 + A2_RING * A2_ROW;
constexpr int OFF_ZROW = OFF_B2 + B2_RING * B2_ROW;   // zero row (2b's vertical padding)
constexpr int P_ROW = 38 * 128;                       // pooled row (conv2d_3b fusion): 38 px x 64 ch, layout of a 2b row
constexpr int OFF_P = OFF_ZROW + A2_ROW;              // two pooled rows (written at odd steps, consumed one step later)
constexpr int SM_LDS = OFF_P + 2 * P_ROW;

template <typename T> struct MmaS;
template <> struct MmaS<__bf16> {
  static __device__ __forceinline__ f32x4_t run(const uint4& w, const uint4& x, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
  }
};
template <> struct MmaS<_Float16> {
  static __device__ __forceinline__ f32x4_t run(const uint4& w, const uint4& x, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), c, 0, 0, 0);
  }
};

// 16-byte chunk `chunk` (0..3 = 8 channels each) of pixel q inside a [px][64 B] row; XOR swizzle as block35.hip
__device__ __forceinline__ int row_chunk(int q, int chunk) { return q * 64 + ((chunk ^ ((0 - (q >> 2)) & 3)) << 4); }

template <typename T>
__device__ __forceinline__ uint2 pack4s(const f32x4_t& v) {
  typedef T t4 __attribute__((ext_vector_type(4)));
  t4 r = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
  return __builtin_bit_cast(uint2, r);
}

// max of 8 non-negative 16-bit floats (the rows are ReLU outputs): for values >= 0 the bf16 / f16 bit patterns order
// like 16-bit integers, so the maximum is four v_pk_max_i16 -- through fp32 converts the pooling alone cost more VALU
// time than both convolutions' MFMAs
__device__ __forceinline__ uint4 max8(const uint4& a, const uint4& b) {
  typedef short s2 __attribute__((ext_vector_type(2)));
  uint4 r;
  r.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a.x), __builtin_bit_cast(s2, b.x)));
  r.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a.y), __builtin_bit_cast(s2, b.y)));
  r.z = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a.z), __builtin_bit_cast(s2, b.z)));
  r.w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a.w), __builtin_bit_cast(s2, b.w)));
  return r;
}

}  // namespace

template <typename T, bool DBG = false>
__global__ __launch_bounds__(512, 2) void stem_mid_kernel(const StemMidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // instrumented build (VNF_STEM_STAMP): cycles per segment summed over the steps, in scalar registers:
  //   0 prologue | 1 conv row (MFMAs + fragment reads) | 2 its epilogue | 3 pooling | 4 conv2d_3b | 5 DMA issue + wait | 6 barrier
  long long tsum[8];
  long long tlast = 0;
  if constexpr (DBG) {
#pragma unroll
    for (int i = 0; i < 8; ++i) tsum[i] = 0;
    tlast = __builtin_readcyclecounter();
  }
  auto stamp = [&](int seg) {
    if constexpr (DBG) {
      const long long t = __builtin_readcyclecounter();
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i == seg) tsum[i] += t - tlast;
      tlast = t;
    }
  };
  const int frow = lane & 15, fgrp = lane >> 4;
  const int img = blockIdx.x;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const char* __restrict__ xg = (const char*)a.x + (size_t)img * W1A * W1A * a.ldx * 2;
  char* __restrict__ yg = (char*)a.y + (size_t)img * WP * WP * a.ldy * 2;

  // zero the A2 ring (its padding columns stay zero for the whole kernel) and the zero row
  for (int i = tid; i < (A2_RING * A2_ROW + A2_ROW) / 16; i += 512) {
    const int off = i * 16 < A2_RING * A2_ROW ? OFF_A2 + i * 16 : OFF_ZROW + (i * 16 - A2_RING * A2_ROW);
    *reinterpret_cast<uint4*>(smem + off) = uint4{0u, 0u, 0u, 0u};
  }

  // 1a row r -> input ring slot r % 6: 5 pieces of 16 px x 64 B; lane -> (px = lane >> 2, physical slot = lane & 3), the
  // swizzle rides on the source address.  Wave 7 (the lightest: one 2a pixel tile, no pooling) issues all 5 pieces.
  auto issue_row = [&](int r) {
    if (wave != 7) return;
    const int rr = min(r, W1A - 1);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int id = i;
      const int p = id * 16 + (lane >> 2), slot = lane & 3;
      const int pc = min(p, W1A - 1);   // pixels 79..: inside the padding of the ring row, any finite data
      glds16(xg + ((size_t)(rr * W1A + pc) * a.ldx + ((slot ^ ((0 - (p >> 2)) & 3)) << 3)) * 2,
             lds0 + OFF_IN + (r % IN_RING) * IN_ROW + id * 1024);
    }
  };
#pragma unroll
  for (int r = 0; r < AHEAD; ++r) issue_row(r);

  // weights: this wave's 2 x 9 A-fragments (two channel tiles, one fragment per tap); fragment f of tile j in the image
  // at wfrag + (j * 9 + f) * 1 KiB, image order: 2a tiles 0,1 then 2b tiles 0..3
  const int is2b = wave < 4;
  const int ct0 = is2b ? 2 * (wave >> 1) : 0;     // first of the wave's two channel tiles inside its convolution
  uint4 wf[2][9];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      wf[c][t] = reinterpret_cast<const uint4*>(a.wfrag)[(size_t)(((is2b ? 2 : 0) + ct0 + c) * 9 + t) * 64 + lane];
  f32x4_t bias[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) bias[c] = *reinterpret_cast<const f32x4_t*>(a.bias + (is2b ? 32 : 0) + 16 * (ct0 + c) + 4 * fgrp);
  // conv2d_3b on the pooled rows (waves 4..7): wave 4+j owns output-channel tile j for the three pixel tiles, and the
  // fifth tile (channels 64..79) is shared: wave 4+i takes its pixel tile i.  Two A-fragments (K = 64) per channel tile.
  const bool f3b = a.w3b != nullptr;
  uint4 w3[2][2] = {{uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}}, {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}}};
  f32x4_t b3[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  if (f3b && !is2b) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ct = u == 0 ? wave - 4 : 4;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        w3[u][ks] = *reinterpret_cast<const uint4*>((const char*)a.w3b + ((size_t)(16 * ct + frow) * a.k3b_pad + 32 * ks + 8 * fgrp) * 2);
      b3[u] = *reinterpret_cast<const f32x4_t*>(a.b3b + 16 * ct + 4 * fgrp);
    }
  }
  // pixel tiles of this wave (see the role table above)
  const int pt0 = is2b ? ((wave & 1) ? 3 : 0) : (wave == 4 ? 2 : wave == 5 ? 0 : wave - 3);
  const int npt = is2b ? ((wave & 1) ? 2 : 3) : (wave == 5 ? 2 : 1);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Per-lane LDS offsets, computed ONCE: a fragment address is (row base, wave-uniform) + (per-lane offset of the tap's
  // column shift) + (pixel tile) * 1024.  The swizzle term only involves ((pixel >> 2) & 3), which a step of 16 pixels
  // does not change -- recomputing it per fragment cost ~10 VALU instructions per MFMA and made the kernel VALU-bound.
  int rd[3];            // read offset of pixel 16*pt0 + frow + k, k = 0..2 (2a: k = dx; 2b: k = dx + 1 with its +1 column pad)
#pragma unroll
  for (int k = 0; k < 3; ++k) rd[k] = row_chunk(16 * pt0 + frow + k, fgrp);
  int st2a[2], st2b[2];   // store offsets of this lane's channel quad per channel tile (+ i * 1024 / (pt0 + i) * 2048)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int c2a = 16 * c + 4 * fgrp;                                          // 2a output channel of the quad
    st2a[c] = row_chunk(16 * pt0 + frow + 1, c2a >> 3) + (c2a & 4) * 2;
    st2b[c] = frow * 128 + (((2 * (ct0 + c) + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8;
  }
  // pooling: 38 px x 8 chunks of 16 B = 304 items per pooled row: one per lane of the four 2b waves (in-kernel timers:
  // they waited at the step barrier for a third of the kernel while wave 4 -- two pooling passes on top of its
  // convolution and conv2d_3b -- was the last to arrive), the other 48 in a second pass of wave 1 (an odd 2b wave: 36
  // MFMAs per step against the even ones' 54)
  const int pid = is2b ? wave * 64 + lane : -1;
  const int pid2 = (wave == 1 && lane < 48) ? 256 + lane : -1;
  int pl[2][3];         // lane item (ox, ch) -> offset of column 2*ox + dx inside a 2b row (second pass: wave 1)
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int id = max(it ? pid2 : pid, 0), px = 2 * (id >> 3) + dx;
      pl[it][dx] = px * 128 + (((id & 7) ^ (px & 7)) << 4);
    }

  // The step loop, instantiated per role (2b / 2a) and pixel-tile count so that the fragment registers of a step's
  // first two filter rows can be loaded BEFORE the previous step's barrier: those rows were complete a barrier earlier
  // (2b: A2 rows b-1, b written in steps s-3, s-2; 2a: input rows landed two and one DMA waits ago), only the third
  // filter row is new.  In-kernel timers: with all three loaded after the barrier a step's 18-54 MFMAs took 1200-1750
  // cycles, two exposed LDS round trips of 400-500 cycles each on a busy LDS.
  auto steps = [&](auto role_tag, auto npt_tag) {
    constexpr bool IS2B = decltype(role_tag)::value;
    constexpr int NPT = decltype(npt_tag)::value;
    uint4 xa[3][NPT], xb[3][NPT];
    uint4 xc[IS2B ? 1 : 3][NPT];   // 2a: the third filter row as well (its input row landed a DMA wait earlier still)
    // LDS base of filter row dy of this role's output row at step st
    auto row_base = [&](int st, int dy) {
      if constexpr (IS2B) {
        const int rr = st - 2 + dy - 1;   // rows -1 and 77: the zero row
        return (rr >= 0 && rr < W2) ? OFF_A2 + (rr % A2_RING) * A2_ROW : OFF_ZROW;
      } else {
        return OFF_IN + ((st + dy) % IN_RING) * IN_ROW;
      }
    };
    auto active = [&](int st) { return IS2B ? (st >= 2 && st - 2 < W2) : st < W2; };
    // the three taps (dx) of one filter row for every pixel tile; 2b: the A2 column of pixel x + dx is x + dx + 1
    auto load = [&](uint4 (&x)[3][NPT], int base) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < NPT; ++i) x[d][i] = *reinterpret_cast<const uint4*>(smem + base + rd[d] + i * 1024);
    };
    auto prefetch = [&](int st) {
      if (active(st)) {
        load(xa, row_base(st, 0));
        load(xb, row_base(st, 1));
        if constexpr (!IS2B) load(xc, row_base(st, 2));
      }
    };
    prefetch(0);

    for (int s = 0; s < (f3b ? 81 : 80); ++s) {
      // Pooling comes first in the step (two of its three rows were complete two barriers ago)
      if (IS2B && s >= 5 && ((s - 5) & 1) == 0) {
        // pooled row p from 2b rows 2p, 2p+1, 2p+2 (the last one written in the previous step)
        const int p = (s - 5) >> 1;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int id = it ? pid2 : pid;
          if (it == 1 && wave != 1) break;   // wave-uniform
          if (id >= 0) {
            const int ox = id >> 3, ch = id & 7;
            uint4 m = uint4{0u, 0u, 0u, 0u};   // >= every candidate's floor: the rows hold ReLU outputs
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const char* rp = smem + OFF_B2 + ((2 * p + dy) % B2_RING) * B2_ROW;
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) m = max8(m, *reinterpret_cast<const uint4*>(rp + pl[it][dx]));
            }
            if (f3b) *reinterpret_cast<uint4*>(smem + OFF_P + (p & 1) * P_ROW + ox * 128 + ((ch ^ (ox & 7)) << 4)) = m;
            else *reinterpret_cast<uint4*>(yg + ((size_t)(p * WP + ox) * a.ldy + ch * 8) * 2) = m;
          }
        }
        stamp(3);
      }
      if (active(s)) {
        // one output row: filter rows 0 and 1 are in xa / xb since before the barrier; row 2 is fetched under their MFMAs
        f32x4_t acc[2][NPT];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < NPT; ++i) acc[c][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto mma = [&](const uint4 (&x)[3][NPT], int t0) {
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int i = 0; i < NPT; ++i)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c][i] = MmaS<T>::run(wf[c][t0 + d], x[d][i], acc[c][i]);
        };
        if constexpr (IS2B) {
          mma(xa, 0);
          load(xa, row_base(s, 2));
          mma(xb, 3);
          mma(xa, 6);
        } else {
          mma(xa, 0);
          mma(xb, 3);
          mma(xc, 6);
        }
        stamp(1);
        if constexpr (IS2B) {
          char* dst = smem + OFF_B2 + ((s - 2) % B2_RING) * B2_ROW + pt0 * 2048;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
              f32x4_t v = acc[c][i];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias[c][e], 0.f);
              // 2b row: [px][8 chunks of 16 B], chunk index XORed with px & 7 -- unswizzled, the 16 pixels of a store
              // would sit 128 B apart on ONE bank (16-way conflict on every store of every 2b wave)
              *reinterpret_cast<uint2*>(dst + st2b[c] + i * 2048) = pack4s<T>(v);
            }
        } else {
          char* dst = smem + OFF_A2 + (s % A2_RING) * A2_ROW;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
              const int x = 16 * (pt0 + i) + frow;
              f32x4_t v = acc[c][i];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias[c][e], 0.f);
              if (x < W2) *reinterpret_cast<uint2*>(dst + st2a[c] + i * 1024) = pack4s<T>(v);
            }
        }
        stamp(2);
      }
      if constexpr (!IS2B) {
      if (f3b && s >= 6 && ((s - 6) & 1) == 0) {
        // conv2d_3b of pooled row p (in LDS since the previous step): same k order and rounding points as the plan's
        // 1x1 convolution (two 32-deep MFMA steps, bias after the sum, ReLU, 16-bit store)
        const int p = (s - 6) >> 1;
        const char* pr = smem + OFF_P + (p & 1) * P_ROW;
        auto bfrag = [&](int pt, int ks) {
          const int px = min(16 * pt + frow, WP - 1);
          return *reinterpret_cast<const uint4*>(pr + px * 128 + (((4 * ks + fgrp) ^ (px & 7)) << 4));
        };
        auto emit = [&](int ct, int pt, const f32x4_t& acc, const f32x4_t& bb) {
          const int px = 16 * pt + frow;
          if (px < WP) {
            f32x4_t v = acc;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bb[e], 0.f);
            *reinterpret_cast<uint2*>(yg + ((size_t)(p * WP + px) * a.ldy + 16 * ct + 4 * fgrp) * 2) = pack4s<T>(v);
          }
        };
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
          const uint4 x0 = bfrag(pt, 0), x1 = bfrag(pt, 1);
          f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
          acc = MmaS<T>::run(w3[0][0], x0, acc);
          acc = MmaS<T>::run(w3[0][1], x1, acc);
          emit(wave - 4, pt, acc, b3[0]);
          if (pt == wave - 4) {       // wave-uniform: the shared fifth channel tile, this wave's pixel tile
            f32x4_t acc4 = {0.f, 0.f, 0.f, 0.f};
            acc4 = MmaS<T>::run(w3[1][0], x0, acc4);
            acc4 = MmaS<T>::run(w3[1][1], x1, acc4);
            emit(4, pt, acc4, b3[1]);
          }
        }
        stamp(4);
      }
      }
      // Row s+9 goes out now (its slot held row s-3, last read -- prefetched -- at the end of step s-4).  Row r is first
      // read at the end of step r-3 (the prefetch of step r-2's third filter row), after the barrier of step r-4, i.e.
      // 5 steps after its issue; every step issues >= 5 memory operations on the DMA wave, so "all but the 25 youngest
      // complete" at the end of each step retires every piece at least 5 steps old.
      if (s + AHEAD < W1A) issue_row(s + AHEAD);
      // next step's first two filter rows (complete since the previous barrier at the latest)
      prefetch(s + 1);
      if (wave == 7) {
        // ... and once the last row has gone out, the allowance shrinks by one row per step so that the guarantee
        // ("every piece at least 5 steps old has landed") also holds for the image's last rows
        const int j = s + AHEAD - (W1A - 1);   // steps since the last issue
        if (j <= 0) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
        else if (j == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (j == 2) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (j == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (j == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      stamp(5);
      // The barrier needs this wave's LDS WRITES done, not the prefetch reads just issued behind them (LDS operations
      // retire in order): wait for all but the youngest KPF.  __syncthreads() would wait for lgkmcnt(0) and put the
      // prefetch's whole round trip back in front of every barrier.  (Global stores -- conv2d_3b's -- need not drain.)
      constexpr int KPF = (IS2B ? 6 : 9) * NPT < 15 ? (IS2B ? 6 : 9) * NPT : 15;
      if (active(s + 1))
        asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(KPF) : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      stamp(6);
    }
  };
  using N1 = std::integral_constant<int, 1>;
  using N2 = std::integral_constant<int, 2>;
  using N3 = std::integral_constant<int, 3>;
  stamp(0);
  if (is2b) {
    if (npt == 3) steps(std::true_type{}, N3{}); else steps(std::true_type{}, N2{});
  } else {
    if (npt == 2) steps(std::false_type{}, N2{}); else steps(std::false_type{}, N1{});
  }
  if constexpr (DBG) {
    if (a.dbg && blockIdx.x == 100 && lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a.dbg[wave * 8 + i] = tsum[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight fragments
// 6 tiles x 9 taps of 1 KiB in MFMA A-fragment order: tiles 0,1 = conv2d_2a channels 0..31, tiles 2..5 = conv2d_2b
// channels 0..63; lane l of fragment (tile, tap) holds k = 32*tap + 8*(l>>4) .. +7 of output channel 16*tile' + (l&15).
__global__ void stem_mid_repack_kernel(StemMidPack p, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (f >= 54) return;
  const int tile = f / 9, tap = f % 9;
  const int conv = tile < 2 ? 0 : 1, r0 = 16 * (tile < 2 ? tile : tile - 2);
  const char* w = (const char*)p.w[conv];
  out[(size_t)f * 64 + lane] = *reinterpret_cast<const uint4*>(w + ((size_t)(r0 + (lane & 15)) * p.kpad[conv] + 32 * tap + 8 * (lane >> 4)) * 2);
}

hipError_t stem_mid_repack(const StemMidPack& p, void* out, hipStream_t s) {
  hipLaunchKernelGGL(stem_mid_repack_kernel, dim3(14), dim3(256), 0, s, p, (uint4*)out);
  return hipGetLastError();
}

hipError_t launch_stem_mid(const StemMidArgs& a, int dtype, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)stem_mid_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS);
    (void)hipFuncSetAttribute((const void*)stem_mid_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS);
    (void)hipGetLastError();
    attr_done = true;
  }
  if (dtype == BF16 && a.n > 100 && getenv("VNF_STEM_STAMP")) {
    // instrumented launch: appends workgroup 100's per-wave segment sums (see the kernel) to the named file
    static long long* dbuf = nullptr;
    if (!dbuf && hipMalloc((void**)&dbuf, 64 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(dbuf, 0, 64 * 8, s);
    StemMidArgs aa = a;
    aa.dbg = dbuf;
    (void)hipFuncSetAttribute((const void*)stem_mid_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS);
    hipLaunchKernelGGL((stem_mid_kernel<__bf16, true>), dim3(a.n), dim3(512), SM_LDS, s, aa);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    long long host[64];
    (void)hipMemcpy(host, dbuf, 64 * 8, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(getenv("VNF_STEM_STAMP"), "a")) {
      fprintf(f, "launch n=%d 3b=%d\n", a.n, a.w3b != nullptr);
      for (int w = 0; w < 8; ++w) {
        fprintf(f, "%d", w);
        for (int i = 0; i < 8; ++i) fprintf(f, " %lld", host[w * 8 + i]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    return hipSuccess;
  }
  if (dtype == BF16)
    hipLaunchKernelGGL(stem_mid_kernel<__bf16>, dim3(a.n), dim3(512), SM_LDS, s, a);
  else if (dtype == F16)
    hipLaunchKernelGGL(stem_mid_kernel<_Float16>, dim3(a.n), dim3(512), SM_LDS, s, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace vnf
