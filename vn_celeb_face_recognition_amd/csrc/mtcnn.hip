// MTCNN P/R/O-Net cascade on the device (gfx950), restating
//   /root/reference/models/mtcnn_utils/detect_face.py:25-185 (detect_face) and helpers 188-306,
//   /root/reference/models/mtcnn.py:38-49, 84-99, 138-157 (the three nets), 326-347 (area ordering).
//
// Everything between "frames are in HBM" and "final boxes" stays on the device: the image pyramid,
// the three nets, threshold + compaction, all four NMS passes, box regression / squaring / padding
// and the per-candidate crop+resize that the reference does in Python loops.  fp32 everywhere
// (thin channels: 3..128; the path is HBM / latency bound, not FLOP bound), explicit op order
// (built with -ffp-contract=off, FMAs only where written) so box arithmetic and IoU tests are
// bit-identical to the fp32 numpy / torch-CPU statements of the oracle.
//
// Kernel map:
//   K1 pyramid_kernel        u8 frame -> all pyramid levels (adaptive-average bins, normalised)
//   K2 pnet_conv1_pool / pnet_conv2 / pnet_conv3_heads   (all levels and frames per launch)
//   K3 threshold + compaction fused into pnet_conv3_heads (wave-aggregated atomic slots;
//      order restored by the sort keys, which carry the cell index)
//   K4 nms_scale_kernel (per level x frame, IoU 0.5), nms_image_kernel (per frame, IoU 0.7,
//      + regress, rerec, pad), stage2_post_kernel (IoU 0.7 + bbreg + rerec + pad)
//   K5 crop_resize_kernel    box table -> N x 3 x {24,48}^2 (area bins, also up-sampling)
//   K6 rnet_kernel / onet_kernel   one workgroup per candidate, activations resident in LDS
//   K7 stage3_post_kernel    landmarks, bbreg, "Min" NMS, area-descending order
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <cmath>
#include <vector>

#include "engine.h"
#include "nms_device.h"
#include "split_f16.h"

namespace vnf {

typedef float float2_t __attribute__((ext_vector_type(2)));

constexpr int MAX_LEVELS = 24;
// Candidate tables.  Stage 1 is sized by the pyramid itself: every (level, frame) list has room for all cells of the
// level, so it cannot overflow.  The stage-2 / stage-3 tables hold `keep` rows per frame, a RUN-TIME capacity
// (vnf_mtcnn_cfg.max_candidates, default KEEP).  The NMS kernels keep their sort keys and kept boxes in LDS while a
// list fits the constants below and switch to global-memory scratch beyond them -- the reference has no cap at all
// (detect_face.py:79-93,203-218) and neither has the arithmetic here; only the row tables of stages 2 / 3 are bounded,
// by a capacity the caller can raise (the host layer grows it and retries on VNF_E_CAPACITY).
constexpr int CAP_LDS_KEYS = 8192;   // sort keys held in LDS by the stage-1 NMS kernels
constexpr int KEEP = 2048;           // kept boxes / post-kernel keys held in LDS; default rows per frame of the stage tables


struct LevelDesc {
  int Hs, Ws, Hp, Wp, H2, W2, oh, ow;
  float scale;
  int off_px, off_p1, off_c2, off_out;  // prefix offsets (in pixels of that stage) over levels
};

struct LevelTable {
  int n;
  int tot_px, tot_p1, tot_c2, tot_out;
  LevelDesc l[MAX_LEVELS];
};

struct PNetW {  // transposed to [cin][3][3][cout] so one tap's output-channel weights are contiguous
  const float *w1, *b1, *a1, *w2, *b2, *a2, *w3, *b3, *a3, *w41, *b41, *w42, *b42;
};

struct Cand { float score, r0, r1, r2, r3; int cell; };
struct Row { float x1, y1, x2, y2, score; int y, ey, x, ex; };  // stage-2 / stage-3 table row

__device__ __forceinline__ int find_level(const LevelTable& t, int idx, int which) {
  int l = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i) {
    const int off = which == 0 ? t.l[i].off_px : which == 1 ? t.l[i].off_p1 : which == 2 ? t.l[i].off_c2 : t.l[i].off_out;
    if (idx >= off) l = i;
  }
  return l;
}

// --------------------------------------------------------------------------------------------- K1
// detect_face.py:71-72: imresample(imgs, (int(h*s+1), int(w*s+1))) then (x-127.5)*0.0078125.
// interpolate(mode='area') == adaptive average pooling: bin [floor(i*H/oh), ceil((i+1)*H/oh)),
// value = sum / kh / kw (two divisions, as ATen rounds).  Pixel sums of 8-bit data are exact in fp32.
__global__ void pyramid_kernel(const uint8_t* __restrict__ frames, int H, int W, LevelTable t, float* __restrict__ lvl) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= t.tot_px) return;
  const int img = blockIdx.y;
  const int li = find_level(t, idx, 0);
  const LevelDesc L = t.l[li];
  const int p = idx - L.off_px, y = p / L.Ws, x = p - y * L.Ws;
  const int h0 = (int)(((long long)y * H) / L.Hs), h1 = (int)((((long long)(y + 1)) * H + L.Hs - 1) / L.Hs);
  const int w0 = (int)(((long long)x * W) / L.Ws), w1 = (int)((((long long)(x + 1)) * W + L.Ws - 1) / L.Ws);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const uint8_t* base = frames + (size_t)img * H * W * 3;
  for (int yy = h0; yy < h1; ++yy) {
    const uint8_t* row = base + ((size_t)yy * W + w0) * 3;
    for (int xx = 0; xx < w1 - w0; ++xx) {
      s0 += (float)row[3 * xx];
      s1 += (float)row[3 * xx + 1];
      s2 += (float)row[3 * xx + 2];
    }
  }
  const float kh = (float)(h1 - h0), kw = (float)(w1 - w0);
  float* o = lvl + ((size_t)img * 3) * t.tot_px + L.off_px + p;
  o[0] = ((s0 / kh) / kw - 127.5f) * 0.0078125f;
  o[(size_t)t.tot_px] = ((s1 / kh) / kw - 127.5f) * 0.0078125f;
  o[2 * (size_t)t.tot_px] = ((s2 / kh) / kw - 127.5f) * 0.0078125f;
}

// Fast path of K1 for rows that are a whole number of 16-byte chunks (W*3 % 16 == 0: 1920, 1280, 640 ...).
// One workgroup per (output row of any level, frame): every lane streams 16-byte chunks of the input
// rows of that bin row (fully coalesced; each level re-reads the u8 frame once, from L2 / Infinity
// Cache after the first), keeps per-byte column sums in registers, parks them in LDS, and the
// output pixels then add their horizontal spans.  Integer sums, so the result is bit-identical to
// the per-pixel kernel above and to the oracle.
__global__ void __launch_bounds__(256) pyramid_rows_kernel(const uint8_t* __restrict__ frames, int H, int W,
                                                            LevelTable t, float* __restrict__ lvl, const int* __restrict__ row_order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* colsum = reinterpret_cast<unsigned*>(smem);  // W*3 entries
  // XCD-aware dispatch: blockIdx.x is the FRAME (workgroups go to the 8 XCDs round-robin in linear order, so with a
  // multiple of 8 frames every frame stays on one XCD and its private L2), blockIdx.y walks the output rows of ALL levels
  // in the order of the frame rows they read (row_order: sorted by the bin's first input row, tall bins first on ties):
  // the nine levels' readers of a band of the frame run together and the band is fetched from beyond L2 once, not 9x
  const int packed = row_order[blockIdx.y];
  const int li = packed >> 16, r = packed & 0xFFFF;
  const LevelDesc L = t.l[li];
  const int img = blockIdx.x, i = r;
  const int h0 = (int)(((long long)i * H) / L.Hs), h1 = (int)((((long long)(i + 1)) * H + L.Hs - 1) / L.Hs);
  const int rowb = W * 3, nchunk = rowb >> 4;
  const uint8_t* base = frames + (size_t)img * H * rowb;
  for (int c0 = 0; c0 < nchunk; c0 += 512) {   // 2 chunks per thread per sweep
    unsigned acc[2][16];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[q][j] = 0u;
    const int ca = c0 + threadIdx.x, cb = ca + 256;
    const int cac = min(ca, nchunk - 1), cbc = min(cb, nchunk - 1);   // clamped: unconditional loads, results dropped below
    // packed accumulation: bytes 0,2 and 1,3 of every dword add up in two 16-bit lanes of one register (5 VALU ops per
    // dword instead of 11); 256 rows of 255 fit in 16 bits, then the packed sums are flushed into the 32-bit ones
    for (int y0 = h0; y0 < h1; y0 += 256) {
      const int y1 = min(h1, y0 + 256);
      unsigned pe[2][4], po[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int d = 0; d < 4; ++d) { pe[q][d] = 0u; po[q][d] = 0u; }
      // two rows per round, four 16-byte loads in flight; the second row is clamped and masked at an odd tail
      // (slower: four rows per round -- the bins of the first level are only 4-5 rows tall; six-row rounds for the
      // tall bins of the small levels, 0.154 ms -- more requests in flight only crowd the memory system)
      for (int yy = y0; yy < y1; yy += 2) {
        const int yb = min(yy + 1, y1 - 1);
        const unsigned mb = (yy + 1 < y1) ? 0x00FF00FFu : 0u;
        const uint4* r0 = reinterpret_cast<const uint4*>(base + (size_t)yy * rowb);
        const uint4* r1 = reinterpret_cast<const uint4*>(base + (size_t)yb * rowb);
        const uint4 v00 = r0[cac], v01 = r0[cbc], v10 = r1[cac], v11 = r1[cbc];
        const unsigned w0[2][4] = {{v00.x, v00.y, v00.z, v00.w}, {v01.x, v01.y, v01.z, v01.w}};
        const unsigned w1[2][4] = {{v10.x, v10.y, v10.z, v10.w}, {v11.x, v11.y, v11.z, v11.w}};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            pe[q][d] += (w0[q][d] & 0x00FF00FFu) + (w1[q][d] & mb);
            po[q][d] += ((w0[q][d] >> 8) & 0x00FF00FFu) + ((w1[q][d] >> 8) & mb);
          }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          acc[q][d * 4 + 0] += pe[q][d] & 0xFFFFu;
          acc[q][d * 4 + 1] += po[q][d] & 0xFFFFu;
          acc[q][d * 4 + 2] += pe[q][d] >> 16;
          acc[q][d * 4 + 3] += po[q][d] >> 16;
        }
    }
    if (ca < nchunk) {
#pragma unroll
      for (int j = 0; j < 16; ++j) colsum[ca * 16 + j] = acc[0][j];
    }
    if (cb < nchunk) {
#pragma unroll
      for (int j = 0; j < 16; ++j) colsum[cb * 16 + j] = acc[1][j];
    }
  }
  __syncthreads();
  const float kh = (float)(h1 - h0);
  for (int x = threadIdx.x; x < L.Ws; x += blockDim.x) {
    const int w0 = (int)(((long long)x * W) / L.Ws), w1 = (int)((((long long)(x + 1)) * W + L.Ws - 1) / L.Ws);
    unsigned s0 = 0, s1 = 0, s2 = 0;
    for (int xx = w0; xx < w1; ++xx) { s0 += colsum[3 * xx]; s1 += colsum[3 * xx + 1]; s2 += colsum[3 * xx + 2]; }
    const float kw = (float)(w1 - w0);
    float* o = lvl + ((size_t)img * 3) * t.tot_px + L.off_px + i * L.Ws + x;
    o[0] = (((float)s0 / kh) / kw - 127.5f) * 0.0078125f;
    o[(size_t)t.tot_px] = (((float)s1 / kh) / kw - 127.5f) * 0.0078125f;
    o[2 * (size_t)t.tot_px] = (((float)s2 / kh) / kw - 127.5f) * 0.0078125f;
  }
}

// --------------------------------------------------------------------------------------------- K2
// mtcnn.py:39-41: conv1 3->10 (3x3) + PReLU, then MaxPool2d(2,2,ceil_mode=True); one thread per
// pooled pixel, all 10 channels in registers.
__global__ void pnet_conv1_pool_kernel(const float* __restrict__ lvl, LevelTable t, PNetW w, float* __restrict__ p1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= t.tot_p1) return;
  const int img = blockIdx.y;
  const int li = find_level(t, idx, 1);
  const LevelDesc L = t.l[li];
  const int p = idx - L.off_p1, py = p / L.Wp, px = p - py * L.Wp;
  const int Hc = L.Hs - 2, Wc = L.Ws - 2;
  float in[3][4][4];
  const float* src = lvl + ((size_t)img * 3) * t.tot_px + L.off_px;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        // clamped address, no predicate: the 48 loads issue back to back instead of one exec-masked round
        // trip each.  A clamped (out-of-level) value only reaches conv outputs beyond (Hc, Wc), which the
        // window test below skips, so it never contributes.
        const int yy = min(2 * py + dy, L.Hs - 1), xx = min(2 * px + dx, L.Ws - 1);
        in[c][dy][dx] = src[(size_t)c * t.tot_px + yy * L.Ws + xx];
      }
  // the four conv positions of the pooling window advance together, tap by tap: every weight is fetched once per
  // thread (they are wave-uniform scalar loads whose latency the FMAs of the previous tap cover), not once per position.
  // Positions outside (Hc, Wc) are computed on clamped inputs and discarded below.
  float2_t acc2[4][5];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc2[q][j] = float2_t{w.b1[2 * j], w.b1[2 * j + 1]};
  // software-pipelined by hand: tap i+1's ten weights are requested before tap i's 20 packed FMAs; the per-tap
  // barriers keep the compiler from hoisting all 270 scalar loads to the top (they do not fit the SGPR file: it
  // spilled them to VGPR lanes and paid 3 780 v_readlane for 540 FMAs)
  float2_t wv[2][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) wv[0][j] = float2_t{w.w1[2 * j], w.w1[2 * j + 1]};
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int c = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
    if (tap + 1 < 27) {
      const float* ww = w.w1 + (tap + 1) * 10;
#pragma unroll
      for (int j = 0; j < 5; ++j) wv[(tap + 1) & 1][j] = float2_t{ww[2 * j], ww[2 * j + 1]};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = in[c][(q >> 1) + kh][(q & 1) + kw];
      const float2_t v2 = {v, v};
#pragma unroll
      for (int j = 0; j < 5; ++j) acc2[q][j] = __builtin_elementwise_fma(v2, wv[tap & 1][j], acc2[q][j]);
    }
    // every accumulator is made opaque here, so the tap's 20 FMAs cannot be sunk into per-output chains
    // (LLVM otherwise finishes one accumulator over all 27 taps before starting the next, re-reading every weight)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(acc2[q][j]));
    __builtin_amdgcn_sched_barrier(0);
  }
  float best[10];
#pragma unroll
  for (int co = 0; co < 10; ++co) best[co] = -INFINITY;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = 2 * py + (q >> 1) < Hc && 2 * px + (q & 1) < Wc;
#pragma unroll
    for (int co = 0; co < 10; ++co) {
      const float av = acc2[q][co >> 1][co & 1];
      const float a = av > 0.f ? av : av * w.a1[co];
      best[co] = ok ? fmaxf(best[co], a) : best[co];
    }
  }
  float* o = p1 + ((size_t)img * 10) * t.tot_p1 + L.off_p1 + p;
#pragma unroll
  for (int co = 0; co < 10; ++co) o[(size_t)co * t.tot_p1] = best[co];
}

// K2 on the matrix pipe.  The per-pixel kernel above needs its 270 weights as wave-uniform scalars at every tap; they
// do not fit the SGPR file, so each wave fetches them again and again and waits on that (measured: 134 us per 16
// frames, 30 us with the weights held fixed).  Here the weights are an MFMA operand: 9 VGPRs per lane, loaded once.
// One workgroup per (pooled row of any level, frame): the four input rows go to LDS, a wave takes tiles of
// 2 conv rows x 8 conv columns = 16 pixels, one v_mfma_f32_16x16x4_f32 per tap (A = weights [16 ch pad][4 ch pad],
// B = pixels, accumulator preset with the bias), PReLU, the 2x2 max over the lane quartet {l, l^1, l^8, l^9} by DPP,
// pooled values staged in LDS and written as whole rows per channel.
typedef float f32x4p_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
}
__device__ __forceinline__ int pnet_row_stride(int Ws) { return ((Ws + 2 + 13) & ~31) + 18 >= Ws + 2 ? ((Ws + 2 + 13) & ~31) + 18 : ((Ws + 2 + 13) & ~31) + 50; }

__global__ void __launch_bounds__(256) pnet_conv1_pool_mfma_kernel(const float* __restrict__ lvl, LevelTable t, PNetW w,
                                                                    float* __restrict__ p1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int li = 0, py = blockIdx.x;
  while (li + 1 < t.n && py >= t.l[li].Hp) { py -= t.l[li].Hp; ++li; }
  const LevelDesc L = t.l[li];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int Hc = L.Hs - 2, Wc = L.Ws - 2;
  const int Wsp = pnet_row_stride(L.Ws);            // LDS row stride: = 18 mod 32 floats, so the (row, channel) lane groups
  float* s_in = reinterpret_cast<float*>(smem);     // [3 ch][4 rows][Wsp]            spread over the banks
  float* s_out = s_in + 12 * Wsp;                   // [10][Wp]
  const float* src = lvl + ((size_t)img * 3) * t.tot_px + L.off_px;
  for (int i = tid; i < 12 * L.Ws; i += 256) {
    const int rr = i / L.Ws, x = i - rr * L.Ws, c = rr >> 2, dy = rr & 3;
    const int yy = min(2 * py + dy, L.Hs - 1);       // a clamped row only feeds conv rows >= Hc, which are masked
    s_in[rr * Wsp + x] = src[(size_t)c * t.tot_px + yy * L.Ws + x];
  }
  const int wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lm = lane & 15, dy = lm >> 3, dx = lm & 7;
  float wa[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) wa[tap] = (lm < 10 && lg < 3) ? w.w1[(lg * 9 + tap) * 10 + lm] : 0.f;
  float bias[4], slope[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = lg * 4 + e;
    bias[e] = ch < 10 ? w.b1[ch] : 0.f;
    slope[e] = ch < 10 ? w.a1[ch] : 0.f;
  }
  __syncthreads();
  const int cg = min(lg, 2);                          // channel 3 is the zero pad of the k dimension (its weights are 0)
  const bool row_ok = 2 * py + dy < Hc;
  const int ntile = (L.Wp + 3) >> 2;
  for (int tx = wave; tx < ntile; tx += 8) {          // tiles tx and tx + 4 together: two accumulator chains interleave
    f32x4p_t acc[2];
    float xb[2][9];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int x0 = min(8 * (tx + 4 * u) + dx, L.Ws - 3);   // clamped columns only feed conv columns >= Wc (masked)
      const float* b0 = s_in + (cg * 4 + dy) * Wsp + x0;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xb[u][kh * 3 + kw] = b0[kh * Wsp + kw];
      acc[u] = f32x4p_t{bias[0], bias[1], bias[2], bias[3]};
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tap], xb[0][tap], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tap], xb[1][tap], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int txu = tx + 4 * u;
      const bool ok = row_ok && 8 * txu + dx < Wc;
      const int pxx = 4 * txu + (dx >> 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[u][e] > 0.f ? acc[u][e] : acc[u][e] * slope[e];
        v = ok ? v : -INFINITY;
        v = fmaxf(v, dpp_xor1(v));
        v = fmaxf(v, dpp_xor8(v));
        if ((lm & 9) == 0 && pxx < L.Wp && lg * 4 + e < 10) s_out[(lg * 4 + e) * L.Wp + pxx] = v;
      }
    }
  }
  __syncthreads();
  float* o = p1 + ((size_t)img * 10) * t.tot_p1 + L.off_p1 + (size_t)py * L.Wp;
  for (int i = tid; i < 10 * L.Wp; i += 256) {
    const int ch = i / L.Wp, x = i - ch * L.Wp;
    o[(size_t)ch * t.tot_p1 + x] = s_out[i];
  }
}

// Same arithmetic without the LDS stage: every lane fetches its nine B values straight from the level (the 9-fold
// reuse between taps and the 2-column overlap of neighbouring tiles are L1 hits), two tiles in flight per wave so the
// two accumulator chains interleave, the next pair's loads issued before the current pair's MFMAs.  No barrier at all.
__global__ void __launch_bounds__(256) pnet_conv1_pool_direct_kernel(const float* __restrict__ lvl, LevelTable t, PNetW w,
                                                                      float* __restrict__ p1) {
  int li = 0, py = blockIdx.y;   // frame on x: one XCD (and its L2) per frame, see pyramid_rows_kernel
  while (li + 1 < t.n && py >= t.l[li].Hp) { py -= t.l[li].Hp; ++li; }
  const LevelDesc L = t.l[li];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int Hc = L.Hs - 2, Wc = L.Ws - 2;
  const int wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lm = lane & 15, dy = lm >> 3, dx = lm & 7;
  float wa[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) wa[tap] = (lm < 10 && lg < 3) ? w.w1[(lg * 9 + tap) * 10 + lm] : 0.f;
  float bias[4], slope[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = lg * 4 + e;
    bias[e] = ch < 10 ? w.b1[ch] : 0.f;
    slope[e] = ch < 10 ? w.a1[ch] : 0.f;
  }
  const float* g0 = lvl + ((size_t)img * 3 + min(lg, 2)) * t.tot_px + L.off_px;
  const float* rowp[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) rowp[kh] = g0 + (size_t)min(2 * py + dy + kh, L.Hs - 1) * L.Ws;
  const bool row_ok = 2 * py + dy < Hc;
  const int ntile = (L.Wp + 3) >> 2;
  float* o = p1 + ((size_t)img * 10) * t.tot_p1 + L.off_p1 + (size_t)py * L.Wp;
  auto fetch = [&](int tx, float (&xb)[9]) {
    const int x0 = min(8 * tx + dx, L.Ws - 3);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xb[kh * 3 + kw] = rowp[kh][x0 + kw];
  };
  auto finish = [&](int tx, f32x4p_t acc) {
    const bool ok = row_ok && 8 * tx + dx < Wc;
    const int pxx = 4 * tx + (dx >> 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[e] > 0.f ? acc[e] : acc[e] * slope[e];
      v = ok ? v : -INFINITY;
      v = fmaxf(v, dpp_xor1(v));
      v = fmaxf(v, dpp_xor8(v));
      if ((lm & 9) == 0 && pxx < L.Wp && lg * 4 + e < 10) o[(size_t)(lg * 4 + e) * t.tot_p1 + pxx] = v;
    }
  };
  // tiles wave, wave+4 form the first pair, then +8 ...; a tile index beyond ntile is clamped for the loads and skipped
  // (one pooled row per WAVE instead of per workgroup was slower: the long rows of the first level then set the time)
  float xa[9], xb[9];
  int tx = wave;
  if (tx >= ntile) return;
  fetch(tx, xa);
  fetch(min(tx + 4, ntile - 1), xb);
  for (; tx < ntile; tx += 8) {
    float na[9], nb[9];
    const int nx = tx + 8;
    if (nx < ntile) { fetch(nx, na); fetch(min(nx + 4, ntile - 1), nb); }
    f32x4p_t a0 = {bias[0], bias[1], bias[2], bias[3]}, a1 = a0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tap], xa[tap], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tap], xb[tap], a1, 0, 0, 0);
    }
    finish(tx, a0);
    if (tx + 4 < ntile) finish(tx + 4, a1);
    if (nx < ntile) {
#pragma unroll
      for (int i = 0; i < 9; ++i) { xa[i] = na[i]; xb[i] = nb[i]; }
    }
  }
}

// mtcnn.py:42-43: conv2 10->16 (3x3) + PReLU
__global__ void pnet_conv2_kernel(const float* __restrict__ p1, LevelTable t, PNetW w, float* __restrict__ c2) {
  const int idx = blockIdx.y * blockDim.x + threadIdx.x;
  if (idx >= t.tot_c2) return;
  const int img = blockIdx.x;
  const int li = find_level(t, idx, 2);
  const LevelDesc L = t.l[li];
  const int p = idx - L.off_c2, y = p / L.W2, x = p - y * L.W2;
  const float* src = p1 + ((size_t)img * 10) * t.tot_p1 + L.off_p1;
  float2_t acc2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc2[j] = float2_t{w.b2[2 * j], w.b2[2 * j + 1]};
#pragma unroll 1
  for (int c = 0; c < 10; ++c) {
    const float* sc = src + (size_t)c * t.tot_p1 + y * L.Wp + x;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float v = sc[kh * L.Wp + kw];
        const float2_t v2 = {v, v};
        const float* ww = w.w2 + ((c * 3 + kh) * 3 + kw) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[j] = __builtin_elementwise_fma(v2, float2_t{ww[2 * j], ww[2 * j + 1]}, acc2[j]);
      }
  }
  float* o = c2 + ((size_t)img * 16) * t.tot_c2 + L.off_c2 + p;
#pragma unroll
  for (int co = 0; co < 16; ++co) {
    const float a = acc2[co >> 1][co & 1];
    o[(size_t)co * t.tot_c2] = a > 0.f ? a : a * w.a2[co];
  }
}

// mtcnn.py:44-49: conv3 16->32 + PReLU, conv4_1 (1x1 ->2) + softmax, conv4_2 (1x1 -> 4);
// detect_face.py:209: mask = prob[:,1] >= thr, fused: survivors are appended to the
// (level, frame) candidate list.  prob_dbg / reg_dbg (optional) receive the dense maps.
__global__ void pnet_conv3_heads_kernel(const float* __restrict__ c2, LevelTable t, PNetW w, float thr, int B, int cap_out,
                                        Cand* __restrict__ cand, int* __restrict__ cells, int* __restrict__ cand_cnt,
                                        float* __restrict__ prob_dbg, float* __restrict__ reg_dbg) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= t.tot_out) return;
  const int img = blockIdx.y;
  const int li = find_level(t, idx, 3);
  const LevelDesc L = t.l[li];
  const int p = idx - L.off_out, y = p / L.ow, x = p - y * L.ow;
  const float* src = c2 + ((size_t)img * 16) * t.tot_c2 + L.off_c2;
  // two output channels per v_pk_fma_f32: the same IEEE fma per channel in the same (c,kh,kw) order, at twice
  // the scalar-FMA rate (this kernel is FMA-bound: 4608 FMAs per output cell)
  float2_t acc2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc2[j] = float2_t{w.b3[2 * j], w.b3[2 * j + 1]};
#pragma unroll 1
  for (int c = 0; c < 16; ++c) {
    const float* sc = src + (size_t)c * t.tot_c2 + y * L.W2 + x;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float v = sc[kh * L.W2 + kw];
        const float2_t v2 = {v, v};
        const float* ww = w.w3 + ((c * 3 + kh) * 3 + kw) * 32;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc2[j] = __builtin_elementwise_fma(v2, float2_t{ww[2 * j], ww[2 * j + 1]}, acc2[j]);
      }
  }
  float acc[32];
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc[2 * j] = acc2[j][0]; acc[2 * j + 1] = acc2[j][1]; }
  float a0 = w.b41[0], a1 = w.b41[1], r0 = w.b42[0], r1 = w.b42[1], r2 = w.b42[2], r3 = w.b42[3];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float v = acc[c] > 0.f ? acc[c] : acc[c] * w.a3[c];
    a0 = fmaf(v, w.w41[c * 2 + 0], a0);
    a1 = fmaf(v, w.w41[c * 2 + 1], a1);
    r0 = fmaf(v, w.w42[c * 4 + 0], r0);
    r1 = fmaf(v, w.w42[c * 4 + 1], r1);
    r2 = fmaf(v, w.w42[c * 4 + 2], r2);
    r3 = fmaf(v, w.w42[c * 4 + 3], r3);
  }
  const float m = fmaxf(a0, a1);
  const float e0 = expf(a0 - m), e1 = expf(a1 - m);
  const float prob = e1 / (e0 + e1);
  if (prob_dbg) {
    prob_dbg[(size_t)img * t.tot_out + idx] = prob;
    float* rd = reg_dbg + ((size_t)img * 4) * t.tot_out + idx;
    rd[0] = r0; rd[(size_t)t.tot_out] = r1; rd[2 * (size_t)t.tot_out] = r2; rd[3 * (size_t)t.tot_out] = r3;
  }
  if (prob >= thr) {
    // the record goes to its cell's slot of a dense per-frame table, the cell index to the level's compact list (one
    // entry per cell at most: the list cannot overflow)
    const int slot = atomicAdd(&cand_cnt[li * B + img], 1);
    Cand c;
    c.score = prob; c.r0 = r0; c.r1 = r1; c.r2 = r2; c.r3 = r3; c.cell = p;
    cand[(size_t)img * cap_out + idx] = c;
    cells[(size_t)img * cap_out + L.off_out + slot] = p;
  }
}

__device__ __forceinline__ float4 cell_box(int cell, int ow, float scale) {
  // detect_face.py:214-216: stride 2, cellsize 12; fp32 division by the fp32-rounded scale
  const int y = cell / ow, x = cell - y * ow;
  const float fx = (float)x, fy = (float)y;
  return float4{floorf((2.f * fx + 1.f) / scale), floorf((2.f * fy + 1.f) / scale),
                floorf((2.f * fx + 12.f) / scale), floorf((2.f * fy + 12.f) / scale)};
}

// --------------------------------------------------------------------------------------------- K4a
// detect_face.py:79: batched_nms(..., 0.5) within each (scale, image).  Visiting order = stable
// score-descending over nonzero() order (y, x): key = (inverted score | cell | slot).
// global-memory fallback of the NMS kernels (per frame `stride` entries; a level's region starts at its off_out)
struct NmsScratch {
  unsigned long long* keys;
  float4* kbox;
  int* keep;
  float4* reg;
  int stride;
};

template <bool KEYS_G, bool KEPT_G>
__device__ __forceinline__ void nms_scale_body(const Cand* __restrict__ cd, const int* __restrict__ cl, int n, int ow, float scale,
                                               float thr, unsigned long long* keys, float4* kbox, int* keepl, float4* s_cbox,
                                               int* s_alive, int* __restrict__ out_cells, int* __restrict__ out_cnt, int* status) {
  const int npad = next_pow2(n);
  // visiting order = stable score-descending over nonzero() order (y, x): key = (inverted score | cell); cells are unique
  for (int i = threadIdx.x; i < npad; i += blockDim.x)
    keys[i] = i < n ? ((unsigned long long)inv_score_bits(cd[cl[i]].score) << 32) | (unsigned)cl[i] : ~0ull;
  __syncthreads();
  block_sort(keys, n, npad);
  auto getbox = [&](int r) { return cell_box((int)(keys[r] & 0xFFFFFFFFu), ow, scale); };
  const int nk = block_greedy_nms<NMS_TV>(n, thr, getbox, keepl, kbox, n, s_cbox, s_alive, status);
  for (int k = threadIdx.x; k < nk; k += blockDim.x) out_cells[k] = (int)(keys[keepl[k]] & 0xFFFFFFFFu);
  if (threadIdx.x == 0) *out_cnt = nk;
}

__global__ void __launch_bounds__(256) nms_scale_kernel(const Cand* __restrict__ cand, const int* __restrict__ cells,
                                                         const int* __restrict__ cand_cnt, LevelTable t, int B, int cap_out, float thr,
                                                         int* __restrict__ keep1c, int* __restrict__ keep_cnt, int* __restrict__ status,
                                                         NmsScratch g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);              // CAP_LDS_KEYS * 8
  float4* s_kbox = reinterpret_cast<float4*>(smem + CAP_LDS_KEYS * 8);                  // KEEP * 16
  int* s_keep = reinterpret_cast<int*>(smem + CAP_LDS_KEYS * 8 + KEEP * 16);            // KEEP * 4
  float4* s_cbox = reinterpret_cast<float4*>(smem + CAP_LDS_KEYS * 8 + KEEP * 20);      // 256 * 16
  int* s_alive = reinterpret_cast<int*>(smem + CAP_LDS_KEYS * 8 + KEEP * 20 + 256 * 16);
  const int li = blockIdx.x, img = blockIdx.y, seg = li * B + img;
  const int n = cand_cnt[seg];
  if (n == 0) {
    if (threadIdx.x == 0) keep_cnt[seg] = 0;
    return;
  }
  const LevelDesc& L = t.l[li];
  const size_t base = (size_t)img * cap_out + L.off_out;
  const Cand* cd = cand + base;
  const int* cl = cells + base;
  int* oc = keep1c + base;
  const size_t gb = (size_t)img * g.stride + L.off_out;
  // LDS while the list fits; a level with more candidates than the LDS tables hold takes global-memory scratch
  if (n <= KEEP) nms_scale_body<false, false>(cd, cl, n, L.ow, L.scale, thr, keys, s_kbox, s_keep, s_cbox, s_alive, oc, keep_cnt + seg, status);
  else if (n <= CAP_LDS_KEYS) nms_scale_body<false, true>(cd, cl, n, L.ow, L.scale, thr, keys, g.kbox + gb, g.keep + gb, s_cbox, s_alive, oc, keep_cnt + seg, status);
  else nms_scale_body<true, true>(cd, cl, n, L.ow, L.scale, thr, g.keys + gb, g.kbox + gb, g.keep + gb, s_cbox, s_alive, oc, keep_cnt + seg, status);
}

// --------------------------------------------------------------------------------------------- K4b
// detect_face.py:83-104: concatenate the per-scale survivors, batched_nms(..., 0.7) per image,
// regress with (w,h) WITHOUT +1, rerec (square), pad (trunc + clamp).  Table order = visiting
// order (score descending; ties: scale order, then within-scale order).
__device__ __forceinline__ void rerec_pad(float& x1, float& y1, float& x2, float& y2, int W, int H, Row& r) {
  const float h = y2 - y1, w = x2 - x1;
  const float l = fmaxf(w, h);
  x1 = x1 + w * 0.5f - l * 0.5f;
  y1 = y1 + h * 0.5f - l * 0.5f;
  x2 = x1 + l;
  y2 = y1 + l;
  int ix = (int)truncf(x1), iy = (int)truncf(y1), iex = (int)truncf(x2), iey = (int)truncf(y2);
  if (ix < 1) ix = 1;
  if (iy < 1) iy = 1;
  if (iex > W) iex = W;
  if (iey > H) iey = H;
  r.x1 = x1; r.y1 = y1; r.x2 = x2; r.y2 = y2;
  r.x = ix; r.y = iy; r.ex = iex; r.ey = iey;
}

template <bool KEYS_G, bool KEPT_G, typename Locate>
__device__ __forceinline__ void nms_image_body(Locate locate, const LevelTable& t, int n, float thr, int W, int H, int KR,
                                               unsigned long long* keys, float4* kbox, int* keepl, float4* s_cbox, int* s_alive,
                                               Row* __restrict__ rows_img, int* __restrict__ row_cnt_img, int* status) {
  const int npad = next_pow2(n);
  for (int i = threadIdx.x; i < npad; i += blockDim.x) {
    if (i < n) {
      int l;
      const Cand* c = locate(i, l);
      keys[i] = ((unsigned long long)inv_score_bits(c->score) << 32) | (unsigned)i;
    } else {
      keys[i] = ~0ull;
    }
  }
  __syncthreads();
  block_sort(keys, n, npad);
  auto getbox = [&](int r) {
    int l;
    const Cand* c = locate((int)(keys[r] & 0xFFFFFFFFu), l);
    return cell_box(c->cell, t.l[l].ow, t.l[l].scale);
  };
  int nk = block_greedy_nms<NMS_TV>(n, thr, getbox, keepl, kbox, n, s_cbox, s_alive, status);
  if (nk > KR) {     // more survivors than the stage-2 table has rows: the call fails (the host layer grows the table)
    if (threadIdx.x == 0) atomicOr(status, ST_OVER_KEEP);
    nk = KR;
  }
  for (int k = threadIdx.x; k < nk; k += blockDim.x) {
    int l;
    const Cand* c = locate((int)(keys[keepl[k]] & 0xFFFFFFFFu), l);
    const float4 b = kbox[k];
    const float regw = b.z - b.x, regh = b.w - b.y;
    float x1 = b.x + c->r0 * regw, y1 = b.y + c->r1 * regh, x2 = b.z + c->r2 * regw, y2 = b.w + c->r3 * regh;
    Row r;
    rerec_pad(x1, y1, x2, y2, W, H, r);
    r.score = c->score;
    rows_img[k] = r;
  }
  if (threadIdx.x == 0) *row_cnt_img = nk;
}

__global__ void __launch_bounds__(256) nms_image_kernel(const Cand* __restrict__ cand, const int* __restrict__ keep1c,
                                                         const int* __restrict__ keep1_cnt, LevelTable t, int B, int cap_out,
                                                         float thr, int W, int H, int KR, Row* __restrict__ rows,
                                                         int* __restrict__ row_cnt, int* __restrict__ status, NmsScratch g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);           // CAP_LDS_KEYS * 8
  float4* s_kbox = reinterpret_cast<float4*>(smem + CAP_LDS_KEYS * 8);
  int* s_keep = reinterpret_cast<int*>(smem + CAP_LDS_KEYS * 8 + KEEP * 16);
  float4* s_cbox = reinterpret_cast<float4*>(smem + CAP_LDS_KEYS * 8 + KEEP * 20);
  int* s_alive = reinterpret_cast<int*>(smem + CAP_LDS_KEYS * 8 + KEEP * 20 + 256 * 16);
  __shared__ int s_off[MAX_LEVELS + 1];
  const int img = blockIdx.x;
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int l = 0; l < t.n; ++l) { s_off[l] = acc; acc += keep1_cnt[l * B + img]; }
    s_off[t.n] = acc;
  }
  __syncthreads();
  const int n = s_off[t.n];
  if (n == 0) {
    if (threadIdx.x == 0) row_cnt[img] = 0;
    return;
  }
  const size_t base = (size_t)img * cap_out;
  auto locate = [&](int gi, int& l) -> const Cand* {  // gathered index -> record (per-scale survivors are lists of cells)
    l = 0;
    for (int i = 1; i < t.n; ++i)
      if (gi >= s_off[i]) l = i;
    return cand + base + t.l[l].off_out + keep1c[base + t.l[l].off_out + (gi - s_off[l])];
  };
  Row* ro = rows + (size_t)img * KR;
  const size_t gb = (size_t)img * g.stride;
  if (n <= KEEP) nms_image_body<false, false>(locate, t, n, thr, W, H, KR, keys, s_kbox, s_keep, s_cbox, s_alive, ro, row_cnt + img, status);
  else if (n <= CAP_LDS_KEYS) nms_image_body<false, true>(locate, t, n, thr, W, H, KR, keys, g.kbox + gb, g.keep + gb, s_cbox, s_alive, ro, row_cnt + img, status);
  else nms_image_body<true, true>(locate, t, n, thr, W, H, KR, g.keys + gb, g.kbox + gb, g.keep + gb, s_cbox, s_alive, ro, row_cnt + img, status);
}

// --------------------------------------------------------------------------------------------- K5
// detect_face.py:109-114 / 138-143: imgs[i, :, y-1:ey, x-1:ex] -> imresample(S,S) -> normalise.
// One workgroup per candidate; output NCHW fp32 (3,S,S).  Degenerate rectangles (the reference
// silently drops them from im_data, which would desynchronise its tables) are flagged and zeroed.
// Output addressing of the crop kernels.  Planar: (3,S,S) per candidate at [img][KEEP] (the LDS-resident
// nets).  Compact NHWC4: candidate offs[img]+k-c0 of a dense batch, 4 floats per pixel (RGB + 0), the
// input layout of the MFMA R/O-Net plans; candidates outside [c0, c0+cap) are skipped.
struct CropDst {
  float* base;      // nullptr: candidate not in this chunk
  int cs, ps;       // channel stride, pixel stride (floats)
};
__device__ __forceinline__ CropDst crop_dst(float* out, const int* offs, int c0, int cap, int img, int k, int S, int KR) {
  if (!offs) return CropDst{out + ((size_t)img * KR + k) * 3 * S * S, S * S, 1};
  const int ci = offs[img] + k - c0;
  if (ci < 0 || ci >= cap) return CropDst{nullptr, 0, 0};
  return CropDst{out + (size_t)ci * S * S * 4, 1, 4};
}

__global__ void __launch_bounds__(256) crop_resize_kernel(const uint8_t* __restrict__ frames, int H, int W,
                                                           const Row* __restrict__ rows, const int* __restrict__ row_cnt,
                                                           int S, float* __restrict__ out, int* __restrict__ status,
                                                           const int* __restrict__ offs, int c0, int cap, int KR) {
  const int k = blockIdx.x, img = blockIdx.y;
  if (k >= row_cnt[img]) return;
  const CropDst d = crop_dst(out, offs, c0, cap, img, k, S, KR);
  if (!d.base) return;
  const Row r = rows[(size_t)img * KR + k];
  const int y0 = r.y - 1, x0 = r.x - 1, ch = r.ey - y0, cw = r.ex - x0;
  float* o = d.base;
  if (d.ps == 4 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) o[i * 4 + 3] = 0.f;
  if ((ch <= 0 || cw <= 0) && blockIdx.z != 0) return;
  if (ch <= 0 || cw <= 0) {
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) { o[i * d.ps] = 0.f; o[i * d.ps + d.cs] = 0.f; o[i * d.ps + 2 * d.cs] = 0.f; }
    if (threadIdx.x == 0) atomicOr(status, ST_DEGENERATE);
    return;
  }
  const uint8_t* base = frames + ((size_t)img * H + y0) * (size_t)W * 3 + (size_t)x0 * 3;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
    const int oy = i / S, ox = i - oy * S;
    const int h0 = (oy * ch) / S, h1 = ((oy + 1) * ch + S - 1) / S;
    const int w0 = (ox * cw) / S, w1 = ((ox + 1) * cw + S - 1) / S;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int yy = h0; yy < h1; ++yy) {
      const uint8_t* row = base + ((size_t)yy * W + w0) * 3;
      for (int xx = 0; xx < w1 - w0; ++xx) {
        s0 += (float)row[3 * xx]; s1 += (float)row[3 * xx + 1]; s2 += (float)row[3 * xx + 2];
      }
    }
    const float kh = (float)(h1 - h0), kw = (float)(w1 - w0);
    o[i * d.ps] = ((s0 / kh) / kw - 127.5f) * 0.0078125f;
    o[i * d.ps + d.cs] = ((s1 / kh) / kw - 127.5f) * 0.0078125f;
    o[i * d.ps + 2 * d.cs] = ((s2 / kh) / kw - 127.5f) * 0.0078125f;
  }
}

// Fast path of K5 (rows of W*3 % 16 == 0, crops up to CROP_MAXB bytes wide): each wave owns output
// rows wave, wave+4, ...; its lanes stream the 16-byte chunks that cover the crop's byte span of
// every input row of the bin row (coalesced dwordx4 loads instead of per-pixel byte loads), keep
// per-byte column sums in registers, park them in a wave-private LDS strip, and then add the
// horizontal bin spans.  Integer sums: bit-identical to the scalar kernel and to the oracle.
constexpr int CROP_MAXB = 4096;  // bytes of crop row per strip (1365 px)

__global__ void __launch_bounds__(256) crop_resize_rows_kernel(const uint8_t* __restrict__ frames, int H, int W,
                                                                const Row* __restrict__ rows, const int* __restrict__ row_cnt,
                                                                int S, float* __restrict__ out, int* __restrict__ status,
                                                                const int* __restrict__ offs, int c0, int cap, int KR) {
  // per-byte column sums of one bin row are at most (rows of a bin) x 255: 16 bits hold 257 rows, deeper bins (frames
  // taller than ~6000 px at S = 24) take the per-pixel path.  Half the LDS of 32-bit sums -> twice the resident waves.
  __shared__ __attribute__((aligned(16))) unsigned short strips[4][CROP_MAXB + 32];
  const int k = blockIdx.x, img = blockIdx.y;
  if (k >= row_cnt[img]) return;
  const CropDst d = crop_dst(out, offs, c0, cap, img, k, S, KR);
  if (!d.base) return;
  const Row r = rows[(size_t)img * KR + k];
  const int y0 = r.y - 1, x0 = r.x - 1, ch = r.ey - y0, cw = r.ex - x0;
  float* o = d.base;
  if (d.ps == 4 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) o[i * 4 + 3] = 0.f;
  if ((ch <= 0 || cw <= 0) && blockIdx.z != 0) return;
  if (ch <= 0 || cw <= 0) {
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) { o[i * d.ps] = 0.f; o[i * d.ps + d.cs] = 0.f; o[i * d.ps + 2 * d.cs] = 0.f; }
    if (threadIdx.x == 0) atomicOr(status, ST_DEGENERATE);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowb = W * 3, bs = x0 * 3, be = (x0 + cw) * 3;
  const int c_lo = bs >> 4, nch = ((be + 15) >> 4) - c_lo, off = bs - (c_lo << 4);
  const bool slow = nch * 16 > CROP_MAXB + 32 || (ch + S - 1) / S + 1 > 257;
  if (slow && blockIdx.z != 0) return;
  if (slow) {  // wider than a strip (or bins too deep for 16-bit sums): per-pixel path for this candidate
    const uint8_t* base = frames + ((size_t)img * H + y0) * (size_t)W * 3 + (size_t)x0 * 3;
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
      const int oy = i / S, ox = i - oy * S;
      const int h0 = (oy * ch) / S, h1 = ((oy + 1) * ch + S - 1) / S;
      const int w0 = (ox * cw) / S, w1 = ((ox + 1) * cw + S - 1) / S;
      unsigned s0 = 0, s1 = 0, s2 = 0;
      for (int yy = h0; yy < h1; ++yy) {
        const uint8_t* row = base + ((size_t)yy * W + w0) * 3;
        for (int xx = 0; xx < w1 - w0; ++xx) { s0 += row[3 * xx]; s1 += row[3 * xx + 1]; s2 += row[3 * xx + 2]; }
      }
      const float kh = (float)(h1 - h0), kw = (float)(w1 - w0);
      o[i * d.ps] = (((float)s0 / kh) / kw - 127.5f) * 0.0078125f;
      o[i * d.ps + d.cs] = (((float)s1 / kh) / kw - 127.5f) * 0.0078125f;
      o[i * d.ps + 2 * d.cs] = (((float)s2 / kh) / kw - 127.5f) * 0.0078125f;
    }
    return;
  }
  const uint8_t* fbase = frames + (size_t)img * H * rowb;
  unsigned short* cs = strips[wave];
  // blockIdx.z splits the S output rows into gridDim.z groups, so one large box (its bins are tens of input rows
  // deep) is spread over several workgroups instead of setting the duration of the whole launch
  const int zrows = (S + (int)gridDim.z - 1) / (int)gridDim.z;
  const int oy_lo = (int)blockIdx.z * zrows, oy_hi = min(S, oy_lo + zrows);
  // per-byte column sums of input rows [h0,h1) of one 16-byte chunk column: four independent loads in flight per
  // step (clamped row + byte mask instead of a branch, so the loads are not serialised behind their predicates)
  auto colsum = [&](int c, int h0, int h1, unsigned (&acc)[16]) {
    // bytes 0,2 / 1,3 of each dword accumulate in the two 16-bit halves of one register (a bin has at most 257 rows
    // here: deeper ones took the per-pixel path above), unpacked once at the end
    unsigned pe[4] = {0u, 0u, 0u, 0u}, po[4] = {0u, 0u, 0u, 0u};
    const uint8_t* p0 = fbase + (size_t)y0 * rowb + ((size_t)(c_lo + c) << 4);
    for (int yy = h0; yy < h1; yy += 4) {
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const uint4*>(p0 + (size_t)min(yy + j, h1 - 1) * rowb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned msk = (yy + j < h1) ? 0x00FF00FFu : 0u;
        const unsigned wv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          pe[d] += wv[d] & msk;
          po[d] += (wv[d] >> 8) & msk;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      acc[d * 4 + 0] = pe[d] & 0xFFFFu;
      acc[d * 4 + 1] = po[d] & 0xFFFFu;
      acc[d * 4 + 2] = pe[d] >> 16;
      acc[d * 4 + 3] = po[d] >> 16;
    }
  };
  if (nch <= 32) {
    // narrow crops (the common case: a 100-px box spans ~20 chunks): a wave takes R = 64/nch output rows at
    // once, lane -> (row sub, chunk c), so the loads keep the whole wave busy
    // R rows per wave and pass, but no more than spreads the group's rows over the four waves
    const int R = min(64 / nch, (oy_hi - oy_lo + 3) / 4), sub = lane / nch, c = lane - sub * nch;
    for (int oy0 = oy_lo + wave * R; oy0 < oy_hi; oy0 += 4 * R) {
      const int oy = oy0 + sub;
      if (sub < R && oy < oy_hi) {
        unsigned acc[16];
        colsum(c, (oy * ch) / S, ((oy + 1) * ch + S - 1) / S, acc);
        uint4* dst = reinterpret_cast<uint4*>(cs + (sub * nch + c) * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          dst[j] = uint4{acc[8 * j] | (acc[8 * j + 1] << 16), acc[8 * j + 2] | (acc[8 * j + 3] << 16),
                         acc[8 * j + 4] | (acc[8 * j + 5] << 16), acc[8 * j + 6] | (acc[8 * j + 7] << 16)};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int q = lane; q < R * 3 * S; q += 64) {
        const int s2 = q / (3 * S), q2 = q - s2 * 3 * S;
        const int oy2 = oy0 + s2;
        if (oy2 < oy_hi) {
          const int cch = q2 / S, ox = q2 - cch * S;
          const int h0 = (oy2 * ch) / S, h1 = ((oy2 + 1) * ch + S - 1) / S;
          const int w0 = (ox * cw) / S, w1 = ((ox + 1) * cw + S - 1) / S;
          const unsigned short* row = cs + s2 * nch * 16 + off;
          unsigned sum = 0;
          for (int xx = w0; xx < w1; ++xx) sum += row[xx * 3 + cch];
          o[(oy2 * S + ox) * d.ps + cch * d.cs] = (((float)sum / (float)(h1 - h0)) / (float)(w1 - w0) - 127.5f) * 0.0078125f;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  for (int oy = oy_lo + wave; oy < oy_hi; oy += 4) {
    const int h0 = (oy * ch) / S, h1 = ((oy + 1) * ch + S - 1) / S;
    for (int cbase = 0; cbase < nch; cbase += 64) {
      const int c = cbase + lane;
      if (c < nch) {
        unsigned acc[16];
        colsum(c, h0, h1, acc);
        uint4* dst = reinterpret_cast<uint4*>(cs + c * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          dst[j] = uint4{acc[8 * j] | (acc[8 * j + 1] << 16), acc[8 * j + 2] | (acc[8 * j + 3] << 16),
                         acc[8 * j + 4] | (acc[8 * j + 5] << 16), acc[8 * j + 6] | (acc[8 * j + 7] << 16)};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float kh = (float)(h1 - h0);
    for (int q = lane; q < 3 * S; q += 64) {
      const int cch = q / S, ox = q - cch * S;
      const int w0 = (ox * cw) / S, w1 = ((ox + 1) * cw + S - 1) / S;
      unsigned sum = 0;
      for (int xx = w0; xx < w1; ++xx) sum += cs[off + xx * 3 + cch];
      o[(oy * S + ox) * d.ps + cch * d.cs] = (((float)sum / kh) / (float)(w1 - w0) - 127.5f) * 0.0078125f;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// --------------------------------------------------------------------------------------------- K6a
// R-Net / O-Net front: conv1 (3 -> 28 / 32, 3x3) + PReLU + MaxPool(3, 2, ceil_mode) in one kernel (mtcnn.py:84-87 /
// 138-141).  The conv1 map is the largest tensor of the cascade (46x46x32 floats per O-Net candidate, 208 MB for 767
// candidates); here it only ever exists in LDS.  One workgroup per (band of BANDP pooled rows, candidate): the crop
// rows the band needs go to LDS, every wave computes all 32 output channels of 16 conv pixels per round on the exact-fp32 MFMA (weights stay
// in 18 registers per lane), the PReLU outputs are parked in LDS as
// [pixel][32] with the 16-byte chunk index XOR-swizzled by the pixel, and the pooled rows are reduced from there and
// written as whole NHWC rows.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct FrontW { const float* w; const float* b; const float* a; };   // [32][9 taps][4 channels (3 + zero)], [32], [32]

// SPLIT: the pooled map is written as split-f16 (hi, lo) pairs, the storage of the F16X2 plans (split_f16.h).
__device__ __forceinline__ float split_pack(float v) {
  const sf16 h(v);
  return __builtin_bit_cast(float, h);
}
// MM16 (with SPLIT): conv1 itself on the 16-bit MFMA with split-f16 operands -- the crop pixels are split once when
// they are copied to LDS (a pixel's 3 + 1 channels as four (hi, lo) pairs are the same 16 bytes as its four floats), a
// k block is four taps x four channels, so the 9 taps are three MFMA pairs per 16-channel tile (96 cycles) instead of
// nine f32 MFMAs (288); ~22 significant bits per operand like every later layer of the split plans.
template <int S, int BANDP, int NT, bool SPLIT, bool MM16 = false>
__global__ void __launch_bounds__(NT) net_front_kernel(const float* __restrict__ crops, FrontW fw, float* __restrict__ p1) {
  constexpr int C = S - 2;                 // conv1 rows / cols
  constexpr int P = (C - 3 + 1) / 2 + 1;   // ceil((C - 3) / 2) + 1
  constexpr int CR = 2 * BANDP + 1;        // conv rows of a band
  constexpr int IR = CR + 2;               // crop rows of a band
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* s_in = reinterpret_cast<float4*>(smem);                    // [IR][S] NHWC4
  float4* s_cv = reinterpret_cast<float4*>(smem) + IR * S;           // [CR * C][8 chunks]
  const int band = blockIdx.x, cand = blockIdx.y, t = threadIdx.x;
  const int p0 = band * BANDP, np = min(BANDP, P - p0);
  const int cr0 = 2 * p0, ncr = min(CR, C - cr0), nir = ncr + 2;
  const float4* src = reinterpret_cast<const float4*>(crops) + ((size_t)cand * S + cr0) * S;
  for (int i = t; i < nir * S; i += NT) {
    float4 v = src[i];
    if constexpr (MM16) v = float4{split_pack(v.x), split_pack(v.y), split_pack(v.z), split_pack(v.w)};
    s_in[i] = v;
  }
  __syncthreads();
  // conv1 as 16x16x4 fp32 MFMAs, one per (tap, 16-channel tile): A = weights (lane: channel l&15, input channel l>>4),
  // B = crop pixels (lane: pixel l&15, input channel l>>4; channel 3 is the zero pad), D = 4 consecutive output
  // channels of one pixel per lane.  Same k order as the plan's implicit-GEMM conv (tap-major), bias after the sum.
  const int wave = t >> 6, lane = t & 63, lg = lane >> 4, lm = lane & 15;
  float wa[2][MM16 ? 1 : 9];
  uint4 ws[2][MM16 ? 3 : 1];   // MM16: A fragments, lane (channel lm, group lg) = tap 4 blk + lg, input channels 0..3 as (hi, lo) pairs
  if constexpr (MM16) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int blk = 0; blk < 3; ++blk) {
        const int tap = 4 * blk + lg;
        float4 w4 = float4{0.f, 0.f, 0.f, 0.f};
        if (tap < 9) w4 = *reinterpret_cast<const float4*>(fw.w + ((ct * 16 + lm) * 9 + tap) * 4);
        const float4 sp = float4{split_pack(w4.x), split_pack(w4.y), split_pack(w4.z), split_pack(w4.w)};
        ws[ct][blk] = __builtin_bit_cast(uint4, sp);
      }
  } else {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wa[ct][tap] = fw.w[((ct * 16 + lm) * 9 + tap) * 4 + lg];
  }
  float bias[2][4], slope[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias[ct][e] = fw.b[ct * 16 + lg * 4 + e]; slope[ct][e] = fw.a[ct * 16 + lg * 4 + e]; }
  const int npx = ncr * C;
  const float* s_inf = reinterpret_cast<const float*>(s_in);
  for (int tile = wave; tile * 16 < npx; tile += NT / 64) {
    const int px = tile * 16 + lm, pxc = min(px, npx - 1);
    const int r = pxc / C, x = pxc - r * C;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    if constexpr (MM16) {
      typedef _Float16 f16x8f_t __attribute__((ext_vector_type(8)));
      const uint4* s_inu = reinterpret_cast<const uint4*>(s_in);
#pragma unroll
      for (int blk = 0; blk < 3; ++blk) {
        const int tap = 4 * blk + lg;
        uint4 xf = uint4{0u, 0u, 0u, 0u};
        if (tap < 9) xf = s_inu[(r + tap / 3) * S + x + tap % 3];
        const uint4 xr = {(xf.x >> 16) | (xf.x << 16), (xf.y >> 16) | (xf.y << 16), (xf.z >> 16) | (xf.z << 16),
                          (xf.w >> 16) | (xf.w << 16)};
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8f_t, ws[ct][blk]), __builtin_bit_cast(f16x8f_t, xf), acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8f_t, ws[ct][blk]), __builtin_bit_cast(f16x8f_t, xr), acc[ct], 0, 0, 0);
        }
      }
    } else {
      float xb[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xb[kh * 3 + kw] = s_inf[((r + kh) * S + x + kw) * 4 + lg];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][tap], xb[tap], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][tap], xb[tap], acc[1], 0, 0, 0);
      }
    }
    if (px < npx) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float4 o;
        float* op = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[ct][e] + bias[ct][e];
          op[e] = v > 0.f ? v : v * slope[ct][e];
        }
        s_cv[px * 8 + ((ct * 4 + lg) ^ (px & 7))] = o;
      }
    }
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(p1) + ((size_t)cand * P + p0) * P * 8;
  for (int i = t; i < np * P * 8; i += NT) {
    const int q = i & 7, pp = i >> 3, py = pp / P, pxx = pp - py * P;
    float4 m = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int rr = 2 * py + dy, xx = 2 * pxx + dx;      // band-local conv row, conv col
        if (rr < ncr && xx < C) {
          const int px = rr * C + xx;
          const float4 v = s_cv[px * 8 + (q ^ (px & 7))];
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    if (SPLIT) m = float4{split_pack(m.x), split_pack(m.y), split_pack(m.z), split_pack(m.w)};
    dst[i] = m;
  }
}

// --------------------------------------------------------------------------------------------- K6b
// R-Net / O-Net conv2 (32 -> 48 / 64, 3x3) + PReLU + MaxPool(3, 2, ceil_mode) in one kernel (mtcnn.py:88-90 / 142-144)
// for the split-f16 plans: one workgroup per candidate, the pooled conv1 map (net_front_kernel's output: [pixel][32
// channels] of (hi, lo) pairs) in LDS, the convolution as 16x16x32 f16 MFMAs on the interleaved split operands -- per
// 16 k values (one tap, 16 channels) the chunk pair and the pair with the activation's halves swapped, exactly
// mma_chunk<sf16>, in the plan's k order (tap-major) with its fp32 sum, bias and PReLU -- weights as 18 A-fragments per
// 16-channel tile in REGISTERS, the conv map ([pixel][channels of the pass] fp32) only ever in LDS, pooled from there
// in fp32 and written as split-f16 NHWC rows for the plan's conv3 (the plan splits the map first and pools the split
// values: the same up to the split format's 2^-22 rounding; detections agree to 1e-3 px / 1e-6, tested).  The unfused plan wrote and re-read that map (O-Net: 99 MB
// per 880 candidates) and paid two launches: 0.089 + 0.029 ms (O-Net), 0.057 + 0.017 ms (R-Net) per 16 frames.
typedef _Float16 f16x8m_t __attribute__((ext_vector_type(8)));
struct MidW { const uint4* w; const float* b; const float* a; };   // [CO/16][18][64 lanes] fragments, [CO], [CO]

template <int PI, int CO, int NWAVE, int NPASS>
__global__ void __launch_bounds__(NWAVE * 64) net_mid_kernel(const float* __restrict__ p1, MidW mw, float* __restrict__ p2) {
  constexpr int C = PI - 2, NPX = C * C, PO = (C - 3 + 1) / 2 + 1;
  constexpr int NCT = CO / 16, CTP = NCT / NPASS, NMG = NWAVE / CTP, NMT = (NPX + 15) / 16;
  constexpr int CHP = CTP * 4;   // 16-byte fp32 chunks per conv pixel in one pass
  static_assert(NCT % NPASS == 0 && NWAVE % CTP == 0, "channel tiles divide over passes and waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_in = reinterpret_cast<uint4*>(smem);                       // [PI * PI][8 chunks of 4 (hi, lo) channels], chunk ^ (px & 7)
  float4* s_cv = reinterpret_cast<float4*>(smem + PI * PI * 128);    // [NPX][CHP]
  const int cand = blockIdx.x, t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63, frow = lane & 15, g = lane >> 4;
  const int ctl = wave % CTP, mg = wave / CTP;
  // the first pass's weight fragments travel while the input map is copied to LDS; the next pass's while this one pools
  uint4 wf[18];
#pragma unroll
  for (int kb = 0; kb < 18; ++kb) wf[kb] = mw.w[((size_t)ctl * 18 + kb) * 64 + lane];
  {
    const uint4* src = reinterpret_cast<const uint4*>(p1) + (size_t)cand * PI * PI * 8;
    for (int i = t; i < PI * PI * 8; i += NWAVE * 64) {
      const int q = i >> 3, ch = i & 7;
      s_in[q * 8 + (ch ^ (q & 7))] = src[i];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    const int ct = pass * CTP + ctl;
    float bias[4], slope[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias[e] = mw.b[ct * 16 + 4 * g + e]; slope[e] = mw.a[ct * 16 + 4 * g + e]; }
    for (int tile = mg; tile < NMT; tile += NMG) {
      const int px = tile * 16 + frow, pxc = min(px, NPX - 1);
      const int oy = pxc / C, ox = pxc - oy * C;
      const int q0 = oy * PI + ox;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int q = q0 + (tap / 3) * PI + tap % 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint4 xf = s_in[q * 8 + ((4 * h + g) ^ (q & 7))];
          const uint4 xr = {(xf.x >> 16) | (xf.x << 16), (xf.y >> 16) | (xf.y << 16), (xf.z >> 16) | (xf.z << 16),
                            (xf.w >> 16) | (xf.w << 16)};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m_t, wf[2 * tap + h]),
                                                       __builtin_bit_cast(f16x8m_t, xf), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m_t, wf[2 * tap + h]),
                                                       __builtin_bit_cast(f16x8m_t, xr), acc, 0, 0, 0);
        }
      }
      if (px < NPX) {
        float4 o;
        float* op = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[e] + bias[e];
          op[e] = v > 0.f ? v : v * slope[e];
        }
        s_cv[px * CHP + ctl * 4 + g] = o;
      }
    }
    if (pass + 1 < NPASS) {
#pragma unroll
      for (int kb = 0; kb < 18; ++kb) wf[kb] = mw.w[((size_t)(ct + CTP) * 18 + kb) * 64 + lane];
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(p2) + (size_t)cand * PO * PO * (CO / 4) + pass * CHP;
    for (int i = t; i < PO * PO * CHP; i += NWAVE * 64) {
      const int chunk = i % CHP, pp = i / CHP, py = pp / PO, pxx = pp - py * PO;
      float4 m = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int rr = 2 * py + dy, xx = 2 * pxx + dx;
          if (rr < C && xx < C) {
            const float4 v = s_cv[(rr * C + xx) * CHP + chunk];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
          }
        }
      dst[(size_t)pp * (CO / 4) + chunk] = float4{split_pack(m.x), split_pack(m.y), split_pack(m.z), split_pack(m.w)};
    }
    if (pass + 1 < NPASS) __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------- K6 building blocks
// Direct convolution / pooling / dense layers over activations resident in LDS (CHW fp32),
// weights in their PyTorch layout read through L1/L2 (shared by every workgroup).
__device__ __forceinline__ float prelu(float v, float a) { return v > 0.f ? v : v * a; }

// Register-tiled direct convolution: a thread computes COB output channels x PXB consecutive pixels,
// so each LDS input value feeds COB*KS FMAs and each weight PXB FMAs.  wt is the layer's weight
// transposed to [CIN][KS][KS][ldw] (output channel fastest), so the COB weights of one tap are one
// or two 16-byte loads, identical across the lanes that share a channel block (L1 broadcast).
// Per output the FMA order is (c, kh, kw), as in the reference's direct statement of the conv.
template <int CIN, int KS, int COB, int PXB>
__device__ void lds_conv_prelu(const float* __restrict__ in, int Hi, int Wi, float* __restrict__ out, int cout,
                               const float* __restrict__ wt, int ldw, const float* __restrict__ b,
                               const float* __restrict__ a) {
  const int Ho = Hi - KS + 1, Wo = Wi - KS + 1;
  const int xg = (Wo + PXB - 1) / PXB, per_cb = Ho * xg, n = (cout / COB) * per_cb;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int cb = i / per_cb, p = i - cb * per_cb, y = p / xg, x0 = (p - y * xg) * PXB;
    float acc[COB][PXB];
#pragma unroll
    for (int co = 0; co < COB; ++co)
#pragma unroll
      for (int px = 0; px < PXB; ++px) acc[co][px] = b[cb * COB + co];
#pragma unroll 1
    for (int c = 0; c < CIN; ++c) {
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const float* ir = in + c * Hi * Wi + (y + kh) * Wi + x0;
        float v[PXB + KS - 1];
#pragma unroll
        for (int j = 0; j < PXB + KS - 1; ++j) v[j] = (x0 + j < Wi) ? ir[j] : 0.f;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const float* ww = wt + (size_t)((c * KS + kh) * KS + kw) * ldw + cb * COB;
#pragma unroll
          for (int co = 0; co < COB; ++co)
#pragma unroll
            for (int px = 0; px < PXB; ++px) acc[co][px] = fmaf(v[px + kw], ww[co], acc[co][px]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COB; ++co)
#pragma unroll
      for (int px = 0; px < PXB; ++px)
        if (x0 + px < Wo) out[(cb * COB + co) * Ho * Wo + y * Wo + x0 + px] = prelu(acc[co][px], a[cb * COB + co]);
  }
}

// The same register-tiled convolution with the weights of ONE input channel at a time staged in
// LDS (double-buffered, one barrier per input channel): the inner loop then touches no global
// memory, which matters at one workgroup per CU where nothing else hides an L2 round trip.
// Every thread owns NTILE output tiles for the whole channel loop.  wbuf: 2 * KS*KS * cout floats.
template <int CIN, int KS, int COB, int PXB, int NTILE>
__device__ void lds_conv_prelu_ws(const float* __restrict__ in, int Hi, int Wi, float* __restrict__ out, int cout,
                                  const float* __restrict__ wt, int ldw, const float* __restrict__ b,
                                  const float* __restrict__ a, float* __restrict__ wbuf) {
  const int Ho = Hi - KS + 1, Wo = Wi - KS + 1;
  const int xg = (Wo + PXB - 1) / PXB, per_cb = Ho * xg, n = (cout / COB) * per_cb;
  const int wn = KS * KS * cout;  // floats of one input channel's weights
  float acc[NTILE][COB][PXB];
  int tcb[NTILE], ty[NTILE], tx[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int i = threadIdx.x + t * blockDim.x;
    const int ii = i < n ? i : 0;
    tcb[t] = ii / per_cb;
    const int p = ii - tcb[t] * per_cb;
    ty[t] = p / xg;
    tx[t] = (p - ty[t] * xg) * PXB;
    if (i >= n) tcb[t] = -1;
#pragma unroll
    for (int co = 0; co < COB; ++co)
#pragma unroll
      for (int px = 0; px < PXB; ++px) acc[t][co][px] = tcb[t] >= 0 ? b[tcb[t] * COB + co] : 0.f;
  }
  // stage channel 0
  for (int i = threadIdx.x; i < wn; i += blockDim.x) wbuf[i] = wt[(size_t)(i / cout) * ldw + (i % cout)];
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < CIN; ++c) {
    float* wcur = wbuf + (c & 1) * wn;
    float* wnxt = wbuf + ((c + 1) & 1) * wn;
    // prefetch the next channel's weights into registers (<= 3 per thread), park them after the math
    float pre[3];
    const bool more = c + 1 < CIN;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      pre[u] = (more && i < wn) ? wt[(size_t)((c + 1) * KS * KS + i / cout) * ldw + (i % cout)] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      if (tcb[t] >= 0) {
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          const float* ir = in + c * Hi * Wi + (ty[t] + kh) * Wi + tx[t];
          float v[PXB + KS - 1];
#pragma unroll
          for (int j = 0; j < PXB + KS - 1; ++j) v[j] = (tx[t] + j < Wi) ? ir[j] : 0.f;
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const float* ww = wcur + (kh * KS + kw) * cout + tcb[t] * COB;
#pragma unroll
            for (int co = 0; co < COB; ++co)
#pragma unroll
              for (int px = 0; px < PXB; ++px) acc[t][co][px] = fmaf(v[px + kw], ww[co], acc[t][co][px]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      if (more && i < wn) wnxt[i] = pre[u];
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
    if (tcb[t] >= 0) {
#pragma unroll
      for (int co = 0; co < COB; ++co)
#pragma unroll
        for (int px = 0; px < PXB; ++px)
          if (tx[t] + px < Wo)
            out[(tcb[t] * COB + co) * Ho * Wo + ty[t] * Wo + tx[t] + px] = prelu(acc[t][co][px], a[tcb[t] * COB + co]);
    }
}

// conv + PReLU + MaxPool(PK, 2, ceil_mode=True) fused (recomputes the conv under overlapping windows):
// used where the un-pooled map would not fit LDS.
template <int CIN, int KS, int PK>
__device__ void lds_conv_prelu_pool(const float* __restrict__ in, int Hi, int Wi, float* __restrict__ out, int cout,
                                    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ a) {
  const int Hc = Hi - KS + 1, Wc = Wi - KS + 1;
  const int Hp = (Hc - PK + 1) / 2 + 1, Wp = (Wc - PK + 1) / 2 + 1;  // ceil((Hc-PK)/2)+1
  const int n = cout * Hp * Wp;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int co = i / (Hp * Wp), p = i - co * (Hp * Wp), py = p / Wp, px = p - py * Wp;
    const float* wc = w + (size_t)co * CIN * KS * KS;
    float best = -INFINITY;
    for (int oy = 0; oy < PK; ++oy)
      for (int ox = 0; ox < PK; ++ox) {
        const int y = 2 * py + oy, x = 2 * px + ox;
        if (y < Hc && x < Wc) {
          float acc = b[co];
#pragma unroll 1
          for (int c = 0; c < CIN; ++c) {
            const float* ic = in + c * Hi * Wi + y * Wi + x;
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
              for (int kw = 0; kw < KS; ++kw) acc = fmaf(ic[kh * Wi + kw], wc[(c * KS + kh) * KS + kw], acc);
          }
          best = fmaxf(best, prelu(acc, a[co]));
        }
      }
    out[i] = best;
  }
}

template <int PK>
__device__ void lds_maxpool_ceil(const float* __restrict__ in, int C, int Hi, int Wi, float* __restrict__ out) {
  const int Hp = (Hi - PK + 1) / 2 + 1, Wp = (Wi - PK + 1) / 2 + 1;
  for (int i = threadIdx.x; i < C * Hp * Wp; i += blockDim.x) {
    const int c = i / (Hp * Wp), p = i - c * (Hp * Wp), py = p / Wp, px = p - py * Wp;
    float best = -INFINITY;
    for (int oy = 0; oy < PK; ++oy)
      for (int ox = 0; ox < PK; ++ox) {
        const int y = 2 * py + oy, x = 2 * px + ox;
        if (y < Hi && x < Wi) best = fmaxf(best, in[c * Hi * Wi + y * Wi + x]);
      }
    out[i] = best;
  }
}

// dense layer on x.permute(0,3,2,1) flattened (mtcnn.py:93-94,150-151): feature f = (w*H + h)*C + c.
// `feat` receives the permuted input (nin floats, LDS); wt is the weight transposed to [nin][nout]
// so consecutive threads read consecutive outputs (coalesced).  With more threads than outputs the
// reduction is split over KSPLIT thread groups and combined through `part` (LDS, KSPLIT*nout floats).
__device__ void lds_dense_permuted_prelu(const float* __restrict__ in, int C, int Hh, int Ww, float* __restrict__ feat,
                                         float* __restrict__ part, float* __restrict__ out, int nout,
                                         const float* __restrict__ wt, const float* __restrict__ b,
                                         const float* __restrict__ a) {
  const int nin = C * Hh * Ww;
  for (int f = threadIdx.x; f < nin; f += blockDim.x) {
    const int c = f % C, hw = f / C, h = hw % Hh, w_ = hw / Hh;
    feat[f] = in[c * Hh * Ww + h * Ww + w_];
  }
  __syncthreads();
  const int ksplit = blockDim.x / nout;  // >= 1 (nout <= blockDim)
  const int o = threadIdx.x % nout, g = threadIdx.x / nout;
  if (g < ksplit) {
    const int f0 = (nin * g) / ksplit, f1 = (nin * (g + 1)) / ksplit;
    float acc = 0.f;
    for (int f = f0; f < f1; ++f) acc = fmaf(feat[f], wt[(size_t)f * nout + o], acc);
    part[g * nout + o] = acc;
  }
  __syncthreads();
  if (threadIdx.x < nout) {
    float acc = b[threadIdx.x];
    for (int g2 = 0; g2 < ksplit; ++g2) acc += part[g2 * nout + threadIdx.x];
    out[threadIdx.x] = prelu(acc, a[threadIdx.x]);
  }
}

struct RNetW {
  const float *c1w, *c1b, *a1, *c2w, *c2b, *a2, *c3w, *c3b, *a3, *d4w, *d4b, *a4, *d51w, *d51b, *d52w, *d52b;
};
struct ONetW {
  const float *c1w, *c1b, *a1, *c2w, *c2b, *a2, *c3w, *c3b, *a3, *c4w, *c4b, *a4, *d5w, *d5b, *a5, *d61w, *d61b, *d62w,
      *d62b, *d63w, *d63b;
};

// exclusive prefix of the per-frame candidate counts: compact batch index of (img, k) = offs[img] + k
__global__ void prefix_offsets_kernel(const int* __restrict__ cnt, int B, int* __restrict__ offs) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < B; ++i) { offs[i] = acc; acc += cnt[i]; }
    offs[B] = acc;
  }
}

// head outputs of the MFMA plans (hw floats per candidate: a0, a1, then the regression / landmark
// values) -> the per-frame tables the post kernels read: [softmax prob of class 1, values...]
__global__ void heads_scatter_kernel(const float* __restrict__ heads, int hw, const int* __restrict__ offs,
                                     const int* __restrict__ cnt, int c0, int cap, float* __restrict__ dst, int nf, int split, int KR) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, img = blockIdx.y;
  if (k >= cnt[img]) return;
  const int ci = offs[img] + k - c0;
  if (ci < 0 || ci >= cap) return;
  const float* hsrc = heads + (size_t)ci * hw;
  auto val = [&](int i) { return split ? (float)__builtin_bit_cast(sf16, hsrc[i]) : hsrc[i]; };
  float* o = dst + ((size_t)img * KR + k) * nf;
  const float a0 = val(0), a1 = val(1);
  const float m = fmaxf(a0, a1);
  const float e0 = expf(a0 - m), e1 = expf(a1 - m);
  o[0] = e1 / (e0 + e1);
  for (int i = 1; i < nf; ++i) o[i] = val(1 + i);
}

// mtcnn.py:84-99.  out: [score, reg0..3] per candidate.
__global__ void __launch_bounds__(256) rnet_kernel(const float* __restrict__ crops, const int* __restrict__ row_cnt,
                                                    RNetW w, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, img = blockIdx.y;
  if (k >= row_cnt[img]) return;
  float* A = reinterpret_cast<float*>(smem);            // 28*22*22 = 13552 floats
  float* Bf = A + 13552;                                // 28*11*11 = 3388 floats (>= 3*24*24 = 1728)
  float* Wb = Bf + 3388;                                // 2 * 9 * 48 = 864 floats: staged weights
  const float* src = crops + ((size_t)img * KEEP + k) * 3 * 24 * 24;
  for (int i = threadIdx.x; i < 1728; i += blockDim.x) Bf[i] = src[i];
  __syncthreads();
  lds_conv_prelu_ws<3, 3, 4, 4, 4>(Bf, 24, 24, A, 28, w.c1w, 28, w.c1b, w.a1, Wb);   // 28 x 22 x 22
  __syncthreads();
  lds_maxpool_ceil<3>(A, 28, 22, 22, Bf);                                            // 28 x 11 x 11
  __syncthreads();
  lds_conv_prelu_ws<28, 3, 4, 3, 2>(Bf, 11, 11, A, 48, w.c2w, 48, w.c2b, w.a2, Wb);  // 48 x 9 x 9
  __syncthreads();
  lds_maxpool_ceil<3>(A, 48, 9, 9, Bf);                                              // 48 x 4 x 4
  __syncthreads();
  lds_conv_prelu_ws<48, 2, 4, 1, 1>(Bf, 4, 4, A, 64, w.c3w, 64, w.c3b, w.a3, Wb);    // 64 x 3 x 3
  __syncthreads();
  lds_dense_permuted_prelu(A, 64, 3, 3, A + 1024, A + 2048, Bf, 128, w.d4w, w.d4b, w.a4);
  __syncthreads();
  if (threadIdx.x < 6) {
    const int o = threadIdx.x;
    const float* wr = o < 2 ? w.d51w + o * 128 : w.d52w + (o - 2) * 128;
    float acc = o < 2 ? w.d51b[o] : w.d52b[o - 2];
    for (int f = 0; f < 128; ++f) acc = fmaf(Bf[f], wr[f], acc);
    A[o] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(A[0], A[1]);
    const float e0 = expf(A[0] - m), e1 = expf(A[1] - m);
    float* o = out + ((size_t)img * KEEP + k) * 5;
    o[0] = e1 / (e0 + e1);
    o[1] = A[2]; o[2] = A[3]; o[3] = A[4]; o[4] = A[5];
  }
}

// mtcnn.py:138-157.  out: [score, reg0..3, lm0..9] per candidate.
__global__ void __launch_bounds__(512) onet_kernel(const float* __restrict__ crops, const int* __restrict__ row_cnt,
                                                    ONetW w, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = blockIdx.x, img = blockIdx.y;
  if (k >= row_cnt[img]) return;
  float* X = reinterpret_cast<float*>(smem);   // 32*23*23 = 16928 floats
  float* Wk = X + 16928;                       // 14112 floats: conv1 round (4*46*46 = 8464) / conv2 half (32*21*21)
  float* Q = Wk + 14112;                       // 6912 floats: input (3*48*48), later the 64x10x10 pooled map
  float* Wb = Q + 6912;                        // 2 * 9 * 64 = 1152 floats: staged weights
  const float* src = crops + ((size_t)img * KEEP + k) * 3 * 48 * 48;
  for (int i = threadIdx.x; i < 6912; i += blockDim.x) Q[i] = src[i];
  __syncthreads();
  for (int rd = 0; rd < 8; ++rd) {  // conv1 3->32 in rounds of 4 channels: the full 32x46x46 map would not fit LDS
    lds_conv_prelu_ws<3, 3, 4, 4, 2>(Q, 48, 48, Wk, 4, w.c1w + rd * 4, 32, w.c1b + rd * 4, w.a1 + rd * 4, Wb);
    __syncthreads();
    lds_maxpool_ceil<3>(Wk, 4, 46, 46, X + rd * 4 * 529);                 // -> 32 x 23 x 23
    __syncthreads();
  }
  for (int half = 0; half < 2; ++half) {  // conv2 32->64 in two 32-channel halves
    lds_conv_prelu_ws<32, 3, 8, 4, 1>(X, 23, 23, Wk, 32, w.c2w + half * 32, 64, w.c2b + half * 32, w.a2 + half * 32, Wb);
    __syncthreads();
    lds_maxpool_ceil<3>(Wk, 32, 21, 21, Q + half * 32 * 100);             // -> 64 x 10 x 10
    __syncthreads();
  }
  lds_conv_prelu_ws<64, 3, 4, 2, 1>(Q, 10, 10, Wk, 64, w.c3w, 64, w.c3b, w.a3, Wb);    // 64 x 8 x 8
  __syncthreads();
  lds_maxpool_ceil<2>(Wk, 64, 8, 8, X);                                   // 64 x 4 x 4
  __syncthreads();
  lds_conv_prelu_ws<64, 2, 4, 1, 1>(X, 4, 4, Wk, 128, w.c4w, 128, w.c4b, w.a4, Wb);    // 128 x 3 x 3
  __syncthreads();
  float* Y = Wk;
  lds_dense_permuted_prelu(Wk, 128, 3, 3, Q, Q + 1152, X, 256, w.d5w, w.d5b, w.a5);
  __syncthreads();
  if (threadIdx.x < 16) {
    const int o = threadIdx.x;
    const float* wr = o < 2 ? w.d61w + o * 256 : o < 6 ? w.d62w + (o - 2) * 256 : w.d63w + (o - 6) * 256;
    float acc = o < 2 ? w.d61b[o] : o < 6 ? w.d62b[o - 2] : w.d63b[o - 6];
    for (int f = 0; f < 256; ++f) acc = fmaf(X[f], wr[f], acc);
    Y[o] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(Y[0], Y[1]);
    const float e0 = expf(Y[0] - m), e1 = expf(Y[1] - m);
    float* o = out + ((size_t)img * KEEP + k) * 15;
    o[0] = e1 / (e0 + e1);
    for (int i = 0; i < 14; ++i) o[1 + i] = Y[2 + i];
  }
}

// --------------------------------------------------------------------------------------------- stage-2 post
// detect_face.py:119-131: keep score > thr, batched_nms(0.7) per image, bbreg (w,h WITH +1), rerec;
// then pad for stage 3 (136).  Visiting order: score descending, ties by stage-1 table order.
template <bool BIG>
__device__ __forceinline__ void stage2_post_body(const Row* __restrict__ r, const float* __restrict__ ro, int n0, float thr_score,
                                                 float thr_nms, int W, int H, unsigned long long* keys, float4* kbox, int* keepl,
                                                 float4* s_cbox, int* s_alive, Row* __restrict__ out, int* __restrict__ out_cnt,
                                                 int* status) {
  __shared__ int s_n;
  const int npad = next_pow2(max(n0, 1));
  for (int i = threadIdx.x; i < npad; i += blockDim.x)
    keys[i] = (i < n0 && ro[i * 5] > thr_score) ? ((unsigned long long)inv_score_bits(ro[i * 5]) << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  block_sort(keys, n0, npad);
  if (threadIdx.x == 0) {
    int lo = 0, hi = n0;  // first padded key
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] == ~0ull) hi = mid; else lo = mid + 1; }
    s_n = lo;
  }
  __syncthreads();
  const int n = s_n;
  auto getbox = [&](int q) { const Row& b = r[(int)(keys[q] & 0xFFFFFFFFu)]; return float4{b.x1, b.y1, b.x2, b.y2}; };
  const int nk = block_greedy_nms<NMS_TV>(n, thr_nms, getbox, keepl, kbox, max(n, 1), s_cbox, s_alive, status);
  for (int k = threadIdx.x; k < nk; k += blockDim.x) {
    const int src = (int)(keys[keepl[k]] & 0xFFFFFFFFu);
    const float4 b = kbox[k];
    const float* mv = ro + src * 5 + 1;
    const float w = b.z - b.x + 1.f, h = b.w - b.y + 1.f;
    float x1 = b.x + mv[0] * w, y1 = b.y + mv[1] * h, x2 = b.z + mv[2] * w, y2 = b.w + mv[3] * h;
    Row o;
    rerec_pad(x1, y1, x2, y2, W, H, o);
    o.score = ro[src * 5];
    out[k] = o;
  }
  if (threadIdx.x == 0) *out_cnt = nk;
}

__global__ void __launch_bounds__(256) stage2_post_kernel(const Row* __restrict__ rows, const int* __restrict__ row_cnt,
                                                           const float* __restrict__ rout, float thr_score, float thr_nms,
                                                           int W, int H, int KR, Row* __restrict__ rows3, int* __restrict__ row3_cnt,
                                                           int* __restrict__ status, NmsScratch g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);  // KEEP * 8
  float4* s_kbox = reinterpret_cast<float4*>(smem + KEEP * 8);
  int* s_keep = reinterpret_cast<int*>(smem + KEEP * 8 + KEEP * 16);
  float4* s_cbox = reinterpret_cast<float4*>(smem + KEEP * 28);
  int* s_alive = reinterpret_cast<int*>(smem + KEEP * 28 + 256 * 16);
  const int img = blockIdx.x;
  const int n0 = row_cnt[img];
  const Row* r = rows + (size_t)img * KR;
  const float* ro = rout + (size_t)img * KR * 5;
  Row* out = rows3 + (size_t)img * KR;     // survivors are a subset of the n0 <= KR input rows: no overflow
  const size_t gb = (size_t)img * g.stride;
  if (n0 <= KEEP) stage2_post_body<false>(r, ro, n0, thr_score, thr_nms, W, H, keys, s_kbox, s_keep, s_cbox, s_alive, out, row3_cnt + img, status);
  else stage2_post_body<true>(r, ro, n0, thr_score, thr_nms, W, H, g.keys + gb, g.kbox + gb, g.keep + gb, s_cbox, s_alive, out, row3_cnt + img, status);
}

// --------------------------------------------------------------------------------------------- K7
// detect_face.py:148-169: keep score > thr, landmarks, bbreg, nms_numpy(0.7, 'Min') per image
// (visit from the highest score; equal scores -- softmax saturates to exactly 1.0f on clear faces --
// are visited in table order: the reference leaves tie order to np.argsort's unstable default
// sort, which is implementation defined; the oracle pins the same rule), then mtcnn.py:334-340: order by box area
// descending (argsort ascending, reversed).  fin: [x1,y1,x2,y2,score, 10 landmark coords] rows.
template <bool BIG>
__device__ __forceinline__ void stage3_post_body(const Row* __restrict__ r, const float* __restrict__ oo, int n0, float thr_score,
                                                 float thr_nms, int select_largest, unsigned long long* keys, float4* kbox,
                                                 int* keepl, float4* reg, float4* s_cbox, int* s_alive, float* __restrict__ fo,
                                                 int* __restrict__ out_cnt, int* status) {
  __shared__ int s_n;
  for (int i = threadIdx.x; i < n0; i += blockDim.x) {  // bbreg of every row (w,h WITH +1)
    const Row& b = r[i];
    const float* mv = oo + i * 15 + 1;
    const float w = b.x2 - b.x1 + 1.f, h = b.y2 - b.y1 + 1.f;
    reg[i] = float4{b.x1 + mv[0] * w, b.y1 + mv[1] * h, b.x2 + mv[2] * w, b.y2 + mv[3] * h};
  }
  const int npad = next_pow2(max(n0, 1));
  for (int i = threadIdx.x; i < npad; i += blockDim.x)  // ties: earlier row first (see header note on ties)
    keys[i] = (i < n0 && oo[i * 15] > thr_score) ? ((unsigned long long)inv_score_bits(oo[i * 15]) << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  block_sort(keys, n0, npad);
  if (threadIdx.x == 0) {
    int lo = 0, hi = n0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] == ~0ull) hi = mid; else lo = mid + 1; }
    s_n = lo;
  }
  __syncthreads();
  const int n = s_n;
  auto srcof = [&](int q) { return (int)(keys[q] & 0xFFFFFFFFu); };
  auto getbox = [&](int q) { return reg[srcof(q)]; };
  const int nk = block_greedy_nms<NMS_MIN>(n, thr_nms, getbox, keepl, kbox, max(n, 1), s_cbox, s_alive, status);
  __syncthreads();
  // final order: area descending (argsort ascending reversed: ties -> later pick first)
  // the score-sorted keys are dead after this: resolve kept ranks to source rows, then reuse `keys`
  for (int k = threadIdx.x; k < nk; k += blockDim.x) keepl[k] = srcof(keepl[k]);
  __syncthreads();
  const int kpad = next_pow2(max(nk, 1));
  for (int k = threadIdx.x; k < kpad; k += blockDim.x) {
    if (k < nk) {
      const float4 b = kbox[k];
      const float area = (b.z - b.x) * (b.w - b.y);
      // areas may be negative in degenerate cases: map float to an order-preserving unsigned
      unsigned u = __float_as_uint(area);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      keys[k] = select_largest ? ((unsigned long long)(0xFFFFFFFFu - u) << 32) | (unsigned)(0x7FFFFFFF - k)
                               : (unsigned long long)k;
    } else {
      keys[k] = ~0ull;
    }
  }
  __syncthreads();
  block_sort(keys, nk, kpad);
  for (int q = threadIdx.x; q < nk; q += blockDim.x) {
    const int k = select_largest ? 0x7FFFFFFF - (int)(keys[q] & 0xFFFFFFFFu) : (int)keys[q];
    const int src = keepl[k];
    const Row& b = r[src];
    const float4 bb = kbox[k];
    float* o = fo + q * 15;
    o[0] = bb.x; o[1] = bb.y; o[2] = bb.z; o[3] = bb.w; o[4] = oo[src * 15];
    // detect_face.py:159-163 (boxes BEFORE bbreg): px = w_i * p + x1 - 1
    const float w_i = b.x2 - b.x1 + 1.f, h_i = b.y2 - b.y1 + 1.f;
    const float* lm = oo + src * 15 + 5;
    for (int j = 0; j < 5; ++j) {
      o[5 + 2 * j] = w_i * lm[j] + b.x1 - 1.f;
      o[6 + 2 * j] = h_i * lm[5 + j] + b.y1 - 1.f;
    }
  }
  if (threadIdx.x == 0) *out_cnt = nk;
}

__global__ void __launch_bounds__(256) stage3_post_kernel(const Row* __restrict__ rows3, const int* __restrict__ row3_cnt,
                                                           const float* __restrict__ oout, float thr_score, float thr_nms,
                                                           int select_largest, int KR, float* __restrict__ fin,
                                                           int* __restrict__ fin_cnt, int* __restrict__ status, NmsScratch g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  float4* s_kbox = reinterpret_cast<float4*>(smem + KEEP * 8);
  int* s_keep = reinterpret_cast<int*>(smem + KEEP * 8 + KEEP * 16);
  float4* s_cbox = reinterpret_cast<float4*>(smem + KEEP * 28);
  int* s_alive = reinterpret_cast<int*>(smem + KEEP * 28 + 256 * 16);
  float4* s_reg = reinterpret_cast<float4*>(smem + KEEP * 28 + 256 * 20);  // KEEP * 16: boxes after bbreg
  const int img = blockIdx.x;
  const int n0 = row3_cnt[img];
  const Row* r = rows3 + (size_t)img * KR;
  const float* oo = oout + (size_t)img * KR * 15;
  float* fo = fin + (size_t)img * KR * 15;
  const size_t gb = (size_t)img * g.stride;
  if (n0 <= KEEP) stage3_post_body<false>(r, oo, n0, thr_score, thr_nms, select_largest, keys, s_kbox, s_keep, s_reg, s_cbox, s_alive, fo, fin_cnt + img, status);
  else stage3_post_body<true>(r, oo, n0, thr_score, thr_nms, select_largest, g.keys + gb, g.kbox + gb, g.keep + gb, g.reg + gb, s_cbox, s_alive, fo, fin_cnt + img, status);
}

// =============================================================================================
// host side
constexpr int FIN_FAST = 32;  // faces per frame covered by the one-copy read-back (VNF_FIN_FAST lowers it: test hook)

struct Mtcnn : HandleBase {
  vnf_mtcnn_cfg cfg;
  PNetW pw; RNetW rw; ONetW ow;
  LevelTable cap_table;  // geometry at (max_height, max_width): sizes the buffers
  float *lvl = nullptr, *p1 = nullptr, *c2 = nullptr;
  Cand* cand = nullptr;                       // stage-1 records, dense by cell: [frame][cap_out]
  int *cells = nullptr, *keep1c = nullptr;    // per (level, frame) compact cell lists: P-Net hits / per-scale NMS survivors
  int keep = KEEP;                            // rows per frame of the stage-2 / stage-3 tables (vnf_mtcnn_cfg.max_candidates)
  NmsScratch scratch{};                       // global-memory fallback of the NMS kernels
  int *cand_cnt = nullptr, *keep1_cnt = nullptr, *row_cnt = nullptr, *row3_cnt = nullptr, *fin_cnt = nullptr, *status = nullptr;
  Row *rows = nullptr, *rows3 = nullptr;
  float *crops = nullptr, *rout = nullptr, *oout = nullptr, *fin = nullptr;
  float *prob_dbg = nullptr, *reg_dbg = nullptr;
  Encoder *renc = nullptr, *oenc = nullptr;  // R-Net / O-Net plans on the exact-f32 MFMA core (candidates = batch)
  int* row_order = nullptr;                   // pyramid dispatch order (device), rebuilt when the frame size changes
  int row_order_h = 0, row_order_w = 0, row_order_cap = 0;
  int pnet1_lds = 0;                          // dynamic LDS granted to pnet_conv1_pool_mfma_kernel
  bool front = false;                         // conv1 + PReLU + pool1 of both nets by net_front_kernel (plans start at conv2)
  FrontW rfw{}, ofw{};
  bool mid = false;                           // conv2 + PReLU + pool2 by net_mid_kernel (split-f16 plans start at conv3)
  MidW rmw{}, omw{};
  int r_cap = 0, o_cap = 0;
  int* offs = nullptr;                        // device: (max_batch + 1) compact-batch offsets
  // final read-back: counts block + the first FIN_FAST rows of every frame packed by one kernel into `stage`,
  // one D2H copy into pinned memory, one host synchronisation (a frame with more faces takes the 2-D copy)
  float* stage = nullptr;
  int* h_pin = nullptr;
  int fin_fast = FIN_FAST;
  int last_b = 0;  // frames of the last vnf_mtcnn_detect (vnf_mtcnn_results_device)
  struct Spec { bool valid = false; int b = 0, H = 0, W = 0, max2 = 0, total2 = 0, max3 = 0, total3 = 0; } spec;   // launch sizes of stages 2 / 3 from the previous call
  long long spec_misses = 0;
  ~Mtcnn() override { delete renc; delete oenc; if (h_pin) (void)hipHostFree(h_pin); }
  size_t cap_px = 0, cap_p1 = 0, cap_c2 = 0, cap_out = 0;
  std::vector<float> h_fin;
  std::vector<int> h_cnt;
  LevelTable last_table;
};


// counts block (row_cnt .. status) followed by [B][FIN_FAST][15] result rows
__global__ void pack_results_kernel(const int* __restrict__ cnt_block, int ncnt, const float* __restrict__ fin,
                                    const int* __restrict__ fin_cnt, int B, int KR, float* __restrict__ stage) {
  int* so = reinterpret_cast<int*>(stage);
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < ncnt; i += gridDim.x * blockDim.x) so[i] = cnt_block[i];
  float* ro = stage + ncnt;
  const int total = B * FIN_FAST * 15;
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < total; i += gridDim.x * blockDim.x) {
    const int img = i / (FIN_FAST * 15), r = i - img * FIN_FAST * 15, k = r / 15;
    ro[i] = k < fin_cnt[img] ? fin[(size_t)img * KR * 15 + r] : 0.f;
  }
}

// device-resident copy of the last detection, frames concatenated in order (the host arrays' layout)
__global__ void results_device_kernel(const float* __restrict__ fin, const int* __restrict__ fin_cnt, int max_out, int KR,
                                      int32_t* __restrict__ fidx, float* __restrict__ boxes, float* __restrict__ probs,
                                      float* __restrict__ points) {
  const int img = blockIdx.x;
  int off = 0;
  for (int i = 0; i < img; ++i) off += fin_cnt[i];
  const int c = fin_cnt[img];
  for (int k = threadIdx.x; k < c; k += blockDim.x) {
    const int o = off + k;
    if (o >= max_out) break;
    const float* f = fin + ((size_t)img * KR + k) * 15;
    if (fidx) fidx[o] = img;
    if (boxes) { boxes[o * 4] = f[0]; boxes[o * 4 + 1] = f[1]; boxes[o * 4 + 2] = f[2]; boxes[o * 4 + 3] = f[3]; }
    if (probs) probs[o] = f[4];
    if (points)
      for (int j = 0; j < 10; ++j) points[o * 10 + j] = f[5 + j];
  }
}

static LevelTable make_levels(int h, int w, int minsize, double factor) {
  // detect_face.py:50-60,71 in python-double arithmetic
  LevelTable t;
  memset(&t, 0, sizeof(t));
  const double m = 12.0 / minsize;
  double minl = std::min(h, w) * m, scale = m;
  int opx = 0, op1 = 0, oc2 = 0, oout = 0;
  while (minl >= 12 && t.n < MAX_LEVELS) {
    LevelDesc& L = t.l[t.n];
    L.Hs = (int)(h * scale + 1);
    L.Ws = (int)(w * scale + 1);
    L.Hp = (L.Hs - 2 + 1) / 2;  // ceil((Hs-2)/2)
    L.Wp = (L.Ws - 2 + 1) / 2;
    L.H2 = L.Hp - 2; L.W2 = L.Wp - 2;
    L.oh = L.H2 - 2; L.ow = L.W2 - 2;
    L.scale = (float)scale;
    L.off_px = opx; L.off_p1 = op1; L.off_c2 = oc2; L.off_out = oout;
    opx += L.Hs * L.Ws; op1 += L.Hp * L.Wp; oc2 += L.H2 * L.W2; oout += L.oh * L.ow;
    ++t.n;
    scale = scale * factor;
    minl = minl * factor;
  }
  t.tot_px = opx; t.tot_p1 = op1; t.tot_c2 = oc2; t.tot_out = oout;
  return t;
}

static const float* up_transposed(Mtcnn& m, const float* w, int cout, int cin, int k) {
  // [cout][cin][k][k] -> [cin][k][k][cout]
  std::vector<float> t((size_t)cout * cin * k * k);
  for (int co = 0; co < cout; ++co)
    for (int c = 0; c < cin; ++c)
      for (int i = 0; i < k * k; ++i) t[((size_t)c * k * k + i) * cout + co] = w[((size_t)co * cin + c) * k * k + i];
  return (const float*)m.upload(t.data(), t.size() * 4);
}

#define GETW(dst, wmref, name, numel)                                                        \
  const float* dst = (wmref).get(name, numel);                                               \
  if (!dst) { delete m; return fail(VNF_E_MISSING, std::string("mtcnn: missing weight ") + (wmref).missing); }
#define UP(ptr, numel) (const float*)m->upload(ptr, (size_t)(numel) * 4)

}  // namespace vnf
using namespace vnf;

extern "C" int vnf_mtcnn_create(const vnf_tensor_desc* pnet, int n_pnet, const vnf_tensor_desc* rnet, int n_rnet,
                                const vnf_tensor_desc* onet, int n_onet, const vnf_mtcnn_cfg* cfg, vnf_handle* out) {
  try {
    if (!pnet || !rnet || !onet || !cfg || !out) return fail(VNF_E_INVALID, "vnf_mtcnn_create: bad argument");
    if (cfg->min_face_size < 1 || cfg->max_batch < 1 || cfg->max_height < 12 || cfg->max_width < 12 ||
        !(cfg->factor > 0.f && cfg->factor < 1.f))
      return fail(VNF_E_INVALID, "vnf_mtcnn_create: bad configuration");
    *out = nullptr;
    Mtcnn* m = new Mtcnn();
    m->kind = 3;
    m->cfg = *cfg;
    (void)hipGetDevice(&m->device);
    WeightMap wp(pnet, n_pnet), wr(rnet, n_rnet), wo(onet, n_onet);
    {
      GETW(c1, wp, "conv1.weight", 270) GETW(b1, wp, "conv1.bias", 10) GETW(a1, wp, "prelu1.weight", 10)
      GETW(c2, wp, "conv2.weight", 1440) GETW(b2, wp, "conv2.bias", 16) GETW(a2, wp, "prelu2.weight", 16)
      GETW(c3, wp, "conv3.weight", 4608) GETW(b3, wp, "conv3.bias", 32) GETW(a3, wp, "prelu3.weight", 32)
      GETW(c41, wp, "conv4_1.weight", 64) GETW(b41, wp, "conv4_1.bias", 2)
      GETW(c42, wp, "conv4_2.weight", 128) GETW(b42, wp, "conv4_2.bias", 4)
      m->pw.w1 = up_transposed(*m, c1, 10, 3, 3); m->pw.b1 = UP(b1, 10); m->pw.a1 = UP(a1, 10);
      m->pw.w2 = up_transposed(*m, c2, 16, 10, 3); m->pw.b2 = UP(b2, 16); m->pw.a2 = UP(a2, 16);
      m->pw.w3 = up_transposed(*m, c3, 32, 16, 3); m->pw.b3 = UP(b3, 32); m->pw.a3 = UP(a3, 32);
      m->pw.w41 = up_transposed(*m, c41, 2, 32, 1); m->pw.b41 = UP(b41, 2);
      m->pw.w42 = up_transposed(*m, c42, 4, 32, 1); m->pw.b42 = UP(b42, 4);
    }
    {
      GETW(c1, wr, "conv1.weight", 756) GETW(b1, wr, "conv1.bias", 28) GETW(a1, wr, "prelu1.weight", 28)
      GETW(c2, wr, "conv2.weight", 12096) GETW(b2, wr, "conv2.bias", 48) GETW(a2, wr, "prelu2.weight", 48)
      GETW(c3, wr, "conv3.weight", 12288) GETW(b3, wr, "conv3.bias", 64) GETW(a3, wr, "prelu3.weight", 64)
      GETW(d4, wr, "dense4.weight", 73728) GETW(d4b, wr, "dense4.bias", 128) GETW(a4, wr, "prelu4.weight", 128)
      GETW(d51, wr, "dense5_1.weight", 256) GETW(d51b, wr, "dense5_1.bias", 2)
      GETW(d52, wr, "dense5_2.weight", 512) GETW(d52b, wr, "dense5_2.bias", 4)
      m->rw = RNetW{up_transposed(*m, c1, 28, 3, 3), UP(b1, 28), UP(a1, 28), up_transposed(*m, c2, 48, 28, 3), UP(b2, 48),
                    UP(a2, 48), up_transposed(*m, c3, 64, 48, 2), UP(b3, 64), UP(a3, 64), up_transposed(*m, d4, 128, 576, 1),
                    UP(d4b, 128), UP(a4, 128), UP(d51, 256), UP(d51b, 2), UP(d52, 512), UP(d52b, 4)};
    }
    {
      GETW(c1, wo, "conv1.weight", 864) GETW(b1, wo, "conv1.bias", 32) GETW(a1, wo, "prelu1.weight", 32)
      GETW(c2, wo, "conv2.weight", 18432) GETW(b2, wo, "conv2.bias", 64) GETW(a2, wo, "prelu2.weight", 64)
      GETW(c3, wo, "conv3.weight", 36864) GETW(b3, wo, "conv3.bias", 64) GETW(a3, wo, "prelu3.weight", 64)
      GETW(c4, wo, "conv4.weight", 32768) GETW(b4, wo, "conv4.bias", 128) GETW(a4, wo, "prelu4.weight", 128)
      GETW(d5, wo, "dense5.weight", 294912) GETW(d5b, wo, "dense5.bias", 256) GETW(a5, wo, "prelu5.weight", 256)
      GETW(d61, wo, "dense6_1.weight", 512) GETW(d61b, wo, "dense6_1.bias", 2)
      GETW(d62, wo, "dense6_2.weight", 1024) GETW(d62b, wo, "dense6_2.bias", 4)
      GETW(d63, wo, "dense6_3.weight", 2560) GETW(d63b, wo, "dense6_3.bias", 10)
      m->ow = ONetW{up_transposed(*m, c1, 32, 3, 3), UP(b1, 32), UP(a1, 32), up_transposed(*m, c2, 64, 32, 3), UP(b2, 64), UP(a2, 64),
                    up_transposed(*m, c3, 64, 64, 3), UP(b3, 64), UP(a3, 64), up_transposed(*m, c4, 128, 64, 2), UP(b4, 128),
                    UP(a4, 128), up_transposed(*m, d5, 256, 1152, 1), UP(d5b, 256), UP(a5, 256),
                    UP(d61, 512), UP(d61b, 2), UP(d62, 1024), UP(d62b, 4), UP(d63, 2560), UP(d63b, 10)};
    }
    {
      static const int lds_nets = getenv("VNF_MTCNN_LDSNETS") ? atoi(getenv("VNF_MTCNN_LDSNETS")) : 0;
      if (!lds_nets) {
        m->r_cap = std::min(cfg->max_batch * KEEP, 8192);
        m->o_cap = std::min(cfg->max_batch * KEEP, 2048);
        m->renc = new Encoder();
        m->renc->max_streams = 1;  // the detector shares the GPU with the embedding stream: no forks of its own
        m->renc->tune_batch = std::max(1, m->r_cap / 2);  // typical stage-2 load, not the capacity
        // VNF_MTCNN_DTYPE=f32 keeps the R/O-Net plans (and conv1 in net_front_kernel) on the exact-f32 MFMA; the default is
        // split-f16 (two 16-bit MFMAs per product, ~22 significant bits) for every layer of both nets
        static const bool plans_f32 = getenv("VNF_MTCNN_DTYPE") && !strcmp(getenv("VNF_MTCNN_DTYPE"), "f32");
        m->renc->kind = 1; m->renc->arch = -2; m->renc->dtype = F32; m->renc->max_batch = m->r_cap;
        static const bool front_env = !getenv("VNF_MTCNN_FRONT") || atoi(getenv("VNF_MTCNN_FRONT")) != 0;
        m->front = front_env;
        if (m->front && !plans_f32) m->renc->dtype = F16X2;
        if (m->front) {
          auto pack_front = [&](WeightMap& wm, int cout, FrontW& fw) -> bool {
            const float* c1 = wm.get("conv1.weight", (int64_t)cout * 27);
            const float* b1 = wm.get("conv1.bias", cout);
            const float* a1 = wm.get("prelu1.weight", cout);
            if (!c1 || !b1 || !a1) return false;
            std::vector<float> w(32 * 36, 0.f), b(32, 0.f), a(32, 0.f);
            for (int co = 0; co < cout; ++co) {
              for (int c = 0; c < 3; ++c)
                for (int kh = 0; kh < 3; ++kh)
                  for (int kw = 0; kw < 3; ++kw) w[(co * 9 + kh * 3 + kw) * 4 + c] = c1[((co * 3 + c) * 3 + kh) * 3 + kw];
              b[co] = b1[co]; a[co] = a1[co];
            }
            fw.w = (const float*)m->upload(w.data(), w.size() * 4);
            fw.b = (const float*)m->upload(b.data(), b.size() * 4);
            fw.a = (const float*)m->upload(a.data(), a.size() * 4);
            return fw.w && fw.b && fw.a;
          };
          if (!pack_front(wr, 28, m->rfw) || !pack_front(wo, 32, m->ofw)) { delete m; return fail(VNF_E_MISSING, "mtcnn: conv1 weights"); }
        }
        const bool mid_env = !getenv("VNF_MTCNN_MID") || atoi(getenv("VNF_MTCNN_MID")) != 0;   // read per handle
        m->mid = mid_env && m->front && m->renc->dtype == F16X2;
        if (m->mid) {
          // conv2 weights [cout][cin][3][3] -> MFMA A-fragments of interleaved split-f16: fragment (ct, kb = 2 tap + half),
          // lane (row r, group g) = the 4 k values (channels 16 half + 4 g .. + 3 of the tap) of output channel 16 ct + r
          // as (hi, lo) pairs; input channels beyond cin (R-Net: 28 of 32) are zero
          auto pack_mid = [&](WeightMap& wm, int cout, int cin, MidW& mw) -> bool {
            const float* c2 = wm.get("conv2.weight", (int64_t)cout * cin * 9);
            const float* b2 = wm.get("conv2.bias", cout);
            const float* a2 = wm.get("prelu2.weight", cout);
            if (!c2 || !b2 || !a2) return false;
            std::vector<uint32_t> w((size_t)(cout / 16) * 18 * 64 * 4, 0u);
            for (int ct = 0; ct < cout / 16; ++ct)
              for (int kb = 0; kb < 18; ++kb)
                for (int l = 0; l < 64; ++l)
                  for (int e = 0; e < 4; ++e) {
                    const int co = 16 * ct + (l & 15), c = 16 * (kb & 1) + 4 * (l >> 4) + e, tap = kb >> 1;
                    const float v = c < cin ? c2[((size_t)(co * cin + c) * 3 + tap / 3) * 3 + tap % 3] : 0.f;
                    const sf16 sv(v);
                    uint32_t bits;
                    memcpy(&bits, &sv, 4);
                    w[(((size_t)ct * 18 + kb) * 64 + l) * 4 + e] = bits;
                  }
            mw.w = (const uint4*)m->upload(w.data(), w.size() * 4);
            mw.b = (const float*)m->upload(b2, (size_t)cout * 4);
            mw.a = (const float*)m->upload(a2, (size_t)cout * 4);
            return mw.w && mw.b && mw.a;
          };
          if (!pack_mid(wr, 48, 28, m->rmw) || !pack_mid(wo, 64, 32, m->omw)) { delete m; return fail(VNF_E_MISSING, "mtcnn: conv2 weights"); }
          (void)hipFuncSetAttribute((const void*)net_mid_kernel<23, 64, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 23 * 23 * 128 + 21 * 21 * 8 * 16);
          (void)hipGetLastError();
        }
        int rr = build_rnet(*m->renc, wr, m->front, m->mid);
        if (rr == VNF_OK) rr = m->renc->finalize();
        m->oenc = new Encoder();
        m->oenc->max_streams = 1;
        m->oenc->tune_batch = std::max(1, m->o_cap / 4);
        m->oenc->kind = 1; m->oenc->arch = -3; m->oenc->dtype = m->renc->dtype; m->oenc->max_batch = m->o_cap;
        if (rr == VNF_OK) rr = build_onet(*m->oenc, wo, m->front, m->mid);
        if (rr == VNF_OK) rr = m->oenc->finalize();
        if (rr != VNF_OK) { delete m; return rr; }
      }
    }
    const int B = cfg->max_batch;
    m->cap_table = make_levels(cfg->max_height, cfg->max_width, cfg->min_face_size, (double)cfg->factor);
    // other aspect ratios up to the same bounds can need slightly more: 10 % head-room
    m->cap_px = (size_t)(m->cap_table.tot_px * 1.1) + 4096; m->cap_p1 = (size_t)(m->cap_table.tot_p1 * 1.1) + 4096;
    m->cap_c2 = (size_t)(m->cap_table.tot_c2 * 1.1) + 4096; m->cap_out = (size_t)(m->cap_table.tot_out * 1.1) + 4096;
    {
      int rows_cap = 0;
      for (int l = 0; l < m->cap_table.n; ++l) rows_cap += m->cap_table.l[l].Hs;
      m->row_order_cap = (int)(rows_cap * 1.1) + 64;
      m->row_order = (int*)m->dalloc((size_t)m->row_order_cap * 4);
      if (!m->row_order) { delete m; return VNF_E_HIP; }
    }
    m->lvl = (float*)m->dalloc(m->cap_px * 3 * B * 4);
    m->p1 = (float*)m->dalloc(m->cap_p1 * 10 * B * 4);
    m->c2 = (float*)m->dalloc(m->cap_c2 * 16 * B * 4);
    const size_t nseg = (size_t)MAX_LEVELS * B;
    // rows per frame of the stage-2 / stage-3 tables: run-time (max_candidates), at least the LDS fast-path size
    m->keep = std::max(KEEP, cfg->max_candidates);
    if (!m->renc && m->keep != KEEP) { delete m; return fail(VNF_E_INVALID, "mtcnn: the LDS-resident nets (VNF_MTCNN_LDSNETS) keep the 2048-row tables"); }
    const size_t KR = (size_t)m->keep;
    m->cand = (Cand*)m->dalloc((size_t)B * m->cap_out * sizeof(Cand));
    m->cells = (int*)m->dalloc((size_t)B * m->cap_out * 4);
    m->keep1c = (int*)m->dalloc((size_t)B * m->cap_out * 4);
    m->cand_cnt = (int*)m->dalloc((nseg * 2 + (size_t)B * 3 + 16) * 4);
    m->keep1_cnt = m->cand_cnt + nseg;
    m->row_cnt = m->keep1_cnt + nseg;
    m->row3_cnt = m->row_cnt + B;
    m->fin_cnt = m->row3_cnt + B;
    m->status = m->fin_cnt + B;
    m->rows = (Row*)m->dalloc((size_t)B * KR * sizeof(Row));
    m->rows3 = (Row*)m->dalloc((size_t)B * KR * sizeof(Row));
    m->crops = m->renc ? (float*)m->dalloc(16) : (float*)m->dalloc((size_t)B * KEEP * 3 * 48 * 48 * 4);   // planar crops: LDS-resident nets only
    m->rout = (float*)m->dalloc((size_t)B * KR * 5 * 4);
    m->oout = (float*)m->dalloc((size_t)B * KR * 15 * 4);
    m->fin = (float*)m->dalloc((size_t)B * KR * 15 * 4);
    {
      // NMS scratch in global memory (lists longer than the LDS tables): per frame max(cells of the pyramid, rows)
      const size_t st = std::max(m->cap_out, KR);
      m->scratch.stride = (int)st;
      m->scratch.keys = (unsigned long long*)m->dalloc((size_t)B * st * 8);
      m->scratch.kbox = (float4*)m->dalloc((size_t)B * st * 16);
      m->scratch.keep = (int*)m->dalloc((size_t)B * st * 4);
      m->scratch.reg = (float4*)m->dalloc((size_t)B * st * 16);
      if (!m->scratch.keys || !m->scratch.kbox || !m->scratch.keep || !m->scratch.reg || !m->cells || !m->keep1c) { delete m; return VNF_E_HIP; }
    }
    {
      const size_t sb = ((size_t)B * 3 + 16) * 4 + (size_t)B * FIN_FAST * 15 * 4;
      m->stage = (float*)m->dalloc(sb);
      if (hipHostMalloc((void**)&m->h_pin, sb, hipHostMallocDefault) != hipSuccess) m->h_pin = nullptr;
      if (!m->stage || !m->h_pin) { delete m; return fail(VNF_E_HIP, "mtcnn: read-back buffers"); }
      if (const char* ff = getenv("VNF_FIN_FAST")) m->fin_fast = std::max(0, std::min(FIN_FAST, atoi(ff)));
    }
    m->offs = (int*)m->dalloc((size_t)(B + 1) * 4);
    if (!m->lvl || !m->p1 || !m->c2 || !m->cand || !m->cand_cnt || !m->rows || !m->rows3 || !m->crops ||
        !m->rout || !m->oout || !m->fin || !m->pw.w1 || !m->ow.d63b) {
      delete m;
      return VNF_E_HIP;
    }
    // Dynamic LDS above 64 KiB: opt in with the exact sizes (gfx950 has 160 KiB per workgroup).
    // The attribute call is advisory on some ROCm builds; launch errors are checked at run time.
    {
      int lds_max = 0;
      VNF_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, m->device));
      const int need_img = CAP_LDS_KEYS * 8 + KEEP * 20 + 256 * 20, need_scale = need_img;
      const int need_post = KEEP * 44 + 256 * 20, need_r = (13552 + 3388 + 864) * 4, need_o = (16928 + 14112 + 6912 + 1152) * 4;
      (void)hipFuncSetAttribute((const void*)nms_image_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_img);
      (void)hipFuncSetAttribute((const void*)net_front_kernel<24, 11, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
      (void)hipFuncSetAttribute((const void*)net_front_kernel<24, 11, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
      {
        int wmax = 0;
        for (int l = 0; l < m->cap_table.n; ++l) wmax = std::max(wmax, m->cap_table.l[l].Ws);
        wmax = (int)(wmax * 1.1) + 64;   // same head-room as the level buffers
        const int need_p1 = std::min((12 * (wmax + 64) + 10 * ((wmax - 1) / 2)) * 4, 160 * 1024);
        if (hipFuncSetAttribute((const void*)pnet_conv1_pool_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_p1) == hipSuccess)
          m->pnet1_lds = need_p1;
      }
      (void)hipFuncSetAttribute((const void*)nms_scale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_scale);
      (void)hipFuncSetAttribute((const void*)stage2_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_post);
      (void)hipFuncSetAttribute((const void*)stage3_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_post);
      (void)hipFuncSetAttribute((const void*)rnet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_r);
      (void)hipFuncSetAttribute((const void*)onet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need_o);
      (void)hipGetLastError();
      if (lds_max < need_o) {
        delete m;
        return fail(VNF_E_INVALID, "mtcnn: device reports " + std::to_string(lds_max) + " B of LDS per workgroup, need " + std::to_string(need_o));
      }
    }
    VNF_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<vnf_handle>(static_cast<HandleBase*>(m));
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

// per-stage device time + algorithmic bytes of one call (vnf_mtcnn_stage_times): events between the stages' launches
struct StageProf {
  std::vector<std::string> name;
  std::vector<double> bytes;
  std::vector<hipEvent_t> ev;
  void mark(const char* n, double b, hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    name.push_back(n); bytes.push_back(b); ev.push_back(e);
  }
  ~StageProf() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};

static int mtcnn_run(Mtcnn* m, const uint8_t* frames, int b, int H, int W, hipStream_t s, std::vector<int>& cnt,
                     std::vector<float>& fin, StageProf* prof = nullptr) {
  const vnf_mtcnn_cfg& cfg = m->cfg;
  // a mark closes the stage named in it: its time is the span since the previous mark
  auto mark = [&](const char* n, double bytes) { if (prof) prof->mark(n, bytes, s); };
  if (b > cfg.max_batch || H > cfg.max_height || W > cfg.max_width) return fail(VNF_E_CAPACITY, "mtcnn: frame batch exceeds handle capacity");
  LevelTable t = make_levels(H, W, cfg.min_face_size, (double)cfg.factor);
  m->last_table = t;
  m->last_b = 0;
  cnt.assign(b, 0);
  fin.clear();
  if (t.n == 0) return VNF_OK;  // image smaller than one cell: no detections
  if ((size_t)t.tot_px > m->cap_px || (size_t)t.tot_p1 > m->cap_p1 || (size_t)t.tot_c2 > m->cap_c2 || (size_t)t.tot_out > m->cap_out)
    return fail(VNF_E_CAPACITY, "mtcnn: pyramid exceeds handle capacity");
  const int B = b;
  const size_t nseg = (size_t)MAX_LEVELS * cfg.max_batch;
  VNF_HIP(hipMemsetAsync(m->cand_cnt, 0, (nseg * 2 + (size_t)cfg.max_batch * 3 + 16) * 4, s));
  mark("begin", 0);
  {
    // bin sums must stay exact in fp32 (< 2^24): always true below 256x256-pixel bins
    const bool fast = (W * 3) % 16 == 0 && (size_t)W * 12 <= 64 * 1024 && ((reinterpret_cast<uintptr_t>(frames)) & 15) == 0;
    if (fast) {
      int rows = 0;
      for (int l = 0; l < t.n; ++l) rows += t.l[l].Hs;
      if (rows > m->row_order_cap) return fail(VNF_E_CAPACITY, "mtcnn: pyramid exceeds handle capacity");
      if (m->row_order_h != H || m->row_order_w != W) {
        std::vector<std::pair<long long, int>> ord;     // (first input row, tall bins first) -> (level << 16 | row)
        for (int l = 0; l < t.n; ++l)
          for (int i = 0; i < t.l[l].Hs; ++i) {
            const long long h0 = ((long long)i * H) / t.l[l].Hs;
            ord.push_back({h0 * 64 + (63 - std::min(l, 63)), (l << 16) | i});
          }
        std::sort(ord.begin(), ord.end());
        std::vector<int> packed(ord.size());
        for (size_t k = 0; k < ord.size(); ++k) packed[k] = ord[k].second;
        VNF_HIP(hipMemcpyAsync(m->row_order, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, s));
        VNF_HIP(hipStreamSynchronize(s));              // the host vector goes away; happens once per frame size
        m->row_order_h = H; m->row_order_w = W;
      }
      hipLaunchKernelGGL(pyramid_rows_kernel, dim3(B, rows), dim3(256), (size_t)W * 12, s, frames, H, W, t, m->lvl, m->row_order);
    } else {
      hipLaunchKernelGGL(pyramid_kernel, dim3((t.tot_px + 255) / 256, B), dim3(256), 0, s, frames, H, W, t, m->lvl);
    }
  }
  // algorithmic bytes per launch: what each kernel must read + write once (SURVEY.md 8d terms, from the level table)
  const double fB = (double)B;
  mark("pyramid", fB * ((double)H * W * 3 + (double)t.tot_px * 12));
  {
    int rows = 0, wmax = 0;
    for (int l = 0; l < t.n; ++l) { rows += t.l[l].Hp; wmax = std::max(wmax, t.l[l].Ws); }
    const size_t lds = ((size_t)12 * (wmax + 64) + (size_t)10 * ((wmax - 1) / 2)) * 4;
    static const int p1_mode = getenv("VNF_PNET1") ? atoi(getenv("VNF_PNET1")) : 2;   // 0 VALU, 1 MFMA via LDS, 2 MFMA direct
    if (p1_mode == 2)
      hipLaunchKernelGGL(pnet_conv1_pool_direct_kernel, dim3(B, rows), dim3(256), 0, s, m->lvl, t, m->pw, m->p1);
    else if (p1_mode == 1 && lds <= (size_t)m->pnet1_lds)
      hipLaunchKernelGGL(pnet_conv1_pool_mfma_kernel, dim3(rows, B), dim3(256), lds, s, m->lvl, t, m->pw, m->p1);
    else
      hipLaunchKernelGGL(pnet_conv1_pool_kernel, dim3((t.tot_p1 + 255) / 256, B), dim3(256), 0, s, m->lvl, t, m->pw, m->p1);
  }
  mark("pnet_conv1_pool", fB * ((double)t.tot_px * 12 + (double)t.tot_p1 * 40));
  hipLaunchKernelGGL(pnet_conv2_kernel, dim3(B, (t.tot_c2 + 255) / 256), dim3(256), 0, s, m->p1, t, m->pw, m->c2);
  mark("pnet_conv2", fB * ((double)t.tot_p1 * 40 + (double)t.tot_c2 * 64));
  const int KR = m->keep, cap_out = (int)m->cap_out;
  hipLaunchKernelGGL(pnet_conv3_heads_kernel, dim3((t.tot_out + 255) / 256, B), dim3(256), 0, s, m->c2, t, m->pw,
                     cfg.thresholds[0], B, cap_out, m->cand, m->cells, m->cand_cnt, m->prob_dbg, m->reg_dbg);
  mark("pnet_conv3_heads", fB * (double)t.tot_c2 * 64);
  const size_t lds_nms = (size_t)CAP_LDS_KEYS * 8 + KEEP * 20 + 256 * 20;
  hipLaunchKernelGGL(nms_scale_kernel, dim3(t.n, B), dim3(256), lds_nms, s, m->cand, m->cells, m->cand_cnt, t, B, cap_out, 0.5f,
                     m->keep1c, m->keep1_cnt, m->status, m->scratch);
  hipLaunchKernelGGL(nms_image_kernel, dim3(B), dim3(256), lds_nms, s, m->cand, m->keep1c, m->keep1_cnt, t, B, cap_out, 0.7f, W, H,
                     KR, m->rows, m->row_cnt, m->status, m->scratch);
  VNF_HIP(hipGetLastError());
  mark("nms_stage1", 0);
  const int ncnt = cfg.max_batch * 3 + 16;
  int* const h = m->h_pin;  // pinned: the copy is a true async DMA, the only wait is the stream synchronisation
  auto read_counts = [&]() -> int {
    VNF_HIP(hipMemcpyAsync(h, m->row_cnt, (size_t)ncnt * 4, hipMemcpyDeviceToHost, s));
    VNF_HIP(hipStreamSynchronize(s));
    const int st = h[cfg.max_batch * 3];
    if (st & (ST_OVER_SCALE | ST_OVER_IMG | ST_OVER_KEEP))
      return fail(VNF_E_CAPACITY, "mtcnn: candidate table overflow (status " + std::to_string(st) + "): a frame has more than " +
                                  std::to_string(m->keep) + " stage-1 survivors; raise vnf_mtcnn_cfg.max_candidates");
    return VNF_OK;
  };
  // ---- stages 2 and 3 as launch sequences sized by (largest per-frame candidate count, total candidates): every kernel
  // reads the true counts from device memory and leaves early past them, so any UPPER bound gives the exact result (the
  // nets then also run on the unused tail rows of the dense batch); an under-estimate leaves candidates out and is
  // detected after the read-back.
  const bool crop_fast = (W * 3) % 16 == 0 && ((reinterpret_cast<uintptr_t>(frames)) & 15) == 0;
  auto crop = [&](const Row* rws, const int* cntp, int maxc, int S, float* dst, const int* offs, int c0, int cap) {
    if (crop_fast)
      // S / 8 row groups per candidate: 8 output rows per workgroup = 4 waves x 2 rows (measured best of 2..8 groups)
      hipLaunchKernelGGL(crop_resize_rows_kernel, dim3(maxc, B, S / 8), dim3(256), 0, s, frames, H, W, rws, cntp, S, dst, m->status, offs, c0, cap, KR);
    else
      hipLaunchKernelGGL(crop_resize_kernel, dim3(maxc, B), dim3(256), 0, s, frames, H, W, rws, cntp, S, dst, m->status, offs, c0, cap, KR);
  };
  // nets on the MFMA core: candidates of all frames form one dense batch, processed in chunks of `cap`
  auto run_net = [&](Encoder* enc, int cap, const Row* rws, const int* cntp, int maxc, int total, int S, int hw, float* dst,
                     int nf) -> int {
    hipLaunchKernelGGL(prefix_offsets_kernel, dim3(1), dim3(64), 0, s, cntp, B, m->offs);
    for (int c0 = 0; c0 < total; c0 += cap) {
      const int n = std::min(cap, total - c0);
      crop(rws, cntp, maxc, S, (float*)enc->bufs[0].ptr, m->offs, c0, n);
      mark(S == 24 ? "crop_resize_24" : "crop_resize_48", (double)n * S * S * 16);  // output bytes only (NHWC4 fp32)
      if (m->front) {
        const bool split = enc->dtype == F16X2;
        const float* cin = (const float*)enc->bufs[0].ptr;
        float* pout = (float*)enc->bufs[1].ptr;
        const size_t lr = (25 * 24 + 22 * 22 * 8) * 16, lo = (11 * 48 + 9 * 46 * 8) * 16;   // (IR * S + conv rows * C * 8) float4
        // R-Net: the whole candidate in one workgroup of 8 waves (no band overlap to recompute; measured 0.065 ms against
        // 0.074 for two bands x 4 waves); O-Net: bands of 4 pooled rows x 8 waves (larger bands / 16 waves were slower)
        // conv1 of the split-f16 plans on the 16-bit MFMA as well (VNF_MTCNN_FRONT16=0: exact-fp32 conv1, three times the
        // MFMA time); the f32 plans (VNF_MTCNN_DTYPE=f32) always take the exact kernel
        static const bool mm16 = !getenv("VNF_MTCNN_FRONT16") || atoi(getenv("VNF_MTCNN_FRONT16")) != 0;
        if (S == 24 && split && mm16) hipLaunchKernelGGL((net_front_kernel<24, 11, 512, true, true>), dim3(1, n), dim3(512), lr, s, cin, m->rfw, pout);
        else if (S == 48 && split && mm16) hipLaunchKernelGGL((net_front_kernel<48, 4, 512, true, true>), dim3(6, n), dim3(512), lo, s, cin, m->ofw, pout);
        else if (S == 24 && split) hipLaunchKernelGGL((net_front_kernel<24, 11, 512, true>), dim3(1, n), dim3(512), lr, s, cin, m->rfw, pout);
        else if (S == 24) hipLaunchKernelGGL((net_front_kernel<24, 11, 512, false>), dim3(1, n), dim3(512), lr, s, cin, m->rfw, pout);
        else if (split) hipLaunchKernelGGL((net_front_kernel<48, 4, 512, true>), dim3(6, n), dim3(512), lo, s, cin, m->ofw, pout);
        else hipLaunchKernelGGL((net_front_kernel<48, 4, 512, false>), dim3(6, n), dim3(512), lo, s, cin, m->ofw, pout);
        mark(S == 24 ? "rnet_front" : "onet_front", 0);
        if (m->mid) {   // conv2 + PReLU + pool2: buffer 1 -> buffer 3 (the plan starts at conv3)
          float* p2o = (float*)enc->bufs[3].ptr;
          if (S == 24) hipLaunchKernelGGL((net_mid_kernel<11, 48, 6, 1>), dim3(n), dim3(384), 11 * 11 * 128 + 9 * 9 * 12 * 16, s, pout, m->rmw, p2o);
          else hipLaunchKernelGGL((net_mid_kernel<23, 64, 8, 2>), dim3(n), dim3(512), 23 * 23 * 128 + 21 * 21 * 8 * 16, s, pout, m->omw, p2o);
        }
      }
      static const bool layers = getenv("VNF_MTCNN_LAYERS") != nullptr;   // diagnostic: per-layer table on stderr
      std::string rep;
      int rc = enc->run(nullptr, n, VNF_F32, nullptr, s, prof && layers ? &rep : nullptr);
      if (rc != VNF_OK) return rc;
      if (!rep.empty()) fprintf(stderr, "%s n=%d\n%s", S == 24 ? "rnet" : "onet", n, rep.c_str());
      mark(S == 24 ? "rnet" : "onet", 0);
      hipLaunchKernelGGL(heads_scatter_kernel, dim3((maxc + 63) / 64, B), dim3(64), 0, s, (const float*)enc->bufs.back().ptr, hw,
                         m->offs, cntp, c0, n, dst, nf, enc->dtype == F16X2 ? 1 : 0, KR);
    }
    return VNF_OK;
  };
  const size_t lds_post = (size_t)KEEP * 28 + 256 * 20 + KEEP * 16;
  auto stage2 = [&](int max2, int total2) -> int {
    if (m->renc) {
      const int rc = run_net(m->renc, m->r_cap, m->rows, m->row_cnt, max2, total2, 24, 8, m->rout, 5);
      if (rc != VNF_OK) return rc;
    } else {
      crop(m->rows, m->row_cnt, max2, 24, m->crops, nullptr, 0, 0);
      hipLaunchKernelGGL(rnet_kernel, dim3(max2, B), dim3(256), (13552 + 3388 + 864) * 4, s, m->crops, m->row_cnt, m->rw, m->rout);
    }
    hipLaunchKernelGGL(stage2_post_kernel, dim3(B), dim3(256), lds_post, s, m->rows, m->row_cnt, m->rout, cfg.thresholds[1], 0.7f,
                       W, H, KR, m->rows3, m->row3_cnt, m->status, m->scratch);
    VNF_HIP(hipGetLastError());
    mark("stage2_post", 0);
    return VNF_OK;
  };
  auto stage3 = [&](int max3, int total3) -> int {
    if (m->oenc) {
      const int rc = run_net(m->oenc, m->o_cap, m->rows3, m->row3_cnt, max3, total3, 48, 16, m->oout, 15);
      if (rc != VNF_OK) return rc;
    } else {
      crop(m->rows3, m->row3_cnt, max3, 48, m->crops, nullptr, 0, 0);
      hipLaunchKernelGGL(onet_kernel, dim3(max3, B), dim3(512), (16928 + 14112 + 6912 + 1152) * 4, s, m->crops, m->row3_cnt, m->ow, m->oout);
    }
    hipLaunchKernelGGL(stage3_post_kernel, dim3(B), dim3(256), lds_post, s, m->rows3, m->row3_cnt, m->oout, cfg.thresholds[2], 0.7f,
                       cfg.select_largest, KR, m->fin, m->fin_cnt, m->status, m->scratch);
    m->last_b = B;
    hipLaunchKernelGGL(pack_results_kernel, dim3(B), dim3(256), 0, s, m->row_cnt, ncnt, m->fin, m->fin_cnt, B, KR, m->stage);
    VNF_HIP(hipGetLastError());
    mark("stage3_post", 0);
    return VNF_OK;
  };
  auto readback = [&]() -> int {
    VNF_HIP(hipMemcpyAsync(h, m->stage, ((size_t)ncnt + (size_t)B * FIN_FAST * 15) * 4, hipMemcpyDeviceToHost, s));
    VNF_HIP(hipStreamSynchronize(s));
    mark("readback", 0);
    const int st = h[cfg.max_batch * 3];
    if (st & (ST_OVER_SCALE | ST_OVER_IMG | ST_OVER_KEEP))
      return fail(VNF_E_CAPACITY, "mtcnn: candidate table overflow (status " + std::to_string(st) + "): a frame has more than " +
                                  std::to_string(m->keep) + " stage-1 survivors; raise vnf_mtcnn_cfg.max_candidates");
    return VNF_OK;
  };
  auto counts_of = [&](int base, int& mx, int& tot) {
    mx = tot = 0;
    for (int i = 0; i < B; ++i) { mx = std::max(mx, h[base + i]); tot += h[base + i]; }
  };
  // an estimate with head room, in whole tiles of the nets' batch dimension
  auto padded = [&](int v, int limit) { return std::min(limit, ((v + v / 8 + 8 + 15) / 16) * 16); };
  // Sizes of stages 2 / 3 WITHOUT asking the device (the reference synchronises at both stage boundaries to shape its
  // tensors, detect_face.py:96-146): a video stream's candidate counts move slowly, so the previous call's counts plus
  // head room size this call's launches, and the one read-back at the end tells whether they covered it.  If not (or on
  // the first call of a frame size) stage-1's counts are read and stages 2 / 3 run with exact bounds: stage 2 by its
  // own counts, stage 3 by stage 2's (it only filters stage-2 rows) -- never a second mid-cascade synchronisation.
  static const int spec_on = getenv("VNF_MTCNN_SPEC") ? atoi(getenv("VNF_MTCNN_SPEC")) : 1;
  Mtcnn::Spec& sp = m->spec;
  int r = VNF_OK;
  bool exact_needed = true;
  if (spec_on && sp.valid && sp.b == B && sp.H == H && sp.W == W) {
    r = stage2(sp.max2, sp.total2);
    if (r == VNF_OK) r = stage3(sp.max3, sp.total3);
    if (r == VNF_OK) r = readback();
    if (r != VNF_OK) return r;
    int mx2, tot2, mx3, tot3;
    counts_of(0, mx2, tot2);
    counts_of(cfg.max_batch, mx3, tot3);
    exact_needed = mx2 > sp.max2 || tot2 > sp.total2 || mx3 > sp.max3 || tot3 > sp.total3;
    if (exact_needed) m->spec_misses++;
  } else {
    r = read_counts();
    if (r != VNF_OK) return r;
    mark("host_sync_1", 0);
  }
  if (exact_needed) {
    // h[0..B) = stage-1 counts (from read_counts, or from the read-back of the speculative pass: stage 1 is not re-run)
    int mx2, tot2;
    counts_of(0, mx2, tot2);
    if (mx2 > 0) {
      r = stage2(mx2, tot2);
      if (r == VNF_OK) r = stage3(mx2, tot2);      // stage-3 rows are a subset of stage-2 rows: exact upper bounds
    } else {
      m->last_b = B;
      hipLaunchKernelGGL(pack_results_kernel, dim3(B), dim3(256), 0, s, m->row_cnt, ncnt, m->fin, m->fin_cnt, B, KR, m->stage);
    }
    if (r == VNF_OK) r = readback();
    if (r != VNF_OK) return r;
  }
  {
    int mx2, tot2, mx3, tot3;
    counts_of(0, mx2, tot2);
    counts_of(cfg.max_batch, mx3, tot3);
    sp.valid = true; sp.b = B; sp.H = H; sp.W = W;
    sp.max2 = padded(mx2, KR); sp.total2 = padded(tot2, B * KR);
    sp.max3 = padded(mx3, KR); sp.total3 = padded(tot3, B * KR);
  }
  int maxf = 0;
  for (int i = 0; i < B; ++i) { cnt[i] = h[2 * cfg.max_batch + i]; maxf = std::max(maxf, cnt[i]); }
  if (maxf == 0) return VNF_OK;
  fin.resize((size_t)B * maxf * 15);
  if (maxf <= m->fin_fast) {
    const float* rows = reinterpret_cast<const float*>(h + ncnt);
    for (int i = 0; i < B; ++i)
      memcpy(&fin[(size_t)i * maxf * 15], rows + (size_t)i * FIN_FAST * 15, (size_t)maxf * 15 * 4);
    return VNF_OK;
  }
  VNF_HIP(hipMemcpy2DAsync(fin.data(), (size_t)maxf * 15 * 4, m->fin, (size_t)KR * 15 * 4, (size_t)maxf * 15 * 4, B,
                           hipMemcpyDeviceToHost, s));
  VNF_HIP(hipStreamSynchronize(s));
  return VNF_OK;
}

extern "C" int vnf_mtcnn_detect(vnf_handle h, const uint8_t* frames, int b, int height, int width, int32_t* counts,
                                float* boxes, float* probs, float* points, int max_out, int32_t* n_out, void* stream) {
  try {
    HandleBase* hb = reinterpret_cast<HandleBase*>(h);
    if (!hb || hb->kind != 3) return fail(VNF_E_INVALID, "not an MTCNN handle");
    if (!frames || b <= 0 || !counts || !n_out) return fail(VNF_E_INVALID, "vnf_mtcnn_detect: bad argument");
    Mtcnn* m = static_cast<Mtcnn*>(hb);
    std::vector<int> cnt;
    std::vector<float> fin;
    int r = mtcnn_run(m, frames, b, height, width, (hipStream_t)stream, cnt, fin);
    if (r != VNF_OK) return r;
    int total = 0, maxf = 0;
    for (int i = 0; i < b; ++i) { counts[i] = cnt[i]; total += cnt[i]; maxf = std::max(maxf, cnt[i]); }
    *n_out = total;
    if (total > max_out) return fail(VNF_E_CAPACITY, "vnf_mtcnn_detect: more faces than max_out");
    int o = 0;
    for (int i = 0; i < b; ++i)
      for (int k = 0; k < cnt[i]; ++k, ++o) {
        const float* f = &fin[((size_t)i * maxf + k) * 15];
        if (boxes) memcpy(boxes + (size_t)o * 4, f, 16);
        if (probs) probs[o] = f[4];
        if (points) memcpy(points + (size_t)o * 10, f + 5, 40);
      }
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

// One detection with HIP events between the cascade's stages (on the caller's stream): a text table, one line per
// stage "name ms algorithmic_bytes" (bytes 0 where the stage is not bandwidth-priced).  Synchronises.
extern "C" int vnf_mtcnn_stage_times(vnf_handle h, const uint8_t* frames, int b, int height, int width, char* report,
                                     int64_t capacity, void* stream) {
  try {
    HandleBase* hb = reinterpret_cast<HandleBase*>(h);
    if (!hb || hb->kind != 3) return fail(VNF_E_INVALID, "not an MTCNN handle");
    if (!frames || b <= 0 || !report || capacity <= 0) return fail(VNF_E_INVALID, "vnf_mtcnn_stage_times: bad argument");
    Mtcnn* m = static_cast<Mtcnn*>(hb);
    std::vector<int> cnt;
    std::vector<float> fin;
    StageProf prof;
    int r = mtcnn_run(m, frames, b, height, width, (hipStream_t)stream, cnt, fin, &prof);
    if (r != VNF_OK) return r;
    VNF_HIP(hipStreamSynchronize((hipStream_t)stream));
    std::string rep;
    char line[160];
    for (size_t i = 1; i < prof.ev.size(); ++i) {
      float ms = 0;
      VNF_HIP(hipEventElapsedTime(&ms, prof.ev[i - 1], prof.ev[i]));
      snprintf(line, sizeof line, "%s %.6f %.0f\n", prof.name[i].c_str(), ms, prof.bytes[i]);
      rep += line;
    }
    strncpy(report, rep.c_str(), (size_t)capacity - 1);
    report[capacity - 1] = 0;
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

extern "C" int vnf_mtcnn_results_device(vnf_handle h, int32_t* frame_idx, float* boxes, float* probs, float* points,
                                        int max_out, void* stream) {
  HandleBase* hb = reinterpret_cast<HandleBase*>(h);
  if (!hb || hb->kind != 3) return fail(VNF_E_INVALID, "not an MTCNN handle");
  Mtcnn* m = static_cast<Mtcnn*>(hb);
  if (max_out < 0) return fail(VNF_E_INVALID, "vnf_mtcnn_results_device: bad argument");
  if (m->last_b == 0 || max_out == 0) return VNF_OK;  // the last detection found nothing
  hipLaunchKernelGGL(results_device_kernel, dim3(m->last_b), dim3(64), 0, (hipStream_t)stream, m->fin, m->fin_cnt, max_out, m->keep,
                     frame_idx, boxes, probs, points);
  VNF_HIP(hipGetLastError());
  return VNF_OK;
}

// Staged-parity hook for the O-stage decode alone (detect_face.py:148-169 + mtcnn.py:334-340): runs stage3_post_kernel
// on a caller-made candidate table of ONE frame -- boxes (n,4) before bbreg and the O-Net outputs (n,15: face
// probability, 4 regression values, 5 x-landmarks, 5 y-landmarks) -- so a test can inject exactly tied scores.
// fin_out receives up to max_out rows [x1,y1,x2,y2,score, 10 landmark coordinates].  Synchronises.
extern "C" int vnf_mtcnn_debug_stage3(vnf_handle h, const float* boxes, const float* onet_out, int n, float* fin_out,
                                      int max_out, int32_t* n_out, void* stream) {
  try {
    HandleBase* hb = reinterpret_cast<HandleBase*>(h);
    if (!hb || hb->kind != 3) return fail(VNF_E_INVALID, "not an MTCNN handle");
    if (!boxes || !onet_out || n < 0 || !fin_out || !n_out) return fail(VNF_E_INVALID, "vnf_mtcnn_debug_stage3: bad argument");
    Mtcnn* m = static_cast<Mtcnn*>(hb);
    if (n > m->keep) return fail(VNF_E_CAPACITY, "vnf_mtcnn_debug_stage3: more rows than the handle's tables hold");
    hipStream_t s = (hipStream_t)stream;
    std::vector<Row> rows((size_t)std::max(n, 1));
    for (int i = 0; i < n; ++i) {
      Row r{};
      r.x1 = boxes[i * 4]; r.y1 = boxes[i * 4 + 1]; r.x2 = boxes[i * 4 + 2]; r.y2 = boxes[i * 4 + 3];
      rows[i] = r;
    }
    VNF_HIP(hipMemcpyAsync(m->rows3, rows.data(), (size_t)n * sizeof(Row), hipMemcpyHostToDevice, s));
    VNF_HIP(hipMemcpyAsync(m->oout, onet_out, (size_t)n * 15 * 4, hipMemcpyHostToDevice, s));
    VNF_HIP(hipMemcpyAsync(m->row3_cnt, &n, 4, hipMemcpyHostToDevice, s));
    VNF_HIP(hipMemsetAsync(m->status, 0, 4, s));
    const size_t lds_post = (size_t)KEEP * 28 + 256 * 20 + KEEP * 16;
    hipLaunchKernelGGL(stage3_post_kernel, dim3(1), dim3(256), lds_post, s, m->rows3, m->row3_cnt, m->oout, m->cfg.thresholds[2],
                       0.7f, m->cfg.select_largest, m->keep, m->fin, m->fin_cnt, m->status, m->scratch);
    VNF_HIP(hipGetLastError());
    int nk = 0;
    VNF_HIP(hipMemcpyAsync(&nk, m->fin_cnt, 4, hipMemcpyDeviceToHost, s));
    VNF_HIP(hipStreamSynchronize(s));
    *n_out = nk;
    if (nk > max_out) return fail(VNF_E_CAPACITY, "vnf_mtcnn_debug_stage3: more rows than max_out");
    VNF_HIP(hipMemcpy(fin_out, m->fin, (size_t)nk * 15 * 4, hipMemcpyDeviceToHost));
    m->last_b = 0;
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

// Staged-parity hook: dense P-Net maps of one pyramid level for frame 0 of a batch (test use).
extern "C" int vnf_mtcnn_debug_pnet(vnf_handle h, const uint8_t* frames, int height, int width, int level,
                                    float* level_out, float* prob_out, float* reg_out, int32_t dims[4], void* stream) {
  try {
    HandleBase* hb = reinterpret_cast<HandleBase*>(h);
    if (!hb || hb->kind != 3) return fail(VNF_E_INVALID, "not an MTCNN handle");
    Mtcnn* m = static_cast<Mtcnn*>(hb);
    LevelTable t = make_levels(height, width, m->cfg.min_face_size, (double)m->cfg.factor);
    if (level < 0 || level >= t.n) return fail(VNF_E_INVALID, "no such level");
    float *pd = nullptr, *rd = nullptr;
    VNF_HIP(hipMalloc(&pd, (size_t)t.tot_out * 4 * m->cfg.max_batch));
    VNF_HIP(hipMalloc(&rd, (size_t)t.tot_out * 16 * m->cfg.max_batch));
    m->prob_dbg = pd; m->reg_dbg = rd;
    std::vector<int> cnt;
    std::vector<float> fin;
    int r = mtcnn_run(m, frames, 1, height, width, (hipStream_t)stream, cnt, fin);
    m->prob_dbg = nullptr; m->reg_dbg = nullptr;
    if (r == VNF_OK) {
      const LevelDesc& L = t.l[level];
      dims[0] = L.Hs; dims[1] = L.Ws; dims[2] = L.oh; dims[3] = L.ow;
      hipError_t e = hipSuccess;
      if (level_out)
        for (int c = 0; c < 3 && e == hipSuccess; ++c)
          e = hipMemcpy(level_out + (size_t)c * L.Hs * L.Ws, m->lvl + (size_t)c * t.tot_px + L.off_px, (size_t)L.Hs * L.Ws * 4, hipMemcpyDeviceToHost);
      if (prob_out && e == hipSuccess) e = hipMemcpy(prob_out, pd + L.off_out, (size_t)L.oh * L.ow * 4, hipMemcpyDeviceToHost);
      if (reg_out)
        for (int c = 0; c < 4 && e == hipSuccess; ++c)
          e = hipMemcpy(reg_out + (size_t)c * L.oh * L.ow, rd + (size_t)c * t.tot_out + L.off_out, (size_t)L.oh * L.ow * 4, hipMemcpyDeviceToHost);
      if (e != hipSuccess) r = fail(VNF_E_HIP, hipGetErrorString(e));
    }
    (void)hipFree(pd);
    (void)hipFree(rd);
    return r;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}
