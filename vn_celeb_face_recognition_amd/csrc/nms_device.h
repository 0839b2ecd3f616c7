// Block-level sort and greedy NMS shared by the detectors (mtcnn.hip, retina.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace vnf {

enum { ST_OVER_SCALE = 1, ST_OVER_IMG = 2, ST_OVER_KEEP = 4, ST_DEGENERATE = 8 };

// overlap tests: NMS_TV = torchvision.ops.nms (IoU, areas without +1, suppress if > thr); NMS_MIN = detect_face.py
// nms_numpy 'Min' (inter / min area, +1 widths, keep if <= thr); NMS_IOU1 = py_cpu_nms (IoU with +1 widths, keep if <= thr)
enum { NMS_TV = 0, NMS_MIN = 1, NMS_IOU1 = 2 };

__device__ __forceinline__ unsigned inv_score_bits(float s) { return 0xFFFFFFFFu - __float_as_uint(s); }  // s >= 0

__device__ inline void block_bitonic_sort(unsigned long long* keys, int npad) {
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = ((i & k) == 0);
          const unsigned long long a = keys[i], b = keys[ixj];
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
}

// Rank sort of n <= E * blockDim.x keys (ascending; unique except for the sentinel ~0, which may repeat): every thread
// counts the keys below each of its E elements (one broadcast LDS read per key, no barrier inside) and scatters them
// to their ranks; the sentinels end up behind the real keys.  For the few hundred candidates
// a frame usually has this is several times cheaper than the log^2 barrier passes of the bitonic network.
template <int E>
__device__ inline void block_rank_sort(unsigned long long* keys, int n) {
  unsigned long long mine[E];
  int rank[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x + e * blockDim.x;
    mine[e] = i < n ? keys[i] : ~0ull;
    rank[e] = 0;
  }
  for (int j = 0; j < n; ++j) {
    const unsigned long long kj = keys[j];
#pragma unroll
    for (int e = 0; e < E; ++e) rank[e] += kj < mine[e] ? 1 : 0;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e)
    if ((int)(threadIdx.x + e * blockDim.x) < n) keys[threadIdx.x + e * blockDim.x] = ~0ull;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (mine[e] != ~0ull) keys[rank[e]] = mine[e];
  __syncthreads();
}

// ascending sort of keys[0, n) (unique keys; keys[n, npad) hold ~0 and stay in place)
__device__ inline void block_sort(unsigned long long* keys, int n, int npad) {
  const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
  if (per <= 1) block_rank_sort<1>(keys, n);
  else if (per == 2) block_rank_sort<2>(keys, n);
  else if (per <= 4) block_rank_sort<4>(keys, n);
  else block_bitonic_sort(keys, npad);
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

template <int MIN_MODE>
__device__ __forceinline__ bool overlaps(const float4 a, float aa, const float4 b, float ab, float thr) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y), xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  if (MIN_MODE == NMS_MIN) {
    const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
    const float inter = w * h;
    return !(inter / fminf(aa, ab) <= thr);
  } else if (MIN_MODE == NMS_IOU1) {
    const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
    const float inter = w * h;
    return !(inter / (aa + ab - inter) <= thr);
  } else {
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    return inter / (aa + ab - inter) > thr;
  }
}
template <int MIN_MODE>
__device__ __forceinline__ float box_area(const float4 b) {
  return MIN_MODE != NMS_TV ? (b.z - b.x + 1.f) * (b.w - b.y + 1.f) : (b.z - b.x) * (b.w - b.y);
}

// Greedy NMS over n boxes already in visiting order.  getbox(rank) returns the box of the rank-th
// candidate.  Kept ranks are appended to s_keep (LDS, capacity keep_cap); returns the kept count
// (uniform).  s_kbox caches the kept boxes, s_cbox / s_alive hold the current chunk.  blockDim.x <= 256.
//
// A chunk of blockDim.x candidates is resolved in three parallel steps instead of one barrier per kept box:
// (1) every candidate is tested against the boxes kept by earlier chunks; (2) every survivor t builds the bit row
// "later survivors of this chunk that t would suppress"; (3) one lane walks the rows in order (a survivor not yet
// removed is kept and ORs its row into the removed set) -- the sequential greedy rule, at one LDS read per KEPT box.
template <int MIN_MODE, typename GetBox>
__device__ inline int block_greedy_nms(int n, float thr, GetBox getbox, int* s_keep, float4* s_kbox, int keep_cap,
                                float4* s_cbox, int* s_alive, int* status) {
  __shared__ unsigned long long s_row[256 * 4];
  __shared__ unsigned long long s_word[8];   // [0,4): survivors per wave, [4,8): kept per wave
  int nkeep = 0;
  const int t = threadIdx.x, BS = blockDim.x;
  for (int base = 0; base < n; base += BS) {
    const int r = base + t;
    const bool valid = r < n;
    float4 box = valid ? getbox(r) : float4{0.f, 0.f, 0.f, 0.f};
    const float area = box_area<MIN_MODE>(box);
    bool alive = valid;
    // four kept boxes per round (their broadcast LDS reads issue together); the wave leaves once none of its
    // candidates is alive.  The tail round re-tests the last kept box, which changes nothing.
    for (int k = 0; k < nkeep; k += 4) {
      if (__ballot(alive) == 0ull) break;
      float4 kb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kb[u] = s_kbox[min(k + u, nkeep - 1)];
      bool hit = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) hit = hit || overlaps<MIN_MODE>(kb[u], box_area<MIN_MODE>(kb[u]), box, area, thr);
      alive = alive && !hit;
    }
    s_cbox[t] = box;
    s_alive[t] = alive ? 1 : 0;
    const unsigned long long bal = __ballot(alive);
    if ((t & 63) == 0) s_word[t >> 6] = bal;
    if (t < 4 && t * 64 >= BS) s_word[t] = 0;
    __syncthreads();
    const int lim = min(BS, n - base);
    unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
    if (alive) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        // survivors after t in word w
        unsigned long long cand = s_word[w];
        if (w * 64 + 63 <= t) cand = 0;
        else if (w * 64 <= t) cand &= ~((2ull << (t & 63)) - 1ull);
        while (cand) {
          const int b = __builtin_ctzll(cand);
          cand &= cand - 1;
          const float4 cb = s_cbox[w * 64 + b];
          if (overlaps<MIN_MODE>(box, area, cb, box_area<MIN_MODE>(cb), thr)) m[w] |= 1ull << b;
        }
      }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) s_row[t * 4 + w] = m[w];
    __syncthreads();
    if (t == 0) {
      unsigned long long rem[4] = {0ull, 0ull, 0ull, 0ull}, kept[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned long long sv = s_word[w];
        unsigned long long avail = sv & ~rem[w];
        while (avail) {
          const int b = __builtin_ctzll(avail);
          kept[w] |= 1ull << b;
          const unsigned long long* row = s_row + (w * 64 + b) * 4;
#pragma unroll
          for (int v = 0; v < 4; ++v) rem[v] |= row[v];
          avail = sv & ~rem[w] & ~((2ull << b) - 1ull);
        }
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) s_word[4 + w] = kept[w];
    }
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned long long kw = s_word[4 + w];
      const int pc = __popcll(kw);
      total += pc;
      if (w < (t >> 6)) before += pc;
      else if (w == (t >> 6)) before += __popcll(kw & ((1ull << (t & 63)) - 1ull));
    }
    const bool kept_me = (s_word[4 + (t >> 6)] >> (t & 63)) & 1ull;
    if (kept_me) {
      const int pos = nkeep + before;
      if (pos < keep_cap) { s_keep[pos] = r; s_kbox[pos] = box; }
    }
    if (t == 0 && nkeep + total > keep_cap) atomicOr(status, ST_OVER_KEEP);
    nkeep = min(keep_cap, nkeep + total);
    (void)lim;
    __syncthreads();
  }
  return nkeep;
}


}  // namespace vnf
