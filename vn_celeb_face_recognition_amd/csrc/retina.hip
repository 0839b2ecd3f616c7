// RetinaFace detector (mobilenet0.25 backbone) on the device: /root/reference/models/retina_face.py:56-232 --
// the network (MobileNetV1-0.25 body, FPN, 3 x SSH, class / bbox / landmark heads: retina_face_utils/components.py) runs as a
// plan on the exact-f32 MFMA core (engine.cpp build_retina_mnet); this file holds what surrounds it:
//
//   retina_prep_kernel     u8 RGB frames -> NHWC4 fp32, channel means (104, 117, 123) subtracted (retina_face.py:158-164)
//   retina_score_kernel    softmax over the 2 classes of every anchor, threshold conf_thres, compaction (191-195)
//   retina_select_kernel   per frame: top-K by score (198-201), prior-box decode of boxes (prior_box.py:20-34,
//                          box_utils.py:209-227), py_cpu_nms (nms/py_cpu_nms.py:10-37: IoU with +1 widths), keep_top_k,
//                          vis_thres filter, landmark decode (box_utils.py:229-247) -> rows [x1,y1,x2,y2,score, 5 x (x,y)]
//
// Score ties: the reference orders candidates with scores.argsort()[::-1] twice (once for the top-K cut, once inside
// py_cpu_nms) -- NumPy's default unstable sort.  Pinned here (and in the oracle's ties="table"): a STABLE ascending sort,
// reversed, both times; i.e. among equal scores the top-K cut prefers the higher anchor index and the NMS then visits the
// lower anchor index first.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"
#include "nms_device.h"
#include "split_f16.h"

namespace vnf {

constexpr int RCAP = 16384;   // anchors above conf_thres per frame (sort keys live in LDS: 128 KiB)
constexpr int RKEEP = 768;    // >= keep_top_k (750)

struct RetinaGeom {
  int H, W, n_anchor;
  int split;                      // the head maps hold split-f16 pairs (F16X2 plan) instead of fp32
  int fh[3], fw[3], off[3];       // feature map sizes, first anchor index of each level
  const float* head[3];           // (B, fh, fw, 32) fp32: [cls 4 | bbox 8 | landmark 20]
};

struct RCand { float score; int anchor; };

__device__ __forceinline__ float hval(const RetinaGeom& g, const float* p) { return g.split ? (float)__builtin_bit_cast(sf16, *p) : *p; }

__device__ __forceinline__ void anchor_loc(const RetinaGeom& g, int a, int& l, int& cell, int& k) {
  l = a >= g.off[2] ? 2 : (a >= g.off[1] ? 1 : 0);
  const int r = a - g.off[l];
  cell = r >> 1;
  k = r & 1;
}

// prior box (cx, cy, s_kx, s_ky) of anchor (level l, cell, k): prior_box.py:20-34 in Python-double arithmetic, rounded to
// fp32 as torch.Tensor(anchors) does
__device__ __forceinline__ float4 prior_of(const RetinaGeom& g, int l, int cell, int k) {
  const int i = cell / g.fw[l], j = cell - i * g.fw[l];
  const double step = (double)(8 << l), ms = (double)((16 << (2 * l)) << k);
  return float4{(float)(((double)j + 0.5) * step / (double)g.W), (float)(((double)i + 0.5) * step / (double)g.H),
                (float)(ms / (double)g.W), (float)(ms / (double)g.H)};
}

__device__ __forceinline__ float4 decode_box(const RetinaGeom& g, int img, int a) {
  int l, cell, k;
  anchor_loc(g, a, l, cell, k);
  const float4 p = prior_of(g, l, cell, k);
  const float* h = g.head[l] + ((size_t)img * g.fh[l] * g.fw[l] + cell) * 32 + 4 + 4 * k;
  const float bx = p.x + (hval(g, h) * 0.1f) * p.z, by = p.y + (hval(g, h + 1) * 0.1f) * p.w;
  const float bw = p.z * expf(hval(g, h + 2) * 0.2f), bh = p.w * expf(hval(g, h + 3) * 0.2f);
  const float x1 = bx - bw / 2.f, y1 = by - bh / 2.f;
  const float x2 = bw + x1, y2 = bh + y1;
  return float4{x1 * (float)g.W, y1 * (float)g.H, x2 * (float)g.W, y2 * (float)g.H};
}

__global__ void retina_prep_kernel(const uint8_t* __restrict__ frames, float* __restrict__ x, size_t npix, int split) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const uint8_t* p = frames + i * 3;
    float4 v = float4{(float)p[0] - 104.f, (float)p[1] - 117.f, (float)p[2] - 123.f, 0.f};
    if (split)   // small integers: exact in the hi half
      v = float4{__builtin_bit_cast(float, sf16(v.x)), __builtin_bit_cast(float, sf16(v.y)), __builtin_bit_cast(float, sf16(v.z)), 0.f};
    *reinterpret_cast<float4*>(x + i * 4) = v;
  }
}

__global__ void retina_score_kernel(RetinaGeom g, int B, float conf_thres, RCand* __restrict__ cand, int* __restrict__ cnt,
                                    int* __restrict__ status) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, img = blockIdx.y;
  if (a >= g.n_anchor) return;
  int l, cell, k;
  anchor_loc(g, a, l, cell, k);
  const float* h = g.head[l] + ((size_t)img * g.fh[l] * g.fw[l] + cell) * 32 + 2 * k;
  const float l0 = hval(g, h), l1 = hval(g, h + 1);
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  const float score = e1 / (e0 + e1);
  if (score > conf_thres) {
    const int slot = atomicAdd(&cnt[img], 1);
    if (slot < RCAP) cand[(size_t)img * RCAP + slot] = RCand{score, a};
    else atomicOr(status, ST_OVER_IMG);
  }
}

// one workgroup per frame
__global__ void __launch_bounds__(256) retina_select_kernel(RetinaGeom g, const RCand* __restrict__ cand, const int* __restrict__ cnt,
                                                            int topk, float nms_thres, int keep_top_k, float vis_thres,
                                                            float* __restrict__ fin, int* __restrict__ fin_cnt, int* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);                 // RCAP * 8
  float4* s_kbox = reinterpret_cast<float4*>(smem + RCAP * 8);                             // RKEEP * 16
  int* s_keep = reinterpret_cast<int*>(smem + RCAP * 8 + RKEEP * 16);                      // RKEEP * 4
  float4* s_cbox = reinterpret_cast<float4*>(smem + RCAP * 8 + RKEEP * 20);                // 256 * 16
  int* s_alive = reinterpret_cast<int*>(smem + RCAP * 8 + RKEEP * 20 + 256 * 16);          // 256 * 4
  const int img = blockIdx.x;
  const int n = min(cnt[img], RCAP);
  float* out = fin + (size_t)img * RKEEP * 15;
  if (n == 0) {
    if (threadIdx.x == 0) fin_cnt[img] = 0;
    return;
  }
  const RCand* c = cand + (size_t)img * RCAP;
  // (1) descending score; among equal scores the HIGHER anchor index first (stable ascending argsort, reversed)
  const int npad = next_pow2(n);
  for (int i = threadIdx.x; i < npad; i += blockDim.x)
    keys[i] = i < n ? ((unsigned long long)inv_score_bits(c[i].score) << 32) | (0xFFFFFFFFu - (unsigned)c[i].anchor) : ~0ull;
  __syncthreads();
  block_sort(keys, n, npad);
  // (2) the top-K survive; py_cpu_nms re-sorts them: equal scores now LOWER anchor index first
  const int K = min(n, topk);
  const int kpad = next_pow2(K);
  for (int i = threadIdx.x; i < npad; i += blockDim.x) {
    const unsigned long long key = keys[i];
    keys[i] = i < K ? (key & 0xFFFFFFFF00000000ull) | (0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu)) : ~0ull;
  }
  __syncthreads();
  block_sort(keys, K, kpad);
  auto anchor_of = [&](int r) { return (int)(keys[r] & 0xFFFFFFFFu); };
  auto getbox = [&](int r) { return decode_box(g, img, anchor_of(r)); };
  // dets are float32 rows [box, score]: the NMS arithmetic is fp32 like py_cpu_nms on that array
  const int nk_all = block_greedy_nms<NMS_IOU1>(K, nms_thres, getbox, s_keep, s_kbox, keep_top_k, s_cbox, s_alive, scratch);
  __syncthreads();
  const int nk = min(nk_all, keep_top_k);
  // kept rows are in descending score order, so "score >= vis_thres" keeps a prefix
  __shared__ int s_nv;
  if (threadIdx.x == 0) s_nv = 0;
  __syncthreads();
  for (int q = threadIdx.x; q < nk; q += blockDim.x) {
    const unsigned long long key = keys[s_keep[q]];
    const float score = __uint_as_float(0xFFFFFFFFu - (unsigned)(key >> 32));
    if (score >= vis_thres) {
      atomicAdd(&s_nv, 1);
      const int a = (int)(key & 0xFFFFFFFFu);
      int l, cell, k;
      anchor_loc(g, a, l, cell, k);
      const float4 p = prior_of(g, l, cell, k);
      const float* h = g.head[l] + ((size_t)img * g.fh[l] * g.fw[l] + cell) * 32 + 12 + 10 * k;
      const float4 b = s_kbox[q];
      float* o = out + (size_t)q * 15;
      o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = score;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        o[5 + 2 * j] = (p.x + (hval(g, h + 2 * j) * 0.1f) * p.z) * (float)g.W;
        o[6 + 2 * j] = (p.y + (hval(g, h + 2 * j + 1) * 0.1f) * p.w) * (float)g.H;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) fin_cnt[img] = s_nv;
}

struct Retina : HandleBase {
  vnf_retina_cfg cfg;
  Encoder* enc = nullptr;
  RetinaGeom geom;
  int head_bufs[3];
  RCand* cand = nullptr;
  int *cnt = nullptr, *fin_cnt = nullptr, *status = nullptr, *scratch = nullptr;
  float* fin = nullptr;
  int last_b = 0;
  ~Retina() override { delete enc; }
};

// device-resident copy of the last detection, frames concatenated in order
__global__ void retina_results_kernel(const float* __restrict__ fin, const int* __restrict__ fin_cnt, int max_out,
                                      int32_t* __restrict__ fidx, float* __restrict__ boxes, float* __restrict__ probs,
                                      float* __restrict__ points) {
  const int img = blockIdx.x;
  int off = 0;
  for (int i = 0; i < img; ++i) off += fin_cnt[i];
  const int c = fin_cnt[img];
  for (int k = threadIdx.x; k < c; k += blockDim.x) {
    const int o = off + k;
    if (o >= max_out) break;
    const float* f = fin + ((size_t)img * RKEEP + k) * 15;
    if (fidx) fidx[o] = img;
    if (boxes) { boxes[o * 4] = f[0]; boxes[o * 4 + 1] = f[1]; boxes[o * 4 + 2] = f[2]; boxes[o * 4 + 3] = f[3]; }
    if (probs) probs[o] = f[4];
    if (points)
      for (int j = 0; j < 10; ++j) points[o * 10 + j] = f[5 + j];
  }
}

}  // namespace vnf
using namespace vnf;

extern "C" int vnf_retina_create(const vnf_tensor_desc* weights, int n_weights, const vnf_retina_cfg* cfg, vnf_handle* out) {
  try {
    if (!weights || !cfg || !out || cfg->max_batch < 1 || cfg->height < 32 || cfg->width < 32 || cfg->keep_top_k < 1 ||
        cfg->keep_top_k > RKEEP || cfg->topk_bf_nms < 1)
      return fail(VNF_E_INVALID, "vnf_retina_create: bad configuration (keep_top_k <= 768, frames >= 32 px)");
    *out = nullptr;
    Retina* r = new Retina();
    r->kind = 5;
    r->cfg = *cfg;
    (void)hipGetDevice(&r->device);
    WeightMap wm(weights, n_weights);
    r->enc = new Encoder();
    Encoder& e = *r->enc;
    if (cfg->compute_dtype != VNF_F32 && cfg->compute_dtype != VNF_F16X2) {
      delete r;
      return fail(VNF_E_INVALID, "vnf_retina_create: compute_dtype must be VNF_F32 or VNF_F16X2");
    }
    e.kind = 1; e.arch = -5; e.dtype = cfg->compute_dtype == VNF_F16X2 ? F16X2 : F32; e.max_batch = cfg->max_batch; e.max_streams = 1;
    int rc = build_retina_mnet(e, wm, cfg->height, cfg->width, r->head_bufs);
    if (rc == VNF_OK) rc = e.finalize();
    if (rc != VNF_OK) { delete r; return rc; }
    RetinaGeom& g = r->geom;
    g.H = cfg->height; g.W = cfg->width;
    int acc = 0;
    for (int l = 0; l < 3; ++l) {
      const Buf& hb = e.bufs[r->head_bufs[l]];
      g.fh[l] = hb.H; g.fw[l] = hb.W; g.off[l] = acc;
      acc += hb.H * hb.W * 2;
      g.head[l] = (const float*)hb.ptr;
    }
    g.n_anchor = acc;
    g.split = e.dtype == F16X2 ? 1 : 0;
    const size_t B = cfg->max_batch;
    r->cand = (RCand*)r->dalloc(B * RCAP * sizeof(RCand));
    r->cnt = (int*)r->dalloc((2 * B + 8) * 4);
    r->fin = (float*)r->dalloc(B * RKEEP * 15 * 4);
    if (!r->cand || !r->cnt || !r->fin) { delete r; return VNF_E_HIP; }
    r->fin_cnt = r->cnt + B;
    r->status = r->fin_cnt + B;
    r->scratch = r->status + 1;
    const int lds = RCAP * 8 + RKEEP * 20 + 256 * 20;
    (void)hipFuncSetAttribute((const void*)retina_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipGetLastError();
    VNF_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<vnf_handle>(static_cast<HandleBase*>(r));
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

extern "C" int vnf_retina_detect(vnf_handle h, const uint8_t* frames, int b, int height, int width, int32_t* counts, float* boxes,
                                 float* probs, float* points, int max_out, int32_t* n_out, void* stream) {
  try {
    HandleBase* hb = reinterpret_cast<HandleBase*>(h);
    if (!hb || hb->kind != 5) return fail(VNF_E_INVALID, "not a RetinaFace handle");
    Retina* r = static_cast<Retina*>(hb);
    if (!frames || b <= 0 || !counts || !n_out) return fail(VNF_E_INVALID, "vnf_retina_detect: bad argument");
    if (b > r->cfg.max_batch || height != r->cfg.height || width != r->cfg.width)
      return fail(VNF_E_CAPACITY, "vnf_retina_detect: the handle was created for another frame size / batch");
    hipStream_t s = (hipStream_t)stream;
    Encoder& e = *r->enc;
    // the activation contexts of the plan are not used here: buffer pointers are fixed after finalize()
    RetinaGeom g = r->geom;
    for (int l = 0; l < 3; ++l) g.head[l] = (const float*)e.bufs[r->head_bufs[l]].ptr;
    const bool stem_in_plan = !e.ops.empty() && e.ops[0].kind == Op::RSTEM;   // conv0 reads the u8 frames itself
    if (!stem_in_plan) {
      const size_t npix = (size_t)b * height * width;
      hipLaunchKernelGGL(retina_prep_kernel, dim3((unsigned)std::min<size_t>((npix + 255) / 256, 16384)), dim3(256), 0, s, frames,
                         (float*)e.bufs[0].ptr, npix, e.dtype == F16X2 ? 1 : 0);
      VNF_HIP(hipGetLastError());
    }
    static const bool layers = getenv("VNF_RETINA_LAYERS") != nullptr;   // diagnostic: per-layer table on stderr
    std::string rep;
    int rc = e.run(stem_in_plan ? frames : nullptr, b, VNF_F32, nullptr, s, layers ? &rep : nullptr);
    if (rc != VNF_OK) return rc;
    if (!rep.empty()) fprintf(stderr, "%s", rep.c_str());
    VNF_HIP(hipMemsetAsync(r->cnt, 0, (2 * (size_t)r->cfg.max_batch + 8) * 4, s));
    hipLaunchKernelGGL(retina_score_kernel, dim3((g.n_anchor + 255) / 256, b), dim3(256), 0, s, g, b, r->cfg.conf_thres, r->cand, r->cnt,
                       r->status);
    const int lds = RCAP * 8 + RKEEP * 20 + 256 * 20;
    hipLaunchKernelGGL(retina_select_kernel, dim3(b), dim3(256), lds, s, g, r->cand, r->cnt, r->cfg.topk_bf_nms, r->cfg.nms_thres,
                       r->cfg.keep_top_k, r->cfg.vis_thres, r->fin, r->fin_cnt, r->scratch);
    VNF_HIP(hipGetLastError());
    r->last_b = b;
    // read-back: counts + status, then the rows (one synchronisation each)
    std::vector<int> hc((size_t)b + 1);
    VNF_HIP(hipMemcpyAsync(hc.data(), r->fin_cnt, (size_t)b * 4, hipMemcpyDeviceToHost, s));
    VNF_HIP(hipMemcpyAsync(&hc[b], r->status, 4, hipMemcpyDeviceToHost, s));
    VNF_HIP(hipStreamSynchronize(s));
    if (hc[b] & ST_OVER_IMG) return fail(VNF_E_CAPACITY, "vnf_retina_detect: more than 16384 anchors above conf_thres in a frame");
    int total = 0, maxf = 0;
    for (int i = 0; i < b; ++i) { counts[i] = hc[i]; total += hc[i]; maxf = std::max(maxf, hc[i]); }
    *n_out = total;
    if (total > max_out) return fail(VNF_E_CAPACITY, "vnf_retina_detect: more faces than max_out");
    if (total == 0) return VNF_OK;
    std::vector<float> rows((size_t)b * maxf * 15);
    VNF_HIP(hipMemcpy2DAsync(rows.data(), (size_t)maxf * 15 * 4, r->fin, (size_t)RKEEP * 15 * 4, (size_t)maxf * 15 * 4, b,
                             hipMemcpyDeviceToHost, s));
    VNF_HIP(hipStreamSynchronize(s));
    int o = 0;
    for (int i = 0; i < b; ++i)
      for (int k = 0; k < hc[i]; ++k, ++o) {
        const float* f = &rows[((size_t)i * maxf + k) * 15];
        if (boxes) memcpy(boxes + (size_t)o * 4, f, 16);
        if (probs) probs[o] = f[4];
        if (points) memcpy(points + (size_t)o * 10, f + 5, 40);
      }
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

extern "C" int vnf_retina_results_device(vnf_handle h, int32_t* frame_idx, float* boxes, float* probs, float* points, int max_out,
                                         void* stream) {
  HandleBase* hb = reinterpret_cast<HandleBase*>(h);
  if (!hb || hb->kind != 5) return fail(VNF_E_INVALID, "not a RetinaFace handle");
  Retina* r = static_cast<Retina*>(hb);
  if (max_out < 0) return fail(VNF_E_INVALID, "vnf_retina_results_device: bad argument");
  if (r->last_b == 0 || max_out == 0) return VNF_OK;
  hipLaunchKernelGGL(retina_results_kernel, dim3(r->last_b), dim3(64), 0, (hipStream_t)stream, r->fin, r->fin_cnt, max_out, frame_idx,
                     boxes, probs, points);
  VNF_HIP(hipGetLastError());
  return VNF_OK;
}

// staged parity: raw head outputs of pyramid level `level` of the last detection as a host (b, fh, fw, 32) fp32 array
extern "C" int vnf_retina_debug_heads(vnf_handle h, int level, int b, float* host_out, int64_t capacity, int32_t dims[2]) {
  HandleBase* hb = reinterpret_cast<HandleBase*>(h);
  if (!hb || hb->kind != 5 || level < 0 || level > 2) return fail(VNF_E_INVALID, "vnf_retina_debug_heads: bad argument");
  Retina* r = static_cast<Retina*>(hb);
  const Buf& bf = r->enc->bufs[r->head_bufs[level]];
  if (dims) { dims[0] = bf.H; dims[1] = bf.W; }
  const int64_t total = (int64_t)b * bf.H * bf.W * 32;
  if (!host_out || total > capacity || b > r->cfg.max_batch) return fail(VNF_E_CAPACITY, "vnf_retina_debug_heads: capacity");
  VNF_HIP(hipDeviceSynchronize());
  VNF_HIP(hipMemcpy(host_out, bf.ptr, (size_t)total * 4, hipMemcpyDeviceToHost));
  if (r->enc->dtype == F16X2)
    for (int64_t i = 0; i < total; ++i) {
      sf16 v;
      memcpy(&v, host_out + i, 4);
      host_out[i] = (float)v;
    }
  return VNF_OK;
}
