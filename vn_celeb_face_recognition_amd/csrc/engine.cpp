// Encoder engine: folds BatchNorm into packed MFMA-ready weights, lays activations out as NHWC
// slices of a small set of device buffers (concat-free inception branches), and replays a
// static plan of convolution / pooling launches on the caller's stream.
#include "engine.h"
#include "split_f16.h"
#include "block35.h"
#include "stem_mid.h"
#include "trunk17.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace vnf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
const char* last_error_cstr() { return g_err.c_str(); }

HandleBase::~HandleBase() {
  for (void* p : allocs) (void)hipFree(p);
}
void* HandleBase::dalloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
    return nullptr;
  }
  allocs.push_back(p);
  return p;
}
void* HandleBase::upload(const void* host, size_t bytes) {
  void* p = dalloc(bytes);
  if (!p) return nullptr;
  hipError_t e = hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error(std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

WeightMap::WeightMap(const vnf_tensor_desc* w, int n) {
  for (int i = 0; i < n; ++i)
    if (w[i].name) m[w[i].name] = &w[i];
}
const float* WeightMap::get(const std::string& name, int64_t numel) {
  auto it = m.find(name);
  if (it == m.end() || it->second->dtype != VNF_F32 || !it->second->data) {
    if (missing.empty()) missing = name;
    return nullptr;
  }
  int64_t n = 1;
  for (int i = 0; i < it->second->ndim; ++i) n *= it->second->shape[i];
  if (n != numel) {
    if (missing.empty()) missing = name + " (unexpected size)";
    return nullptr;
  }
  return (const float*)it->second->data;
}

static inline uint16_t f2bf16(float f) {  // round to nearest even, NaN preserved
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
void convert_to(int dtype, const float* src, void* dst, size_t n) {
  if (dtype == F32) {
    memcpy(dst, src, n * 4);
  } else if (dtype == BF16) {
    uint16_t* d = (uint16_t*)dst;
    for (size_t i = 0; i < n; ++i) d[i] = f2bf16(src[i]);
  } else if (dtype == F16X2) {
    sf16* d = (sf16*)dst;
    for (size_t i = 0; i < n; ++i) d[i] = sf16(src[i]);
  } else if (dtype == F16P) {
    // planar split-f16 WEIGHTS: per K tile of 32 k values [32 hi][32 lo] -- the 128-byte LDS row of a K tile is its
    // four hi chunks followed by its four lo chunks (conv_device.h); n is a whole number of K tiles
    _Float16* d = (_Float16*)dst;
    for (size_t t = 0; t + 32 <= n; t += 32)
      for (int i = 0; i < 32; ++i) {
        const sf16 v(src[t + i]);
        d[2 * t + i] = v.hi;
        d[2 * t + 32 + i] = v.lo;
      }
  } else {
    _Float16* d = (_Float16*)dst;
    for (size_t i = 0; i < n; ++i) d[i] = (_Float16)src[i];
  }
}

// ---------------------------------------------------------------------------------------------
int Encoder::add_buf(int H, int W, int C) {
  Buf b;
  b.H = H; b.W = W; b.C = C;
  bufs.push_back(b);
  return (int)bufs.size() - 1;
}

ConvArgs Encoder::conv_args(const ConvLayer& L, int n0, int nn) const {
  const int es = dtype_size(dtype);
  const Buf& xb = bufs[L.x_buf];
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.dtype = dtype;
  a.x = xb.ptr + ((size_t)n0 * xb.elems_per_image() + L.x_coff) * es;
  a.ldx = xb.C; a.H = L.H; a.W = L.W; a.Cin = L.cin; a.Ho = L.Ho; a.Wo = L.Wo;
  a.KH = L.KH; a.KW = L.KW; a.sh = L.sh; a.sw = L.sw; a.ph = L.ph; a.pw = L.pw;
  a.w = L.w; a.K = L.K; a.Kpad = L.Kpad; a.bias = L.bias; a.ncls = L.ncls; a.cout_pad = L.cout_pad;
  a.ktab = L.ktab; a.M = nn * L.Ho * L.Wo; a.Cout = L.cout; a.nseg = L.nseg;
  for (int i = 0; i < L.nseg; ++i) {
    a.seg[i].c0 = L.seg[i].c0; a.seg[i].c1 = L.seg[i].c1;
    if (L.seg[i].buf == -2) {
      a.seg[i].ptr = emb_raw + (size_t)n0 * 512;
      a.seg[i].ld = 512;
    } else {
      const Buf& ob = bufs[L.seg[i].buf];
      a.seg[i].ptr = ob.ptr + ((size_t)n0 * ob.elems_per_image() + L.seg[i].coff) * (L.out_f32 ? 4 : es);
      a.seg[i].ld = ob.C;
    }
  }
  if (L.res_buf >= 0) {
    const Buf& rb = bufs[L.res_buf];
    a.res = rb.ptr + ((size_t)n0 * rb.elems_per_image() + L.res_coff) * es;
    a.ldres = rb.C;
  }
  a.act = L.act; a.slope = L.slope; a.out_f32 = L.out_f32;
  a.cfg = L.cfg;
  return a;
}

// Pick each convolution's tile configuration by timing the candidates on this device at the
// batch size it will see (measure, don't guess: the best tile depends on M, N, K, the number of
// workgroups and where the operands sit in the cache hierarchy).  ~1 s at create time.
int Encoder::autotune() {
  static const int enabled = getenv("VNF_AUTOTUNE") ? atoi(getenv("VNF_AUTOTUNE")) : 1;
  static const int force = getenv("VNF_FORCE_CFG") ? atoi(getenv("VNF_FORCE_CFG")) : -2;
  if (!enabled && force < -1) return VNF_OK;
  // VNF_TUNE_CACHE=<file>: reuse the choices of an earlier create on this device (lines "key cfg"); lets a
  // profiled run show steady-state launches only and brings create time down to the weight upload
  std::map<std::string, int> cache;
  const char* cache_path = getenv("VNF_TUNE_CACHE");
  bool cache_dirty = false;
  if (cache_path && enabled) {
    if (FILE* f = fopen(cache_path, "r")) {
      char key[256];
      int c;
      while (fscanf(f, "%255s %d", key, &c) == 2) cache[key] = c;
      fclose(f);
    }
  }
  // tuning launches scribble over the activation buffers: nothing of an earlier vnf_embed may still be in flight,
  // and nothing of the tuner when the caller's launches start
  VNF_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  VNF_HIP(hipEventCreate(&e0));
  VNF_HIP(hipEventCreate(&e1));
  // tune_lanes > 1: every candidate is timed as `tune_lanes` concurrent copies on separate streams -- the state the
  // layer actually runs in when independent batches overlap (activation contexts): alone on the GPU a small tile with
  // many workgroups looks best, beside other kernels the tile that moves fewer bytes per FLOP does
  static const int env_lanes = getenv("VNF_TUNE_LANES") ? atoi(getenv("VNF_TUNE_LANES")) : 0;
  const int lanes = env_lanes > 0 ? (env_lanes > 4 ? 4 : env_lanes) : (tune_lanes < 1 ? 1 : tune_lanes);
  hipStream_t lane_s[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t lane_e[4] = {nullptr, nullptr, nullptr, nullptr};
  if (lanes > 1)
    for (int l = 0; l < lanes; ++l) {
      VNF_HIP(hipStreamCreateWithFlags(&lane_s[l], hipStreamNonBlocking));
      VNF_HIP(hipEventCreate(&lane_e[l]));
    }
  for (const Group& g : groups) {
    int part = (max_batch >= 192 && max_streams > 1) ? (max_batch + 1) / 2 : max_batch;  // run() cuts the batch over 2 streams
    if (tune_batch > 0 && tune_batch < part) part = tune_batch;
    const int nn = g.chunk < part ? g.chunk : part;
    for (int oi = g.first; oi < g.last; ++oi) {
      if (ops[oi].kind != Op::CONV) continue;
      bool in_fused = false;
      for (const FusedStack& f : fused) in_fused |= f.active && oi >= f.first && oi < (f.ext ? f.ext_last : f.last);
      if (in_fused) continue;  // replaced by a persistent kernel: nothing to tune
      ConvLayer& L = convs[ops[oi].a];
      float best = 1e30f;
      int best_cfg = -1;
      char key[256];
      snprintf(key, sizeof key, "%s/d%d/n%d/M%d/K%d/N%d/v%d/L%d", L.name.c_str(), dtype, nn, nn * L.Ho * L.Wo, L.Kpad, L.cout,
               conv_num_cfgs(), lanes);
      const auto hit = cache.find(key);
      if (hit != cache.end()) {
        ConvArgs a = conv_args(L, 0, nn);
        if (hit->second == -1 || conv_cfg_ok(a, hit->second)) { L.cfg = hit->second; continue; }
      }
      std::vector<std::pair<float, int>> timed;   // (ms per 4 launches, cfg) of every candidate
      // time one candidate: the minimum over `trials` of `reps` back-to-back launches (per lane), scaled to 4 launches
      auto time_cfg = [&](const ConvArgs& a, int trials, int reps, float* out_ms) -> int {
        float ms = 1e30f;
        for (int trial = 0; trial < trials; ++trial) {
          float t = 0;
          if (lanes <= 1) {
            VNF_HIP(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) (void)launch_conv(a, 0);
            VNF_HIP(hipEventRecord(e1, 0));
            VNF_HIP(hipEventSynchronize(e1));
            VNF_HIP(hipEventElapsedTime(&t, e0, e1));
          } else {
            VNF_HIP(hipDeviceSynchronize());
            VNF_HIP(hipEventRecord(e0, lane_s[0]));
            for (int r = 0; r < reps; ++r)
              for (int l = 0; l < lanes; ++l) (void)launch_conv(a, lane_s[l]);
            for (int l = 0; l < lanes; ++l) VNF_HIP(hipEventRecord(lane_e[l], lane_s[l]));
            for (int l = 0; l < lanes; ++l) {
              float tl = 0;
              VNF_HIP(hipEventSynchronize(lane_e[l]));
              VNF_HIP(hipEventElapsedTime(&tl, e0, lane_e[l]));
              if (tl > t) t = tl;
            }
          }
          t *= 4.f / reps;
          if (t < ms) ms = t;
        }
        *out_ms = ms;
        return VNF_OK;
      };
      static const int logit = getenv("VNF_AUTOTUNE_LOG") ? atoi(getenv("VNF_AUTOTUNE_LOG")) : 0;
      for (int cfg = -1; enabled && cfg < conv_num_cfgs(); ++cfg) {
        ConvArgs a = conv_args(L, 0, nn);
        a.cfg = cfg;
        if (cfg >= 0 && !conv_cfg_ok(a, cfg)) continue;
        if (launch_conv(a, 0) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 1e30f;
        const int rc = time_cfg(a, 2, 4, &ms);
        if (rc != VNF_OK) return rc;
        timed.emplace_back(ms, cfg);
        if (ms < best) { best = ms; best_cfg = cfg; }
        if (logit) fprintf(stderr, "autotune %s cfg %d: %.4f ms\n", L.name.c_str(), cfg, ms / 4);
      }
      // finalists: the first pass is 8 launches per candidate and two candidates a few per cent apart change places from
      // run to run; the ones within 8 % of the best are timed again, longer (VNF_TUNE_FINAL=0: first pass only)
      static const bool finals = !(getenv("VNF_TUNE_FINAL") && atoi(getenv("VNF_TUNE_FINAL")) == 0);
      if (finals && timed.size() > 1) {
        std::sort(timed.begin(), timed.end());
        float fbest = 1e30f;
        int fcfg = best_cfg, nfin = 0;
        for (const auto& tc : timed) {
          if (tc.first > timed[0].first * 1.08f || nfin == 4) break;
          ++nfin;
          ConvArgs a = conv_args(L, 0, nn);
          a.cfg = tc.second;
          float ms = 1e30f;
          const int rc = time_cfg(a, 3, 8, &ms);
          if (rc != VNF_OK) return rc;
          if (logit) fprintf(stderr, "autotune %s final cfg %d: %.4f ms\n", L.name.c_str(), tc.second, ms / 4);
          if (ms < fbest) { fbest = ms; fcfg = tc.second; }
        }
        if (nfin > 1) { best = fbest; best_cfg = fcfg; }
      }
      L.cfg = best_cfg;
      if (cache_path && enabled) { cache[key] = best_cfg; cache_dirty = true; }
      if (force >= -1) {
        ConvArgs a = conv_args(L, 0, nn);
        if (force == -1 || conv_cfg_ok(a, force)) L.cfg = force;
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int l = 0; l < 4; ++l) {
    if (lane_e[l]) (void)hipEventDestroy(lane_e[l]);
    if (lane_s[l]) (void)hipStreamDestroy(lane_s[l]);
  }
  VNF_HIP(hipDeviceSynchronize());
  if (cache_dirty) {
    if (FILE* f = fopen(cache_path, "w")) {
      for (auto& kv : cache) fprintf(f, "%s %d\n", kv.first.c_str(), kv.second);
      fclose(f);
    }
  }
  return VNF_OK;
}

int Encoder::finalize() {
  const int es = dtype_size(dtype);
  for (auto& b : bufs) {
    const size_t bytes = b.elems_per_image() * es * (size_t)max_batch;
    b.ptr = (char*)dalloc(bytes);
    if (!b.ptr) return VNF_E_HIP;
    VNF_HIP(hipMemset(b.ptr, 0, bytes));
  }
  emb_raw = (float*)dalloc((size_t)max_batch * 512 * 4);
  if (!emb_raw) return VNF_E_HIP;
  macs_alg = macs_exec = 0;
  for (auto& c : convs) { macs_alg += c.macs_alg; macs_exec += c.macs_exec; }
  if (groups.empty()) groups.push_back({0, (int)ops.size(), 1 << 30});
  {
    const int rc = prepare_fused();
    if (rc != VNF_OK) return rc;
  }
  tune_dirty = true;  // the first run() picks the tiles (after any set_streams / set_contexts of the caller)
  return VNF_OK;
}

// Fused stacks: the per-wave weight streams are gathered on the device from the packed per-convolution weights the
// plan already uploaded (same folding, same k order), biases are concatenated per block.
int Encoder::prepare_fused() {
  // bit 0: Block17 stack, bit 1: Block35, bit 2: stem 2a+2b+pool, bit 3: conv2d_3b inside the stem kernel, bit 4: the five
  // Block35 in one launch (with bit 1), bit 5 (off by default: measured at parity with the separate launch):
  // mixed_6a.branch1.0 inside that launch; read at create time
  const int enabled = getenv("VNF_FUSE") ? atoi(getenv("VNF_FUSE")) : 31;
  for (FusedStack& f : fused) {
    f.active = false;
    if (!enabled || (dtype != BF16 && dtype != F16 && dtype != F16P) || f.nblocks < 1 || f.nblocks > T17_MAX_BLOCKS) continue;
    if (f.kind == 2) {
      if (!(enabled & 4)) continue;
      const ConvLayer& c2a = convs[f.conv0];
      const ConvLayer& c2b = convs[f.conv0 + 1];
      if (c2a.cout != 32 || c2a.K != 288 || c2b.cout != 64 || c2b.K != 288 || c2a.ncls != 1 || c2b.ncls != 1)
        return fail(VNF_E_INVALID, "fused stem: unexpected layer shapes");
      StemMidPack pk;
      pk.w[0] = c2a.w; pk.kpad[0] = c2a.Kpad;
      pk.w[1] = c2b.w; pk.kpad[1] = c2b.Kpad;
      bool ext_ok = false;
      if ((enabled & 8) && f.ext_conv >= 0) {
        const ConvLayer& c3b = convs[f.ext_conv];
        ext_ok = c3b.cout == 80 && c3b.K == 64 && c3b.KH == 1 && c3b.ncls == 1 && c3b.nseg == 1 && c3b.res_buf < 0 && c3b.act == ACT_RELU &&
                 bufs[f.ext_out_buf].C == 80;
      }
      if (dtype == F16P && !ext_ok) continue;   // the planar split-f16 stem kernel (stem_mids.hip) always carries conv2d_3b
      f.wstream = dalloc(dtype == F16P ? SMS_WFRAG_BYTES : SM_WFRAG_BYTES);
      f.bias = (float*)dalloc(SM_BIAS * 4);
      if (!f.wstream || !f.bias) return VNF_E_HIP;
      VNF_HIP(hipMemcpy(f.bias, c2a.bias, 32 * 4, hipMemcpyDeviceToDevice));
      VNF_HIP(hipMemcpy(f.bias + 32, c2b.bias, 64 * 4, hipMemcpyDeviceToDevice));
      VNF_HIP(dtype == F16P ? stem_mids_repack(pk, f.wstream, 0) : stem_mid_repack(pk, f.wstream, 0));
      VNF_HIP(hipDeviceSynchronize());
      f.macs_alg = c2a.macs_alg + c2b.macs_alg;
      f.active = true;
      f.ext = ext_ok;
      if (ext_ok) f.macs_alg += convs[f.ext_conv].macs_alg;
      continue;
    }
    if (f.kind == 35) {
      if (!(enabled & 2)) continue;
      std::vector<float> bias((size_t)f.nblocks * B35_BIAS, 0.f);
      static const int rows[5] = {96, 32, 32, 32, 256}, ks[5] = {256, 288, 288, 288, 96}, boff[5] = {0, 96, 128, 160, 192};
      const size_t wimg_bytes = dtype == F16P ? B35S_WIMG_BYTES : B35_WIMG_BYTES;
      f.wstream = dalloc((size_t)f.nblocks * wimg_bytes);
      f.bias = (float*)dalloc(bias.size() * 4);
      if (!f.wstream || !f.bias) return VNF_E_HIP;
      for (int b = 0; b < f.nblocks; ++b) {
        Block35Pack pk;
        pk.bias = f.bias + (size_t)b * B35_BIAS;
        for (int c = 0; c < 5; ++c) {
          const ConvLayer& L = convs[f.conv0 + 5 * b + c];
          if (L.cout != rows[c] || L.K != ks[c] || L.ncls != 1) return fail(VNF_E_INVALID, "fused Block35: unexpected layer shapes");
          pk.w[c] = L.w;
          pk.kpad[c] = L.Kpad;
          VNF_HIP(hipMemcpy(&bias[(size_t)b * B35_BIAS + boff[c]], L.bias, (size_t)rows[c] * 4, hipMemcpyDeviceToHost));
          f.macs_alg += L.macs_alg;
        }
        VNF_HIP(hipMemcpy(f.bias + (size_t)b * B35_BIAS, &bias[(size_t)b * B35_BIAS], (size_t)B35_BIAS * 4, hipMemcpyHostToDevice));
        VNF_HIP(dtype == F16P ? block35s_repack(pk, (char*)f.wstream + (size_t)b * wimg_bytes, 0)
                              : block35_repack(pk, (char*)f.wstream + (size_t)b * wimg_bytes, 0));
      }
      VNF_HIP(hipDeviceSynchronize());
      f.active = true;
      f.stack = (enabled & 16) && dtype != F16P;
      f.ext = false;
      if (f.stack && (enabled & 32) && f.ext_conv >= 0) {
        const ConvLayer& t = convs[f.ext_conv];
        const ConvLayer& last_up = convs[f.conv0 + 5 * (f.nblocks - 1) + 4];
        const bool ok = t.KH == 1 && t.KW == 1 && t.K == 256 && t.cout == 192 && t.ncls == 1 && t.nseg == 1 && t.res_buf < 0 &&
                        t.act == ACT_RELU && !t.out_f32 && t.x_buf == last_up.seg[0].buf && t.x_coff == 0 &&
                        t.seg[0].buf == f.ext_out_buf && t.seg[0].coff == 0 && bufs[f.ext_out_buf].C == 192;
        if (ok) {
          f.wtail = dalloc(B35_TAIL_BYTES);
          if (!f.wtail) return VNF_E_HIP;
          VNF_HIP(block35_tail_repack(t.w, t.Kpad, t.bias, f.wtail, 0));
          VNF_HIP(hipDeviceSynchronize());
          f.ext = true;
          f.macs_alg += t.macs_alg;
        }
      }
      continue;
    }
    if (!(enabled & 1)) continue;
    Trunk17Pack pk;
    memset(&pk, 0, sizeof pk);
    pk.nblocks = f.nblocks;
    std::vector<float> bias((size_t)f.nblocks * T17_BIAS, 0.f);
    static const int rows[4] = {256, 128, 128, 896}, ks[4] = {896, 896, 896, 256}, boff[4] = {0, 256, 384, 512};
    bool ok = true;
    for (int b = 0; b < f.nblocks && ok; ++b)
      for (int c = 0; c < 4 && ok; ++c) {
        const ConvLayer& L = convs[f.conv0 + 4 * b + c];
        if (L.cout != rows[c] || L.K != ks[c] || L.Kpad != ks[c] || L.ncls != 1) { ok = false; break; }
        pk.w[b][c] = L.w;
        pk.kpad[c] = L.Kpad;
        VNF_HIP(hipMemcpy(&bias[(size_t)b * T17_BIAS + boff[c]], L.bias, (size_t)rows[c] * 4, hipMemcpyDeviceToHost));
        f.macs_alg += L.macs_alg;
      }
    if (!ok) return fail(VNF_E_INVALID, "fused Block17 stack: unexpected layer shapes");
    f.wstream = dalloc(dtype == F16P ? trunk17s_stream_bytes(f.nblocks) : trunk17_stream_bytes(f.nblocks));
    f.bias = (float*)upload(bias.data(), bias.size() * 4);
    if (!f.wstream || !f.bias) return VNF_E_HIP;
    VNF_HIP(dtype == F16P ? trunk17s_repack(pk, f.wstream, 0) : trunk17_repack(pk, f.wstream, 0));
    VNF_HIP(hipDeviceSynchronize());
    f.active = true;
  }
  return VNF_OK;
}

// A tap's buffer exists in memory unless every op that writes it sits inside an active fused stack and the stack's own
// kernel does not produce it (conv2d_2a / conv2d_2b / maxpool_3a with the fused stem: those tensors only ever live in LDS).
bool Encoder::buf_materialised(int buf) const {
  for (const FusedStack& f : fused) {
    if (!f.active) continue;
    const int end = f.ext ? f.ext_last : f.last;
    bool written = false;
    for (int oi = f.first; oi < end; ++oi) {
      const Op& op = ops[oi];
      if (op.kind == Op::CONV) {
        const ConvLayer& L = convs[op.a];
        for (int i = 0; i < L.nseg; ++i) written |= L.seg[i].buf == buf;
      } else if (op.kind == Op::MAXPOOL || op.kind == Op::MAXPOOLC) {
        written |= op.b == buf;
      }
    }
    if (!written) continue;
    bool produced = false;
    if (f.kind == 2) produced = buf == (f.ext ? f.ext_out_buf : f.out_buf);
    else if (f.kind == 35)
    {
      for (int b = 0; b < f.nblocks; ++b) produced |= convs[f.conv0 + 5 * b + 4].seg[0].buf == buf;
      produced |= f.ext && buf == f.ext_out_buf;
    }
    else produced = buf == f.out_buf;
    if (!produced) return false;
  }
  return true;
}

struct Piece {  // output channels contributed by one reference conv / linear
  const float* w;  // [cout][cin][KH][KW]
  int cout, cout_pad;
  std::vector<float> scale, bias;  // per logical output channel
  std::vector<float> slope;        // optional PReLU slopes
};

struct SegSpec { int c0, c1, buf, coff; };

struct ConvSpec {
  std::string name;
  int x_buf, x_coff = 0, cin, cin_pad;
  int KH = 1, KW = 1, sh = 1, sw = 1, ph = 0, pw = 0;
  std::vector<Piece> pieces;
  std::vector<SegSpec> segs;
  int res_buf = -1, res_coff = 0;
  int act = ACT_RELU, out_f32 = 0;
  // folded pre-conv BatchNorm (IR-100 bn1): x' = x*pre_s[c] + pre_t[c] on valid (unpadded) taps
  const std::vector<float>* pre_s = nullptr;
  const std::vector<float>* pre_t = nullptr;
};

static int add_conv(Encoder& e, const ConvSpec& s) {
  const int es = dtype_size(e.dtype), ch = dtype_chan_align(e.dtype), bke = 128 / es;
  const bool planar = e.dtype == F16P;
  const Buf& xb = e.bufs[s.x_buf];
  ConvLayer L;
  L.name = s.name;
  L.x_buf = s.x_buf; L.x_coff = s.x_coff; L.cin = s.cin_pad;
  L.H = xb.H; L.W = xb.W;
  L.KH = s.KH; L.KW = s.KW; L.sh = s.sh; L.sw = s.sw; L.ph = s.ph; L.pw = s.pw;
  L.Ho = (L.H + 2 * s.ph - s.KH) / s.sh + 1;
  L.Wo = (L.W + 2 * s.pw - s.KW) / s.sw + 1;
  if (s.cin_pad % ch || s.x_coff % ch || xb.C % ch) return fail(VNF_E_INVALID, s.name + ": channel alignment");
  L.K = s.KH * s.KW * s.cin_pad;
  L.Kpad = (L.K + bke - 1) / bke * bke;
  int cout = 0, cout_logical = 0;
  for (auto& p : s.pieces) { cout += p.cout_pad; cout_logical += p.cout; }
  L.cout = cout;
  L.cout_pad = (cout + 127) / 128 * 128;
  L.ncls = s.pre_s ? 9 : 1;
  if (cout % 8) return fail(VNF_E_INVALID, s.name + ": cout % 8");

  std::vector<float> wpk((size_t)L.cout_pad * L.Kpad, 0.f);
  std::vector<float> bias((size_t)L.ncls * L.cout_pad, 0.f), slope((size_t)L.cout_pad, 0.f);
  bool has_slope = false;
  int co0 = 0;
  for (auto& p : s.pieces) {
    if (!p.w) return fail(VNF_E_MISSING, s.name + ": weight missing");
    for (int co = 0; co < p.cout; ++co) {
      const float sc = p.scale.empty() ? 1.f : p.scale[co];
      float* dst = &wpk[(size_t)(co0 + co) * L.Kpad];
      for (int c = 0; c < s.cin; ++c) {
        const float ps = s.pre_s ? (*s.pre_s)[c] : 1.f;
        for (int kh = 0; kh < s.KH; ++kh)
          for (int kw = 0; kw < s.KW; ++kw) {
            const float wv = p.w[(((size_t)co * s.cin + c) * s.KH + kh) * s.KW + kw];
            dst[(kh * s.KW + kw) * s.cin_pad + c] = wv * sc * ps;
          }
      }
      const float b = p.bias.empty() ? 0.f : p.bias[co];
      if (!s.pre_s) {
        bias[co0 + co] = b;
      } else {
        // border classes: the BN shift only reaches the output through taps that land inside
        // the image; class (r,c) in {first, interior, last}^2 selects the valid tap set.
        for (int rc = 0; rc < 3; ++rc)
          for (int cc = 0; cc < 3; ++cc) {
            double acc = 0;
            for (int kh = 0; kh < s.KH; ++kh) {
              if ((rc == 0 && kh < s.ph) || (rc == 2 && kh >= s.KH - s.ph)) continue;
              for (int kw = 0; kw < s.KW; ++kw) {
                if ((cc == 0 && kw < s.pw) || (cc == 2 && kw >= s.KW - s.pw)) continue;
                for (int c = 0; c < s.cin; ++c)
                  acc += (double)p.w[(((size_t)co * s.cin + c) * s.KH + kh) * s.KW + kw] * (*s.pre_t)[c];
              }
            }
            bias[(size_t)(rc * 3 + cc) * L.cout_pad + co0 + co] = b + (float)(acc * sc);
          }
      }
      if (!p.slope.empty()) { slope[co0 + co] = p.slope[co]; has_slope = true; }
    }
    co0 += p.cout_pad;
  }
  std::vector<char> wdev((size_t)L.cout_pad * L.Kpad * es);
  convert_to(e.dtype, wpk.data(), wdev.data(), wpk.size());
  L.w = e.upload(wdev.data(), wdev.size());
  L.bias = (float*)e.upload(bias.data(), bias.size() * 4);
  if (has_slope) L.slope = (float*)e.upload(slope.data(), slope.size() * 4);
  // gather table: one entry per 16-byte chunk of the K-tile image, 8 per K tile.  Planar split-f16: chunk q of a tile
  // is the hi (q < 4) or lo (q >= 4) plane of the 8-channel unit q & 3, 16 bytes (4 elements) into the unit for lo.
  std::vector<int4> kt(L.Kpad / bke * 8);
  for (int kc = 0; kc < (int)kt.size(); ++kc) {
    const int q = kc & 7;
    const int k = planar ? (kc >> 3) * bke + (q & 3) * 8 : kc * (16 / es);
    if (k < L.K) {
      const int tap = k / s.cin_pad, c = k % s.cin_pad, kh = tap / s.KW, kw = tap % s.KW;
      kt[kc] = int4{(kh * L.W + kw) * xb.C + c + (planar ? (q >> 2) * 4 : 0), kh, kw, 1};
    } else {
      kt[kc] = int4{0, 0, 0, 0};
    }
  }
  L.ktab = (int4*)e.upload(kt.data(), kt.size() * sizeof(int4));
  if (!L.w || !L.bias || !L.ktab) return VNF_E_HIP;

  L.nseg = (int)s.segs.size();
  if (L.nseg < 1 || L.nseg > 4) return fail(VNF_E_INVALID, s.name + ": segments");
  for (int i = 0; i < L.nseg; ++i) {
    L.seg[i].c0 = s.segs[i].c0; L.seg[i].c1 = s.segs[i].c1;
    L.seg[i].buf = s.segs[i].buf; L.seg[i].coff = s.segs[i].coff;
    if (s.segs[i].buf >= 0) {
      const Buf& ob = e.bufs[s.segs[i].buf];
      if (ob.H != L.Ho || ob.W != L.Wo || s.segs[i].coff + (s.segs[i].c1 - s.segs[i].c0) > ob.C)
        return fail(VNF_E_INVALID, s.name + ": output buffer shape");
    }
  }
  L.res_buf = s.res_buf; L.res_coff = s.res_coff;
  L.act = s.act; L.out_f32 = s.out_f32;
  L.macs_alg = (double)L.Ho * L.Wo * cout_logical * (double)(s.KH * s.KW * s.cin);
  const int kstep = planar ? bke : bke / 2;   // k values one MFMA group consumes
  const int k32 = (L.K + kstep - 1) / kstep * kstep;
  L.macs_exec = (double)L.Ho * L.Wo * cout * (double)k32;
  e.convs.push_back(L);
  Op op; op.kind = Op::CONV; op.a = (int)e.convs.size() - 1;
  e.ops.push_back(op);
  return VNF_OK;
}

static bool bn_fold(WeightMap& wm, const std::string& p, int C, float eps, std::vector<float>& s, std::vector<float>& t) {
  const float* g = wm.get(p + ".weight", C);
  const float* b = wm.get(p + ".bias", C);
  const float* m = wm.get(p + ".running_mean", C);
  const float* v = wm.get(p + ".running_var", C);
  if (!g || !b || !m || !v) return false;
  s.resize(C); t.resize(C);
  for (int i = 0; i < C; ++i) {
    const double sc = (double)g[i] / std::sqrt((double)v[i] + (double)eps);
    s[i] = (float)sc;
    t[i] = (float)((double)b[i] - (double)m[i] * sc);
  }
  return true;
}

// BasicConv2d (inception_resnet_v1.py:12-33): conv(no bias) -> BN(eps 1e-3) -> ReLU
static bool basic_piece(WeightMap& wm, const std::string& p, int cin, int cout, int kh, int kw, Piece& out, int cout_pad = 0) {
  out.w = wm.get(p + ".conv.weight", (int64_t)cout * cin * kh * kw);
  out.cout = cout;
  out.cout_pad = cout_pad ? cout_pad : cout;
  return out.w && bn_fold(wm, p + ".bn", cout, 1e-3f, out.scale, out.bias);
}

static void add_maxpool(Encoder& e, int in_buf, int out_buf, int out_coff) {
  Op op; op.kind = Op::MAXPOOL; op.a = in_buf; op.b = out_buf; op.c = out_coff;
  e.ops.push_back(op);
}

#define TRY(x) do { int _r = (x); if (_r != VNF_OK) return _r; } while (0)
#define NEED(x) do { if (!(x)) return fail(VNF_E_MISSING, "missing weight: " + wm.missing); } while (0)

int build_irv1(Encoder& e, WeightMap& wm) {
  e.in_size = 160;
  const int ch = 16 / dtype_size(e.dtype);
  (void)ch;
  const int b_in = e.add_buf(160, 160, 8);
  const int b_1a = e.add_buf(79, 79, 32), b_2a = e.add_buf(77, 77, 32), b_2b = e.add_buf(77, 77, 64);
  const int b_3a = e.add_buf(38, 38, 64), b_3b = e.add_buf(38, 38, 80), b_4a = e.add_buf(36, 36, 192);
  const int x35[3] = {e.add_buf(17, 17, 256), e.add_buf(17, 17, 256), e.add_buf(17, 17, 256)};
  const int t35a = e.add_buf(17, 17, 64), t35b = e.add_buf(17, 17, 32), cat35 = e.add_buf(17, 17, 96);
  const int m6a = e.add_buf(17, 17, 192), m6b = e.add_buf(17, 17, 192);
  const int x17[3] = {e.add_buf(8, 8, 896), e.add_buf(8, 8, 896), e.add_buf(8, 8, 896)};
  const int t17a = e.add_buf(8, 8, 128), t17b = e.add_buf(8, 8, 128), cat17 = e.add_buf(8, 8, 256);
  const int m7a = e.add_buf(8, 8, 768), m7b = e.add_buf(8, 8, 256);
  const int x8[3] = {e.add_buf(3, 3, 1792), e.add_buf(3, 3, 1792), e.add_buf(3, 3, 1792)};
  const int t8a = e.add_buf(3, 3, 192), t8b = e.add_buf(3, 3, 192), cat8 = e.add_buf(3, 3, 384);
  const int pool = e.add_buf(1, 1, 1792);

  { Op op; op.kind = Op::PACK; op.a = b_in; e.ops.push_back(op); }

  auto simple = [&](const std::string& name, int xb, int xoff, int cin, int cin_pad, int cout, int kh, int kw, int st,
                    int ph, int pw, int ob, int ooff, int cout_pad = 0) -> int {
    ConvSpec s;
    s.name = name; s.x_buf = xb; s.x_coff = xoff; s.cin = cin; s.cin_pad = cin_pad;
    s.KH = kh; s.KW = kw; s.sh = s.sw = st; s.ph = ph; s.pw = pw;
    s.pieces.resize(1);
    if (!basic_piece(wm, name, cin, cout, kh, kw, s.pieces[0], cout_pad))
      return fail(VNF_E_MISSING, "missing weight: " + wm.missing);
    const int cp = cout_pad ? cout_pad : cout;
    s.segs.push_back({0, cp, ob, ooff});
    return add_conv(e, s);
  };
  // fused 1x1 reducers of several branches reading the same input: one GEMM, columns routed
  auto fused1x1 = [&](const std::string& name, std::vector<std::string> prefixes, int xb, int cin, int cout_each,
                      std::vector<SegSpec> segs) -> int {
    ConvSpec s;
    s.name = name; s.x_buf = xb; s.cin = s.cin_pad = cin;
    s.pieces.resize(prefixes.size());
    for (size_t i = 0; i < prefixes.size(); ++i)
      if (!basic_piece(wm, prefixes[i], cin, cout_each, 1, 1, s.pieces[i]))
        return fail(VNF_E_MISSING, "missing weight: " + wm.missing);
    s.segs = segs;
    return add_conv(e, s);
  };
  // block-output 1x1 conv with bias, scaled residual and optional ReLU
  // (inception_resnet_v1.py:63-67): relu(conv(cat)*scale + x) == relu(conv_{w*scale} + b*scale + x)
  auto up = [&](const std::string& p, int cat, int cin, int cout, float scale, int xin, int xout, bool relu) -> int {
    ConvSpec s;
    s.name = p + ".conv2d"; s.x_buf = cat; s.cin = s.cin_pad = cin;
    s.pieces.resize(1);
    Piece& pc = s.pieces[0];
    pc.w = wm.get(p + ".conv2d.weight", (int64_t)cout * cin);
    const float* b = wm.get(p + ".conv2d.bias", cout);
    if (!pc.w || !b) return fail(VNF_E_MISSING, "missing weight: " + wm.missing);
    pc.cout = pc.cout_pad = cout;
    pc.scale.assign(cout, scale);
    pc.bias.resize(cout);
    for (int i = 0; i < cout; ++i) pc.bias[i] = b[i] * scale;
    s.segs.push_back({0, cout, xout, 0});
    s.res_buf = xin;
    s.act = relu ? ACT_RELU : ACT_NONE;
    return add_conv(e, s);
  };

  // ---- stem (inception_resnet_v1.py:281-287)
  TRY(simple("conv2d_1a", b_in, 0, 3, 8, 32, 3, 3, 2, 0, 0, b_1a, 0));
  {
    // the first convolution runs as a direct kernel on the caller's NCHW tensor (aux_kernels.hip): the layer stays
    // in `convs` for the FLOP accounting, the PACK + CONV pair of ops becomes one STEM1 op
    static const int direct = getenv("VNF_DIRECT_STEM") ? atoi(getenv("VNF_DIRECT_STEM")) : 1;
    Piece pc;
    if (direct && basic_piece(wm, "conv2d_1a", 3, 32, 3, 3, pc)) {
      std::vector<float> wt(27 * 32 + 32);
      for (int co = 0; co < 32; ++co) {
        for (int k = 0; k < 27; ++k) wt[k * 32 + co] = pc.w[co * 27 + k] * pc.scale[co];
        wt[27 * 32 + co] = pc.bias[co];
      }
      e.stem_wt = (float*)e.upload(wt.data(), wt.size() * 4);
      if (!e.stem_wt) return VNF_E_HIP;
      e.ops.resize(e.ops.size() - 2);
      Op op; op.kind = Op::STEM1; op.a = (int)e.convs.size() - 1; op.b = b_1a;
      e.ops.push_back(op);
    }
  }
  {
    FusedStack f;   // 16-bit compute dtypes: conv2d_2a + conv2d_2b + maxpool_3a as one rolling-row launch (stem_mid.hip)
    f.kind = 2;
    f.first = (int)e.ops.size(); f.conv0 = (int)e.convs.size();
    f.in_buf = b_1a; f.out_buf = b_3a; f.nblocks = 1;
    TRY(simple("conv2d_2a", b_1a, 0, 32, 32, 32, 3, 3, 1, 0, 0, b_2a, 0));
    TRY(simple("conv2d_2b", b_2a, 0, 32, 32, 64, 3, 3, 1, 1, 1, b_2b, 0));
    add_maxpool(e, b_2b, b_3a, 0);
    f.last = (int)e.ops.size();
    TRY(simple("conv2d_3b", b_3a, 0, 64, 64, 80, 1, 1, 1, 0, 0, b_3b, 0));
    f.ext_last = (int)e.ops.size(); f.ext_conv = (int)e.convs.size() - 1; f.ext_out_buf = b_3b;
    e.fused.push_back(f);
  }
  TRY(simple("conv2d_4a", b_3b, 0, 80, 80, 192, 3, 3, 1, 0, 0, b_4a, 0));
  TRY(simple("conv2d_4b", b_4a, 0, 192, 192, 256, 3, 3, 2, 0, 0, x35[0], 0));
  const int stem_end = (int)e.ops.size();
  e.taps["conv2d_1a"] = {b_1a, 0, 32}; e.taps["conv2d_2a"] = {b_2a, 0, 32}; e.taps["conv2d_2b"] = {b_2b, 0, 64};
  e.taps["maxpool_3a"] = {b_3a, 0, 64}; e.taps["conv2d_3b"] = {b_3b, 0, 80}; e.taps["conv2d_4a"] = {b_4a, 0, 192};
  e.taps["conv2d_4b"] = {x35[0], 0, 256};

  // ---- repeat_1: 5 x Block35 (36-67)
  int cur = 0;
  const int r1_first_op = (int)e.ops.size(), r1_first_conv = (int)e.convs.size();
  for (int i = 0; i < 5; ++i) {
    const std::string p = "repeat_1." + std::to_string(i);
    const int X = x35[cur], Y = x35[cur == 1 ? 2 : 1];
    TRY(fused1x1(p + ".reduce", {p + ".branch0", p + ".branch1.0", p + ".branch2.0"}, X, 256, 32,
                 {{0, 32, cat35, 0}, {32, 96, t35a, 0}}));
    TRY(simple(p + ".branch1.1", t35a, 0, 32, 32, 32, 3, 3, 1, 1, 1, cat35, 32));
    TRY(simple(p + ".branch2.1", t35a, 32, 32, 32, 32, 3, 3, 1, 1, 1, t35b, 0));
    TRY(simple(p + ".branch2.2", t35b, 0, 32, 32, 32, 3, 3, 1, 1, 1, cat35, 64));
    TRY(up(p, cat35, 96, 256, 0.17f, X, Y, true));
    cur = (cur == 1 ? 2 : 1);
  }
  e.taps["repeat_1"] = {x35[cur], 0, 256};
  {
    FusedStack f;   // 16-bit compute dtypes: one fused launch per block (block35.hip) or for the whole stack (trunk35.hip)
    f.kind = 35;
    f.first = r1_first_op; f.last = (int)e.ops.size();
    f.nblocks = 5; f.conv0 = r1_first_conv;
    // mixed_6a.branch1.0 (137) reads the stack's output only: listed right behind it so the stack kernel can take it over
    TRY(simple("mixed_6a.branch1.0", x35[cur], 0, 256, 256, 192, 1, 1, 1, 0, 0, m6a, 0));
    f.ext_last = (int)e.ops.size(); f.ext_conv = (int)e.convs.size() - 1; f.ext_out_buf = m6a;
    e.fused.push_back(f);
  }
  // ---- mixed_6a (129-149)
  {
    const int X = x35[cur], O = x17[0];
    TRY(simple("mixed_6a.branch0", X, 0, 256, 256, 384, 3, 3, 2, 0, 0, O, 0));
    TRY(simple("mixed_6a.branch1.1", m6a, 0, 192, 192, 192, 3, 3, 1, 1, 1, m6b, 0));
    TRY(simple("mixed_6a.branch1.2", m6b, 0, 192, 192, 256, 3, 3, 2, 0, 0, O, 384));
    add_maxpool(e, X, O, 640);
  }
  e.taps["mixed_6a"] = {x17[0], 0, 896};
  // ---- repeat_2: 10 x Block17 (70-95)
  cur = 0;
  const int r2_first_op = (int)e.ops.size(), r2_first_conv = (int)e.convs.size();
  for (int i = 0; i < 10; ++i) {
    const std::string p = "repeat_2." + std::to_string(i);
    const int X = x17[cur], Y = x17[cur == 1 ? 2 : 1];
    TRY(fused1x1(p + ".reduce", {p + ".branch0", p + ".branch1.0"}, X, 896, 128,
                 {{0, 128, cat17, 0}, {128, 256, t17a, 0}}));
    TRY(simple(p + ".branch1.1", t17a, 0, 128, 128, 128, 1, 7, 1, 0, 3, t17b, 0));
    TRY(simple(p + ".branch1.2", t17b, 0, 128, 128, 128, 7, 1, 1, 3, 0, cat17, 128));
    TRY(up(p, cat17, 256, 896, 0.10f, X, Y, true));
    cur = (cur == 1 ? 2 : 1);
  }
  e.taps["repeat_2"] = {x17[cur], 0, 896};
  {
    // 16-bit compute dtypes run the whole stack as one persistent kernel (trunk17.hip); the plan ops above stay as
    // the fp32 / split-f16 path, the FLOP accounting and the source of the packed weights
    FusedStack f;
    f.first = r2_first_op; f.last = (int)e.ops.size();
    f.in_buf = x17[0]; f.out_buf = x17[cur];
    f.nblocks = 10; f.conv0 = r2_first_conv;
    e.fused.push_back(f);
  }
  // ---- mixed_7a (152-181)
  {
    const int X = x17[cur], O = x8[0];
    TRY(fused1x1("mixed_7a.reduce", {"mixed_7a.branch0.0", "mixed_7a.branch1.0", "mixed_7a.branch2.0"}, X, 896, 256,
                 {{0, 768, m7a, 0}}));
    TRY(simple("mixed_7a.branch0.1", m7a, 0, 256, 256, 384, 3, 3, 2, 0, 0, O, 0));
    TRY(simple("mixed_7a.branch1.1", m7a, 256, 256, 256, 256, 3, 3, 2, 0, 0, O, 384));
    TRY(simple("mixed_7a.branch2.1", m7a, 512, 256, 256, 256, 3, 3, 1, 1, 1, m7b, 0));
    TRY(simple("mixed_7a.branch2.2", m7b, 0, 256, 256, 256, 3, 3, 2, 0, 0, O, 640));
    add_maxpool(e, X, O, 896);
  }
  e.taps["mixed_7a"] = {x8[0], 0, 1792};
  // ---- repeat_3 (5 x Block8, scale 0.2) + block8 (scale 1, no ReLU) (98-126, 247-254)
  cur = 0;
  for (int i = 0; i < 6; ++i) {
    const std::string p = i < 5 ? "repeat_3." + std::to_string(i) : std::string("block8");
    const int X = x8[cur], Y = x8[cur == 1 ? 2 : 1];
    TRY(fused1x1(p + ".reduce", {p + ".branch0", p + ".branch1.0"}, X, 1792, 192,
                 {{0, 192, cat8, 0}, {192, 384, t8a, 0}}));
    TRY(simple(p + ".branch1.1", t8a, 0, 192, 192, 192, 1, 3, 1, 0, 1, t8b, 0));
    TRY(simple(p + ".branch1.2", t8b, 0, 192, 192, 192, 3, 1, 1, 1, 0, cat8, 192));
    TRY(up(p, cat8, 384, 1792, i < 5 ? 0.20f : 1.0f, X, Y, i < 5));
    cur = (cur == 1 ? 2 : 1);
    if (i == 4) e.taps["repeat_3"] = {x8[cur], 0, 1792};
  }
  e.taps["block8"] = {x8[cur], 0, 1792};
  // ---- tail (294-302): avgpool -> last_linear (no bias) -> last_bn (eps 1e-3) -> L2 normalise
  { Op op; op.kind = Op::AVGPOOL; op.a = x8[cur]; op.b = pool; e.ops.push_back(op); }
  {
    ConvSpec s;
    s.name = "last_linear"; s.x_buf = pool; s.cin = s.cin_pad = 1792;
    s.pieces.resize(1);
    Piece& pc = s.pieces[0];
    pc.w = wm.get("last_linear.weight", 512 * 1792);
    pc.cout = pc.cout_pad = 512;
    NEED(pc.w && bn_fold(wm, "last_bn", 512, 1e-3f, pc.scale, pc.bias));
    s.segs.push_back({0, 512, -2, 0});
    s.act = ACT_NONE; s.out_f32 = 1;
    TRY(add_conv(e, s));
  }
  { Op op; op.kind = Op::L2NORM; e.ops.push_back(op); }

  // unfused, the stem runs in sub-batches of 128 images so its big producer -> consumer tensors stay inside the
  // Infinity Cache; with conv2d_2a/2b/maxpool fused (one workgroup per image, no big intermediate) a sub-batch would
  // only leave half the CUs without a workgroup
  const int fuse_mask = getenv("VNF_FUSE") ? atoi(getenv("VNF_FUSE")) : 31;
  int chunk = ((fuse_mask & 4) && (e.dtype == BF16 || e.dtype == F16 || (e.dtype == F16P && (fuse_mask & 8)))) ? 256 : 128;
  if (const char* c = getenv("VNF_STEM_CHUNK")) chunk = atoi(c) > 0 ? atoi(c) : chunk;
  e.groups.push_back({0, stem_end, chunk});
  e.groups.push_back({stem_end, (int)e.ops.size(), 1 << 30});
  return VNF_OK;
}

// A linear layer as a 1x1 convolution over a 1x1 "image" (used by the MLP classifier).
int add_linear(Encoder& e, const std::string& name, const float* w, const float* b, int cin, int cout, int cout_pad,
               int x_buf, int o_buf, int act) {
  ConvSpec s;
  s.name = name; s.x_buf = x_buf; s.cin = s.cin_pad = cin;
  s.pieces.resize(1);
  Piece& pc = s.pieces[0];
  pc.w = w; pc.cout = cout; pc.cout_pad = cout_pad;
  pc.bias.assign(b, b + cout);
  s.segs.push_back({0, cout_pad, o_buf, 0});
  s.act = act;
  return add_conv(e, s);
}

// IResNet-100 (models/iresnet_encoder.py:26-61, 64-159).  Per IBasicBlock two launches:
//   A: conv1(bn1(x)) -> bn2 -> PReLU.  bn1 sits BEFORE a zero-padded conv, so it cannot be folded
//      into a plain bias: its scale goes into the weights, and its shift becomes a bias that depends
//      on which taps fall inside the image -- one of 9 border classes, picked in the epilogue.
//      bn2 folds into per-output scale / bias, PReLU runs in the epilogue.
//   B: conv2 (stride) -> bn3, + identity (x, or the 1x1-stride-2 downsample branch with its BN).
// The head (bn2 -> flatten (C,H,W) -> fc -> features BN1d) is ONE 7x7 "convolution" over the
// NHWC map: fc.weight viewed as (512, 512, 7, 7) is exactly that conv's weight.
int build_ir100(Encoder& e, WeightMap& wm) {
  e.in_size = 112;
  const float EPS = 2e-5f;
  const int b_in = e.add_buf(112, 112, 8);
  { Op op; op.kind = Op::PACK; op.a = b_in; e.ops.push_back(op); }
  const int planes[4] = {64, 128, 256, 512}, nblk[4] = {3, 13, 30, 3};
  int H = 112;
  int x = e.add_buf(112, 112, 64);
  {  // stem: conv1 3x3 p1 (3->64) -> bn1 -> PReLU (iresnet_encoder.py:140-142)
    ConvSpec s;
    s.name = "conv1"; s.x_buf = b_in; s.cin = 3; s.cin_pad = 8; s.KH = s.KW = 3; s.ph = s.pw = 1;
    s.pieces.resize(1);
    Piece& pc = s.pieces[0];
    pc.w = wm.get("conv1.weight", 64 * 27);
    pc.cout = pc.cout_pad = 64;
    const float* sl = wm.get("prelu.weight", 64);
    NEED(pc.w && sl && bn_fold(wm, "bn1", 64, EPS, pc.scale, pc.bias));
    pc.slope.assign(sl, sl + 64);
    s.segs.push_back({0, 64, x, 0});
    s.act = ACT_PRELU;
    TRY(add_conv(e, s));
  }
  e.taps["stem"] = {x, 0, 64};
  int cin = 64;
  std::vector<int> stage_end;
  for (int li = 0; li < 4; ++li) {
    const int P = planes[li], Ho = H / 2;
    const int t_first = e.add_buf(H, H, P);      // conv1 output of the first block (input resolution)
    const int t_rest = e.add_buf(Ho, Ho, P);
    const int dsb = e.add_buf(Ho, Ho, P);        // downsample branch
    const int y[2] = {e.add_buf(Ho, Ho, P), e.add_buf(Ho, Ho, P)};
    int cur = -1;
    for (int b = 0; b < nblk[li]; ++b) {
      const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
      const int xin = b == 0 ? x : y[cur];
      const int xout = b == 0 ? y[0] : y[cur ^ 1];
      const int ci = b == 0 ? cin : P, t1 = b == 0 ? t_first : t_rest, st = b == 0 ? 2 : 1;
      std::vector<float> s1, t1v;
      NEED(bn_fold(wm, p + ".bn1", ci, EPS, s1, t1v));
      {
        ConvSpec s;
        s.name = p + ".conv1"; s.x_buf = xin; s.cin = s.cin_pad = ci; s.KH = s.KW = 3; s.ph = s.pw = 1;
        s.pieces.resize(1);
        Piece& pc = s.pieces[0];
        pc.w = wm.get(p + ".conv1.weight", (int64_t)P * ci * 9);
        pc.cout = pc.cout_pad = P;
        const float* sl = wm.get(p + ".prelu.weight", P);
        NEED(pc.w && sl && bn_fold(wm, p + ".bn2", P, EPS, pc.scale, pc.bias));
        pc.slope.assign(sl, sl + P);
        s.pre_s = &s1; s.pre_t = &t1v;
        s.segs.push_back({0, P, t1, 0});
        s.act = ACT_PRELU;
        TRY(add_conv(e, s));
      }
      if (b == 0) {
        ConvSpec s;
        s.name = p + ".downsample"; s.x_buf = xin; s.cin = s.cin_pad = ci; s.sh = s.sw = 2;
        s.pieces.resize(1);
        Piece& pc = s.pieces[0];
        pc.w = wm.get(p + ".downsample.0.weight", (int64_t)P * ci);
        pc.cout = pc.cout_pad = P;
        NEED(pc.w && bn_fold(wm, p + ".downsample.1", P, EPS, pc.scale, pc.bias));
        s.segs.push_back({0, P, dsb, 0});
        s.act = ACT_NONE;
        TRY(add_conv(e, s));
      }
      {
        ConvSpec s;
        s.name = p + ".conv2"; s.x_buf = t1; s.cin = s.cin_pad = P; s.KH = s.KW = 3; s.ph = s.pw = 1; s.sh = s.sw = st;
        s.pieces.resize(1);
        Piece& pc = s.pieces[0];
        pc.w = wm.get(p + ".conv2.weight", (int64_t)P * P * 9);
        pc.cout = pc.cout_pad = P;
        NEED(pc.w && bn_fold(wm, p + ".bn3", P, EPS, pc.scale, pc.bias));
        s.segs.push_back({0, P, xout, 0});
        s.res_buf = b == 0 ? dsb : xin;
        s.act = ACT_NONE;
        TRY(add_conv(e, s));
      }
      cur = b == 0 ? 0 : cur ^ 1;
    }
    x = y[cur];
    e.taps["layer" + std::to_string(li + 1)] = {x, 0, P};
    cin = P;
    H = Ho;
    stage_end.push_back((int)e.ops.size());
  }
  {  // bn2 -> flatten -> fc(+bias) -> features (iresnet_encoder.py:149-153)
    std::vector<float> s2, t2, sf, tf;
    NEED(bn_fold(wm, "bn2", 512, EPS, s2, t2) && bn_fold(wm, "features", 512, EPS, sf, tf));
    ConvSpec s;
    s.name = "fc"; s.x_buf = x; s.cin = s.cin_pad = 512; s.KH = s.KW = 7;
    s.pieces.resize(1);
    Piece& pc = s.pieces[0];
    pc.w = wm.get("fc.weight", (int64_t)512 * 25088);
    const float* fb = wm.get("fc.bias", 512);
    NEED(pc.w && fb);
    pc.cout = pc.cout_pad = 512;
    pc.scale = sf;
    pc.bias.resize(512);
    for (int i = 0; i < 512; ++i) pc.bias[i] = fb[i] * sf[i] + tf[i];
    s.pre_s = &s2; s.pre_t = &t2;
    s.segs.push_back({0, 512, -2, 0});
    s.act = ACT_NONE; s.out_f32 = 1;
    TRY(add_conv(e, s));
  }
  { Op op; op.kind = Op::COPYOUT; e.ops.push_back(op); }
  int c1 = 32, c2 = 64;
  if (const char* c = getenv("VNF_IR100_CHUNK1")) c1 = atoi(c) > 0 ? atoi(c) : c1;
  if (const char* c = getenv("VNF_IR100_CHUNK2")) c2 = atoi(c) > 0 ? atoi(c) : c2;
  e.groups.push_back({0, stage_end[0], c1});
  e.groups.push_back({stage_end[0], stage_end[1], c2});
  e.groups.push_back({stage_end[1], (int)e.ops.size(), 1 << 30});
  return VNF_OK;
}

// ---------------------------------------------------------------------------------------------
// MTCNN R-Net (mtcnn.py:52-99) and O-Net (102-157) as plans on the exact-f32 MFMA convolution core.
// Candidates are the batch dimension; the crop kernel writes NHWC4 fp32 crops into buffer 0.
// dense4 / dense5 consume x.permute(0,3,2,1) flattened (feature (w*H + h)*C + c), i.e. they are a
// 3x3 "convolution" over the 3x3xC map with weight[o][c][kh=h][kw=w] = dense[o][(w*3 + h)*C + c].
static int mtcnn_conv(Encoder& e, WeightMap& wm, const std::string& name, const std::string& prelu, int xb, int cin,
                      int cin_pad, int cout, int cout_pad, int k, int ob) {
  ConvSpec s;
  s.name = name; s.x_buf = xb; s.cin = cin; s.cin_pad = cin_pad; s.KH = s.KW = k;
  s.pieces.resize(1);
  Piece& pc = s.pieces[0];
  pc.w = wm.get(name + ".weight", (int64_t)cout * cin * k * k);
  const float* b = wm.get(name + ".bias", cout);
  const float* a = wm.get(prelu + ".weight", cout);
  if (!pc.w || !b || !a) return fail(VNF_E_MISSING, "mtcnn: missing weight " + wm.missing);
  pc.cout = cout; pc.cout_pad = cout_pad;
  pc.bias.assign(b, b + cout);
  pc.slope.assign(a, a + cout);
  s.segs.push_back({0, cout_pad, ob, 0});
  s.act = ACT_PRELU;
  return add_conv(e, s);
}

static int mtcnn_dense(Encoder& e, WeightMap& wm, const std::string& name, const std::string& prelu, int xb, int C,
                       int nout, int ob, std::vector<float>& keep) {
  const float* d = wm.get(name + ".weight", (int64_t)nout * C * 9);
  const float* b = wm.get(name + ".bias", nout);
  const float* a = wm.get(prelu + ".weight", nout);
  if (!d || !b || !a) return fail(VNF_E_MISSING, "mtcnn: missing weight " + wm.missing);
  keep.assign((size_t)nout * C * 9, 0.f);
  for (int o = 0; o < nout; ++o)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < 3; ++h)
        for (int w = 0; w < 3; ++w) keep[(((size_t)o * C + c) * 3 + h) * 3 + w] = d[(size_t)o * C * 9 + (w * 3 + h) * C + c];
  ConvSpec s;
  s.name = name; s.x_buf = xb; s.cin = s.cin_pad = C; s.KH = s.KW = 3;
  s.pieces.resize(1);
  Piece& pc = s.pieces[0];
  pc.w = keep.data(); pc.cout = pc.cout_pad = nout;
  pc.bias.assign(b, b + nout);
  pc.slope.assign(a, a + nout);
  s.segs.push_back({0, nout, ob, 0});
  s.act = ACT_PRELU;
  return add_conv(e, s);
}

static int mtcnn_heads(Encoder& e, WeightMap& wm, const std::vector<std::pair<std::string, int>>& heads, int xb, int nin,
                       int ob, int total_pad) {
  ConvSpec s;
  s.name = "heads"; s.x_buf = xb; s.cin = s.cin_pad = nin;
  s.pieces.resize(heads.size());
  int tot = 0;
  for (size_t i = 0; i < heads.size(); ++i) {
    Piece& pc = s.pieces[i];
    pc.w = wm.get(heads[i].first + ".weight", (int64_t)heads[i].second * nin);
    const float* b = wm.get(heads[i].first + ".bias", heads[i].second);
    if (!pc.w || !b) return fail(VNF_E_MISSING, "mtcnn: missing weight " + wm.missing);
    pc.cout = heads[i].second;
    pc.cout_pad = (i + 1 == heads.size()) ? total_pad - tot : heads[i].second;
    pc.bias.assign(b, b + heads[i].second);
    tot += pc.cout_pad;
  }
  s.segs.push_back({0, total_pad, ob, 0});
  s.act = ACT_NONE;
  return add_conv(e, s);
}

static void add_pool_ceil(Encoder& e, int ib, int ob, int k) {
  Op op; op.kind = Op::MAXPOOLC; op.a = ib; op.b = ob; op.c = k;
  e.ops.push_back(op);
}

// front == true: conv1 + PReLU + pool1 are computed by the detector's own fused kernel (mtcnn.hip net_front_kernel),
// which reads buffer 0 (the crops) and writes buffer 1 (the pooled map); the plan starts at conv2.
// front: conv1 + pool1 come from net_front_kernel (the plan starts at conv2); mid: conv2 + pool2 as well, from
// net_mid_kernel (mtcnn.hip: the plan starts at conv3 and reads buffer 3)
int build_rnet(Encoder& e, WeightMap& wm, bool front, bool mid) {
  e.in_size = 24;
  const int in = e.add_buf(24, 24, 4), p1 = e.add_buf(11, 11, 32);
  const int c2 = e.add_buf(9, 9, 48), p2 = e.add_buf(4, 4, 48), c3 = e.add_buf(3, 3, 64), d4 = e.add_buf(1, 1, 128);
  const int c1 = front ? -1 : e.add_buf(22, 22, 32);
  const int hd = e.add_buf(1, 1, 8);   // the heads stay the LAST buffer (mtcnn.hip reads bufs.back())
  static thread_local std::vector<float> keep;
  if (!front) {
    TRY(mtcnn_conv(e, wm, "conv1", "prelu1", in, 3, 4, 28, 32, 3, c1));
    add_pool_ceil(e, c1, p1, 3);
  }
  if (!mid) {
    TRY(mtcnn_conv(e, wm, "conv2", "prelu2", p1, 28, 32, 48, 48, 3, c2));
    add_pool_ceil(e, c2, p2, 3);
  }
  TRY(mtcnn_conv(e, wm, "conv3", "prelu3", p2, 48, 48, 64, 64, 2, c3));
  TRY(mtcnn_dense(e, wm, "dense4", "prelu4", c3, 64, 128, d4, keep));
  TRY(mtcnn_heads(e, wm, {{"dense5_1", 2}, {"dense5_2", 4}}, d4, 128, hd, 8));
  return VNF_OK;
}

int build_onet(Encoder& e, WeightMap& wm, bool front, bool mid) {
  e.in_size = 48;
  const int in = e.add_buf(48, 48, 4), p1 = e.add_buf(23, 23, 32);
  const int c2 = e.add_buf(21, 21, 64), p2 = e.add_buf(10, 10, 64), c3 = e.add_buf(8, 8, 64), p3 = e.add_buf(4, 4, 64);
  const int c4 = e.add_buf(3, 3, 128), d5 = e.add_buf(1, 1, 256);
  const int c1 = front ? -1 : e.add_buf(46, 46, 32);
  const int hd = e.add_buf(1, 1, 16);
  static thread_local std::vector<float> keep;
  if (!front) {
    TRY(mtcnn_conv(e, wm, "conv1", "prelu1", in, 3, 4, 32, 32, 3, c1));
    add_pool_ceil(e, c1, p1, 3);
  }
  if (!mid) {
    TRY(mtcnn_conv(e, wm, "conv2", "prelu2", p1, 32, 32, 64, 64, 3, c2));
    add_pool_ceil(e, c2, p2, 3);
  }
  TRY(mtcnn_conv(e, wm, "conv3", "prelu3", p2, 64, 64, 64, 64, 3, c3));
  add_pool_ceil(e, c3, p3, 2);
  TRY(mtcnn_conv(e, wm, "conv4", "prelu4", p3, 64, 64, 128, 128, 2, c4));
  TRY(mtcnn_dense(e, wm, "dense5", "prelu5", c4, 128, 256, d5, keep));
  TRY(mtcnn_heads(e, wm, {{"dense6_1", 2}, {"dense6_2", 4}, {"dense6_3", 10}}, d5, 256, hd, 16));
  return VNF_OK;
}

// ---------------------------------------------------------------------------------------------
// RetinaFace with the MobileNetV1-0.25 backbone (models/retina_face.py:56-152, retina_face_utils/components.py,
// config.py cfg_mnet) as a plan on the exact-f32 core.  LeakyReLU is the PReLU epilogue with a constant slope;
// relu(cat(...)) of SSH is a ReLU in each branch's last conv, written into its channel slice (concat-free).
static bool bn_fold_at(WeightMap& wm, const std::string& p, int C, std::vector<float>& s, std::vector<float>& t) {
  return bn_fold(wm, p, C, 1e-5f, s, t);
}

int build_retina_mnet(Encoder& e, WeightMap& wm, int H, int W, int head_bufs[3]) {
  e.in_size = 0;
  auto down = [](int v) { return (v + 2 - 3) / 2 + 1; };
  // VNF_RETINA_FUSE=0: the early layers as plan convolutions on an NHWC4 fp32 copy of the frames (buffer 0, written by
  // the caller); default: conv0 straight from the u8 frames (Op::RSTEM) and dw+pw blocks in one kernel (Op::DWPW)
  // (bit 0: stem, bit 1: dw+pw blocks)
  static const int fuse_env = getenv("VNF_RETINA_FUSE") ? atoi(getenv("VNF_RETINA_FUSE")) : 3;
  const bool fused = fuse_env & 1, fused_dw = fuse_env & 2;
  int cur = e.add_buf(fused ? 1 : H, fused ? 1 : W, 4);   // input: NHWC4 (R-104, G-117, B-123, 0); a stub when fused
  int h = H, w = W;
  // conv (3x3 or 1x1) + BN + optional LeakyReLU / ReLU into (buf, channel offset)
  auto conv_bn = [&](const std::string& p, int xb, int cin, int cin_pad, int cout, int k, int stride, int ob, int ooff, int act,
                     float leaky) -> int {
    ConvSpec s;
    s.name = p; s.x_buf = xb; s.cin = cin; s.cin_pad = cin_pad; s.KH = s.KW = k; s.sh = s.sw = stride; s.ph = s.pw = k / 2;
    s.pieces.resize(1);
    Piece& pc = s.pieces[0];
    pc.w = wm.get(p + ".0.weight", (int64_t)cout * cin * k * k);
    pc.cout = pc.cout_pad = cout;
    if (!pc.w || !bn_fold_at(wm, p + ".1", cout, pc.scale, pc.bias)) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
    if (act == ACT_PRELU) pc.slope.assign(cout, leaky);
    s.segs.push_back({0, cout, ob, ooff});
    s.act = act;
    return add_conv(e, s);
  };
  // conv_dw(inp, oup, stride): depthwise 3x3 + BN + leaky 0.1, pointwise 1x1 + BN + leaky 0.1 (components.py:30-40)
  auto conv_dw = [&](const std::string& p, int inp, int oup, int stride) -> int {
    const float* dw = wm.get(p + ".0.weight", (int64_t)inp * 9);
    std::vector<float> sc, sh;
    if (!dw || !bn_fold_at(wm, p + ".1", inp, sc, sh)) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
    std::vector<float> w9c((size_t)9 * inp);
    for (int c = 0; c < inp; ++c)
      for (int t = 0; t < 9; ++t) w9c[(size_t)t * inp + c] = dw[(size_t)c * 9 + t] * sc[c];
    const int ho = stride == 2 ? down(h) : h, wo = stride == 2 ? down(w) : w;
    if (fused_dw && dwpw_supported(inp, oup)) {
      const float* pw = wm.get(p + ".3.weight", (int64_t)oup * inp);
      std::vector<float> ps, pb;
      if (!pw || !bn_fold_at(wm, p + ".4", oup, ps, pb)) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
      std::vector<float> pwf((size_t)oup * inp);
      for (int o = 0; o < oup; ++o)
        for (int c = 0; c < inp; ++c) pwf[(size_t)o * inp + c] = pw[(size_t)o * inp + c] * ps[o];
      DwPwLayer d;
      d.name = p; d.x_buf = cur; d.o_buf = e.add_buf(ho, wo, oup); d.cin = inp; d.cout = oup; d.stride = stride; d.slope = 0.1f;
      d.dw = (float*)e.upload(w9c.data(), w9c.size() * 4);
      d.dbias = (float*)e.upload(sh.data(), sh.size() * 4);
      d.pw = (float*)e.upload(pwf.data(), pwf.size() * 4);
      d.pbias = (float*)e.upload(pb.data(), pb.size() * 4);
      if (!d.dw || !d.dbias || !d.pw || !d.pbias) return VNF_E_HIP;
      e.dwpws.push_back(d);
      Op op; op.kind = Op::DWPW; op.a = (int)e.dwpws.size() - 1;
      e.ops.push_back(op);
      cur = d.o_buf; h = ho; w = wo;
      return VNF_OK;
    }
    DwLayer d;
    d.x_buf = cur; d.o_buf = e.add_buf(ho, wo, inp); d.C = inp; d.stride = stride; d.slope = 0.1f;
    d.w = (float*)e.upload(w9c.data(), w9c.size() * 4);
    d.bias = (float*)e.upload(sh.data(), sh.size() * 4);
    if (!d.w || !d.bias) return VNF_E_HIP;
    e.dws.push_back(d);
    Op op; op.kind = Op::DWCONV; op.a = (int)e.dws.size() - 1;
    e.ops.push_back(op);
    h = ho; w = wo;
    const int ob = e.add_buf(h, w, oup);
    // the pointwise half: Sequential indices 3 (conv) and 4 (bn)
    ConvSpec s;
    s.name = p + ".3"; s.x_buf = d.o_buf; s.cin = s.cin_pad = inp;
    s.pieces.resize(1);
    Piece& pc = s.pieces[0];
    pc.w = wm.get(p + ".3.weight", (int64_t)oup * inp);
    pc.cout = pc.cout_pad = oup;
    if (!pc.w || !bn_fold_at(wm, p + ".4", oup, pc.scale, pc.bias)) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
    pc.slope.assign(oup, 0.1f);
    s.segs.push_back({0, oup, ob, 0});
    s.act = ACT_PRELU;
    TRY(add_conv(e, s));
    cur = ob;
    return VNF_OK;
  };
  // ---- body (components.py:100-121)
  {
    const int ho = down(h), wo = down(w);
    const int ob = e.add_buf(ho, wo, 8);
    if (fused) {
      const float* w0 = wm.get("body.stage1.0.0.weight", 8 * 27);
      std::vector<float> sc, sh;
      if (!w0 || !bn_fold_at(wm, "body.stage1.0.1", 8, sc, sh)) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
      std::vector<float> wa(7 * 64, 0.f);
      for (int s7 = 0; s7 < 7; ++s7)
        for (int lane = 0; lane < 64; ++lane) {
          const int lg = lane >> 4, lm = lane & 15, k = 4 * s7 + lg;
          if (lm < 8 && k < 27) {
            const int tap = k / 3, c = k % 3;
            wa[s7 * 64 + lane] = w0[(lm * 3 + c) * 9 + tap] * sc[lm];
          }
        }
      e.rstem_wa = (float*)e.upload(wa.data(), wa.size() * 4);
      e.rstem_bias = (float*)e.upload(sh.data(), 8 * 4);
      if (!e.rstem_wa || !e.rstem_bias) return VNF_E_HIP;
      Op op; op.kind = Op::RSTEM; op.a = H; op.b = ob; op.c = W;
      e.ops.push_back(op);
    } else {
      TRY(conv_bn("body.stage1.0", cur, 3, 4, 8, 3, 2, ob, 0, ACT_PRELU, 0.1f));
    }
    cur = ob; h = ho; w = wo;
  }
  TRY(conv_dw("body.stage1.1", 8, 16, 1));
  TRY(conv_dw("body.stage1.2", 16, 32, 2));
  TRY(conv_dw("body.stage1.3", 32, 32, 1));
  TRY(conv_dw("body.stage1.4", 32, 64, 2));
  TRY(conv_dw("body.stage1.5", 64, 64, 1));
  const int c1 = cur, h1 = h, w1 = w;
  TRY(conv_dw("body.stage2.0", 64, 128, 2));
  for (int i = 1; i < 6; ++i) TRY(conv_dw("body.stage2." + std::to_string(i), 128, 128, 1));
  const int c2 = cur, h2 = h, w2 = w;
  TRY(conv_dw("body.stage3.0", 128, 256, 2));
  TRY(conv_dw("body.stage3.1", 256, 256, 1));
  const int c3 = cur, h3 = h, w3 = w;
  // ---- FPN (components.py:66-97; out_channels 64 -> leaky 0.1)
  const int o1 = e.add_buf(h1, w1, 64), o2 = e.add_buf(h2, w2, 64), o3 = e.add_buf(h3, w3, 64);
  TRY(conv_bn("fpn.output1", c1, 64, 64, 64, 1, 1, o1, 0, ACT_PRELU, 0.1f));
  TRY(conv_bn("fpn.output2", c2, 128, 128, 64, 1, 1, o2, 0, ACT_PRELU, 0.1f));
  TRY(conv_bn("fpn.output3", c3, 256, 256, 64, 1, 1, o3, 0, ACT_PRELU, 0.1f));
  { Op op; op.kind = Op::UPADD; op.a = o3; op.b = o2; e.ops.push_back(op); }
  const int m2 = e.add_buf(h2, w2, 64);
  TRY(conv_bn("fpn.merge2", o2, 64, 64, 64, 3, 1, m2, 0, ACT_PRELU, 0.1f));
  { Op op; op.kind = Op::UPADD; op.a = m2; op.b = o1; e.ops.push_back(op); }
  const int m1 = e.add_buf(h1, w1, 64);
  TRY(conv_bn("fpn.merge1", o1, 64, 64, 64, 3, 1, m1, 0, ACT_PRELU, 0.1f));
  // ---- SSH x3 + heads (components.py:42-64, retina_face.py:20-54,138-146)
  const int feat_in[3] = {m1, m2, o3}, fh[3] = {h1, h2, h3}, fw[3] = {w1, w2, w3};
  for (int l = 0; l < 3; ++l) {
    const std::string p = "ssh" + std::to_string(l + 1);
    const int cat = e.add_buf(fh[l], fw[l], 64), t5 = e.add_buf(fh[l], fw[l], 16), t7 = e.add_buf(fh[l], fw[l], 16);
    // convolutions that read the same tensor run as ONE GEMM whose column ranges go to different tensors (ReLU = a
    // PReLU slope of 0, LeakyReLU = 0.1 on the other range): conv3X3 | conv5X5_1 on the level's feature map,
    // conv5X5_2 | conv7X7_2 on conv5X5_1's output
    auto conv_pair = [&](const std::string& pa, int na, int ba, int oa, float sa, const std::string& pb, int nb, int bb, int ob2,
                         float sb, int xb, int cin) -> int {
      ConvSpec s;
      s.name = pa + "|" + pb.substr(pb.rfind('.') + 1); s.x_buf = xb; s.cin = s.cin_pad = cin; s.KH = s.KW = 3; s.ph = s.pw = 1;
      s.pieces.resize(2);
      const std::string nm[2] = {pa, pb};
      const int nn[2] = {na, nb};
      const float sl[2] = {sa, sb};
      for (int k = 0; k < 2; ++k) {
        Piece& pc = s.pieces[k];
        pc.w = wm.get(nm[k] + ".0.weight", (int64_t)nn[k] * cin * 9);
        pc.cout = pc.cout_pad = nn[k];
        if (!pc.w || !bn_fold_at(wm, nm[k] + ".1", nn[k], pc.scale, pc.bias)) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
        pc.slope.assign(nn[k], sl[k]);
      }
      s.segs.push_back({0, na, ba, oa});
      s.segs.push_back({na, na + nb, bb, ob2});
      s.act = ACT_PRELU;
      return add_conv(e, s);
    };
    TRY(conv_pair(p + ".conv3X3", 32, cat, 0, 0.f, p + ".conv5X5_1", 16, t5, 0, 0.1f, feat_in[l], 64));
    TRY(conv_pair(p + ".conv5X5_2", 16, cat, 32, 0.f, p + ".conv7X7_2", 16, t7, 0, 0.1f, t5, 16));
    TRY(conv_bn(p + ".conv7x7_3", t7, 16, 16, 16, 3, 1, cat, 48, ACT_RELU, 0.f));
    // the three 1x1 heads of the level as one GEMM: columns [class 4 | bbox 8 | landmark 20]
    const int hb = e.add_buf(fh[l], fw[l], 32);
    ConvSpec s;
    s.name = "heads" + std::to_string(l); s.x_buf = cat; s.cin = s.cin_pad = 64;
    s.pieces.resize(3);
    const char* hn[3] = {"ClassHead.", "BboxHead.", "LandmarkHead."};
    const int hc[3] = {4, 8, 20};
    for (int k = 0; k < 3; ++k) {
      Piece& pc = s.pieces[k];
      const std::string q = std::string(hn[k]) + std::to_string(l) + ".conv1x1";
      pc.w = wm.get(q + ".weight", (int64_t)hc[k] * 64);
      const float* b = wm.get(q + ".bias", hc[k]);
      if (!pc.w || !b) return fail(VNF_E_MISSING, "retina: missing weight " + wm.missing);
      pc.cout = pc.cout_pad = hc[k];
      pc.bias.assign(b, b + hc[k]);
    }
    s.segs.push_back({0, 32, hb, 0});
    s.act = ACT_NONE;
    TRY(add_conv(e, s));
    head_bufs[l] = hb;
  }
  return VNF_OK;
}

// Images are independent, so a batch is cut into `nstreams` contiguous parts that run the whole plan
// concurrently on side streams (fork / join with events on the caller's stream): the small late
// layers (a few hundred workgroups, latency-bound) of one part fill the CUs the other leaves idle,
// and kernel-boundary drains overlap.
Encoder::~Encoder() {
  for (int i = 0; i < 4; ++i) {
    if (join_ev[i]) (void)hipEventDestroy(join_ev[i]);
    if (side[i]) (void)hipStreamDestroy(side[i]);
  }
  if (fork_ev) (void)hipEventDestroy(fork_ev);
  for (hipEvent_t ev : ctx_ev)
    if (ev) (void)hipEventDestroy(ev);
}

int Encoder::select_ctx(hipStream_t s, int* used) {
  const int es = dtype_size(dtype);
  if (ctx_bufs.empty()) {  // context 0 = the buffers finalize() allocated
    ctx_bufs.emplace_back();
    for (auto& b : bufs) ctx_bufs[0].push_back(b.ptr);
    ctx_emb.push_back(emb_raw);
    ctx_ev.push_back(nullptr);
  }
  const int c = next_ctx % n_ctx;
  next_ctx = (c + 1) % n_ctx;
  while ((int)ctx_bufs.size() <= c) {
    std::vector<char*> set;
    for (auto& b : bufs) {
      const size_t bytes = b.elems_per_image() * es * (size_t)max_batch;
      char* p = (char*)dalloc(bytes);
      if (!p) return VNF_E_HIP;
      // on the caller's stream: ordered before this call's kernels (a null-stream hipMemset is not ordered against
      // non-blocking streams and may land after the first layers have written their outputs)
      VNF_HIP(hipMemsetAsync(p, 0, bytes, s));
      set.push_back(p);
    }
    float* er = (float*)dalloc((size_t)max_batch * 512 * 4);
    if (!er) return VNF_E_HIP;
    ctx_bufs.push_back(set);
    ctx_emb.push_back(er);
    ctx_ev.push_back(nullptr);
  }
  for (size_t i = 0; i < bufs.size(); ++i) bufs[i].ptr = ctx_bufs[c][i];
  emb_raw = ctx_emb[c];
  if (ctx_ev[c]) VNF_HIP(hipStreamWaitEvent(s, ctx_ev[c], 0));
  *used = c;
  return VNF_OK;
}

int Encoder::run(const void* x, int n, int x_dtype, float* out, hipStream_t s, std::string* report) {
  if (n < 0 || n > max_batch) return fail(VNF_E_CAPACITY, "batch exceeds max_batch");
  if (n == 0) return VNF_OK;
  if (tune_dirty) {
    tune_dirty = false;
    const int rc = autotune();
    if (rc != VNF_OK) return rc;
  }
  if (n_ctx > 1) {
    int c = 0;
    int rc = select_ctx(s, &c);
    if (rc != VNF_OK) return rc;
    const int keep = n_ctx;
    n_ctx = 1;  // the body below runs once on the selected set
    rc = run(x, n, x_dtype, out, s, report);
    n_ctx = keep;
    if (rc != VNF_OK) return rc;
    if (!ctx_ev[c]) VNF_HIP(hipEventCreateWithFlags(&ctx_ev[c], hipEventDisableTiming));
    VNF_HIP(hipEventRecord(ctx_ev[c], s));
    return VNF_OK;
  }
  static const int env_streams = getenv("VNF_STREAMS") ? atoi(getenv("VNF_STREAMS")) : 2;
  int ns = env_streams < 1 ? 1 : (env_streams > 4 ? 4 : env_streams);
  if (ns > max_streams) ns = max_streams;
  while (ns > 1 && n / ns < 96) --ns;  // below ~100 images a part no longer fills the chip: fixed per-launch latency dominates
  if (ns == 1 || report) return run_range(x, 0, n, x_dtype, out, s, report);
  if (!side[0]) {
    for (int i = 0; i < 4; ++i) {
      VNF_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
      VNF_HIP(hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming));
    }
    VNF_HIP(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
  }
  VNF_HIP(hipEventRecord(fork_ev, s));
  int rc = VNF_OK;
  for (int i = 0; i < ns && rc == VNF_OK; ++i) {
    const int i0 = (int)((long long)n * i / ns), i1 = (int)((long long)n * (i + 1) / ns);
    VNF_HIP(hipStreamWaitEvent(side[i], fork_ev, 0));
    rc = run_range(x, i0, i1, x_dtype, out, side[i], nullptr);
    VNF_HIP(hipEventRecord(join_ev[i], side[i]));
    VNF_HIP(hipStreamWaitEvent(s, join_ev[i], 0));
  }
  return rc;
}

int Encoder::run_range(const void* x, int i0, int i1, int x_dtype, float* out, hipStream_t s, std::string* report) {
  const int n = i1 - i0;
  const int es = dtype_size(dtype);
  const int xes = dtype_size(x_dtype);
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_op, prof_n;
  for (const Group& g : groups) {
    const int step = g.chunk < n ? g.chunk : n;
    for (int n0 = i0; n0 < i1; n0 += step) {
      const int nn = (i1 - n0) < step ? (i1 - n0) : step;
      for (int oi = g.first; oi < g.last; ++oi) {
        const Op& op = ops[oi];
        const FusedStack* fs = nullptr;
        for (const FusedStack& f : fused)
          if (f.active && oi == f.first) fs = &f;
        if (report) {
          hipEvent_t e0;
          VNF_HIP(hipEventCreate(&e0));
          VNF_HIP(hipEventRecord(e0, s));
          prof_ev.push_back(e0);
          prof_op.push_back(oi);
          prof_n.push_back(nn);
        }
        if (fs && fs->kind == 2) {
          const Buf& ib = bufs[fs->in_buf];
          const Buf& ob = bufs[fs->out_buf];
          StemMidArgs sa;
          sa.x = ib.ptr + (size_t)n0 * ib.elems_per_image() * es;
          sa.y = ob.ptr + (size_t)n0 * ob.elems_per_image() * es;
          sa.ldx = ib.C; sa.ldy = ob.C; sa.n = nn;
          sa.wfrag = fs->wstream; sa.bias = fs->bias;
          sa.w3b = nullptr; sa.b3b = nullptr; sa.k3b_pad = 0;
          if (fs->ext) {
            const ConvLayer& c3b = convs[fs->ext_conv];
            const Buf& eb = bufs[fs->ext_out_buf];
            sa.y = eb.ptr + (size_t)n0 * eb.elems_per_image() * es;
            sa.ldy = eb.C;
            sa.w3b = c3b.w; sa.b3b = c3b.bias; sa.k3b_pad = c3b.Kpad;
          }
          hipError_t err = dtype == F16P ? launch_stem_mids(sa, s) : launch_stem_mid(sa, dtype, s);
          if (err != hipSuccess) return fail(VNF_E_HIP, std::string("fused stem: ") + hipGetErrorString(err));
          oi = (fs->ext ? fs->ext_last : fs->last) - 1;
          continue;
        }
        if (fs && fs->kind == 35 && fs->stack) {
          const Buf& ib = bufs[convs[fs->conv0 + 4].res_buf];                               // first block's input
          const Buf& ob = bufs[convs[fs->conv0 + 5 * (fs->nblocks - 1) + 4].seg[0].buf];    // last block's output
          Block35StackArgs ba;
          ba.x = ib.ptr + (size_t)n0 * ib.elems_per_image() * es;
          ba.y = ob.ptr + (size_t)n0 * ob.elems_per_image() * es;
          ba.ldx = ib.C; ba.ldy = ob.C; ba.n = nn; ba.nblocks = fs->nblocks;
          ba.wimg = fs->wstream;
          if (fs->ext) {
            const Buf& tb = bufs[fs->ext_out_buf];
            ba.wtail = fs->wtail;
            ba.ytail = tb.ptr + (size_t)n0 * tb.elems_per_image() * es;
            ba.ldyt = tb.C;
          }
          hipError_t err = launch_block35_stack(ba, dtype, s);
          if (err != hipSuccess) return fail(VNF_E_HIP, std::string("fused Block35 stack: ") + hipGetErrorString(err));
          oi = (fs->ext ? fs->ext_last : fs->last) - 1;
          continue;
        }
        if (fs && fs->kind == 35) {
          for (int b = 0; b < fs->nblocks; ++b) {
            const ConvLayer& up = convs[fs->conv0 + 5 * b + 4];   // residual source = block input, segment 0 = block output
            const Buf& ib = bufs[up.res_buf];
            const Buf& ob = bufs[up.seg[0].buf];
            Block35Args ba;
            ba.x = ib.ptr + (size_t)n0 * ib.elems_per_image() * es;
            ba.y = ob.ptr + (size_t)n0 * ob.elems_per_image() * es;
            ba.ldx = ib.C; ba.ldy = ob.C; ba.n = nn;
            ba.wimg = (const char*)fs->wstream + (size_t)b * (dtype == F16P ? B35S_WIMG_BYTES : B35_WIMG_BYTES);
            ba.zero = conv_zero_page();
            hipError_t err = dtype == F16P ? launch_block35s(ba, s) : launch_block35(ba, dtype, s);
            if (err != hipSuccess) return fail(VNF_E_HIP, std::string("fused Block35: ") + hipGetErrorString(err));
          }
          oi = fs->last - 1;
          continue;
        }
        if (fs) {
          const Buf& ib = bufs[fs->in_buf];
          const Buf& ob = bufs[fs->out_buf];
          Trunk17Args ta;
          ta.x = ib.ptr + (size_t)n0 * ib.elems_per_image() * es;
          ta.y = ob.ptr + (size_t)n0 * ob.elems_per_image() * es;
          ta.ldx = ib.C; ta.ldy = ob.C; ta.n = nn; ta.nblocks = fs->nblocks;
          ta.wstream = fs->wstream; ta.bias = fs->bias;
          hipError_t err = dtype == F16P ? launch_trunk17s(ta, s) : launch_trunk17(ta, dtype, s);
          if (err != hipSuccess) return fail(VNF_E_HIP, std::string("fused Block17 stack: ") + hipGetErrorString(err));
          oi = fs->last - 1;
          continue;
        }
        switch (op.kind) {
          case Op::PACK: {
            const Buf& b = bufs[op.a];
            const char* src = (const char*)x + (size_t)n0 * 3 * in_size * in_size * xes;
            VNF_HIP(launch_pack_input(src, x_dtype, b.ptr + (size_t)n0 * b.elems_per_image() * es, dtype, nn,
                                      in_size * in_size, s));
            break;
          }
          case Op::CONV: {
            const ConvLayer& L = convs[op.a];
            const ConvArgs a = conv_args(L, n0, nn);
            hipError_t err = launch_conv(a, s);
            if (err != hipSuccess) return fail(VNF_E_HIP, L.name + ": " + hipGetErrorString(err));
            break;
          }
          case Op::STEM1: {
            const Buf& ob = bufs[op.b];
            const char* src = (const char*)x + (size_t)n0 * 3 * in_size * in_size * xes;
            VNF_HIP(launch_stem_conv1a(src, x_dtype, ob.ptr + (size_t)n0 * ob.elems_per_image() * es, ob.C, dtype, nn,
                                       stem_wt, s));
            break;
          }
          case Op::MAXPOOL: {
            const Buf& ib = bufs[op.a];
            const Buf& ob = bufs[op.b];
            VNF_HIP(launch_maxpool3s2(ib.ptr + (size_t)n0 * ib.elems_per_image() * es, ib.C,
                                      ob.ptr + ((size_t)n0 * ob.elems_per_image() + op.c) * es, ob.C, dtype, nn, ib.H,
                                      ib.W, ib.C, s));
            break;
          }
          case Op::AVGPOOL: {
            const Buf& ib = bufs[op.a];
            const Buf& ob = bufs[op.b];
            VNF_HIP(launch_avgpool(ib.ptr + (size_t)n0 * ib.elems_per_image() * es, ib.C,
                                   ob.ptr + (size_t)n0 * ob.elems_per_image() * es, dtype, nn, ib.H * ib.W, ib.C, s));
            break;
          }
          case Op::L2NORM:
            VNF_HIP(launch_l2norm(emb_raw + (size_t)n0 * 512, out + (size_t)n0 * 512, nn, 512, s));
            break;
          case Op::MAXPOOLC: {
            const Buf& ib = bufs[op.a];
            const Buf& ob = bufs[op.b];
            VNF_HIP(launch_maxpool_ceil(ib.ptr + (size_t)n0 * ib.elems_per_image() * es, ib.C,
                                        ob.ptr + (size_t)n0 * ob.elems_per_image() * es, ob.C, dtype, nn, ib.H, ib.W, ib.C,
                                        op.c, s));
            break;
          }
          case Op::DWCONV: {
            if (dtype != F32 && dtype != F16X2) return fail(VNF_E_INVALID, "depthwise conv: fp32 / split-f16 plans only");
            const DwLayer& d = dws[op.a];
            const Buf& ib = bufs[d.x_buf];
            const Buf& ob = bufs[d.o_buf];
            VNF_HIP(launch_dwconv3x3((const float*)ib.ptr + (size_t)n0 * ib.elems_per_image(),
                                     (float*)ob.ptr + (size_t)n0 * ob.elems_per_image(), nn, ib.H, ib.W, d.C, d.stride, d.w, d.bias,
                                     d.slope, dtype == F16X2, s));
            break;
          }
          case Op::RSTEM: {   // a = H, c = W of the u8 frames the caller passes as x; b = output buffer
            if ((dtype != F32 && dtype != F16X2) || !x) return fail(VNF_E_INVALID, "retina stem: fp32 / split-f16 plans on caller frames only");
            const Buf& ob = bufs[op.b];
            VNF_HIP(launch_retina_stem((const uint8_t*)x + (size_t)n0 * op.a * op.c * 3, nn, op.a, op.c, rstem_wa, rstem_bias, 0.1f,
                                       (float*)ob.ptr + (size_t)n0 * ob.elems_per_image(), dtype == F16X2, s));
            break;
          }
          case Op::DWPW: {
            if (dtype != F32 && dtype != F16X2) return fail(VNF_E_INVALID, "dw+pw: fp32 / split-f16 plans only");
            const DwPwLayer& d = dwpws[op.a];
            const Buf& ib = bufs[d.x_buf];
            const Buf& ob = bufs[d.o_buf];
            VNF_HIP(launch_dwpw((const float*)ib.ptr + (size_t)n0 * ib.elems_per_image(), (float*)ob.ptr + (size_t)n0 * ob.elems_per_image(),
                                nn, ib.H, ib.W, d.cin, d.cout, d.stride, d.dw, d.dbias, d.slope, d.pw, d.pbias, d.slope, dtype == F16X2, s));
            break;
          }
          case Op::UPADD: {
            if (dtype != F32 && dtype != F16X2) return fail(VNF_E_INVALID, "upsample-add: fp32 / split-f16 plans only");
            const Buf& ib = bufs[op.a];
            const Buf& ob = bufs[op.b];
            VNF_HIP(launch_upsample_add((const float*)ib.ptr + (size_t)n0 * ib.elems_per_image(), ib.H, ib.W,
                                        (float*)ob.ptr + (size_t)n0 * ob.elems_per_image(), ob.H, ob.W, ob.C, nn, dtype == F16X2, s));
            break;
          }
          case Op::COPYOUT:
            VNF_HIP(hipMemcpyAsync(out + (size_t)n0 * 512, emb_raw + (size_t)n0 * 512, (size_t)nn * 512 * 4,
                                   hipMemcpyDeviceToDevice, s));
            break;
        }
      }
    }
  }
  if (report) {
    // per-op device time (events between consecutive launches on the stream), summed over chunks
    hipEvent_t e_end;
    VNF_HIP(hipEventCreate(&e_end));
    VNF_HIP(hipEventRecord(e_end, s));
    VNF_HIP(hipEventSynchronize(e_end));
    prof_ev.push_back(e_end);
    std::vector<double> ms(ops.size(), 0.0);
    for (size_t i = 0; i + 1 < prof_ev.size(); ++i) {
      float t = 0;
      VNF_HIP(hipEventElapsedTime(&t, prof_ev[i], prof_ev[i + 1]));
      ms[prof_op[i]] += t;
    }
    for (auto ev : prof_ev) (void)hipEventDestroy(ev);
    char line[512];
    double total = 0;
    for (size_t oi = 0; oi < ops.size(); ++oi) {
      const Op& op = ops[oi];
      total += ms[oi];
      const FusedStack* fs = nullptr;
      for (const FusedStack& f : fused)
        if (f.active && (int)oi >= f.first && (int)oi < (f.ext ? f.ext_last : f.last)) fs = &f;
      if (fs) {
        if ((int)oi != fs->first) continue;
        const double gf = 2.0 * fs->macs_alg * n / 1e9;
        snprintf(line, sizeof line, "%-28s %-60s %8.4f ms  %8.1f GFLOP %8.1f TFLOP/s\n",
                 fs->kind == 35 ? "repeat_1 (fused blocks)" : fs->kind == 2 ? (fs->ext ? "conv2d_2a+2b+maxpool_3a+3b" : "conv2d_2a+2b+maxpool_3a") : "repeat_2 (persistent trunk)",
                 fs->kind == 35 ? (fs->stack ? (fs->ext ? "5 x Block35 + mixed_6a.branch1.0 in one launch, x in registers" : "5 x Block35 in one launch, x in registers, one workgroup per image")
                                             : "5 x Block35, one launch per block, one workgroup per image")
                 : fs->kind == 2 ? "rolling rows, one launch, one workgroup per image"
                                 : "10 x Block17 in one launch, one workgroup per image",
                 ms[oi], gf, ms[oi] > 0 ? gf / ms[oi] : 0.0);
        *report += line;
        continue;
      }
      if (op.kind == Op::CONV) {
        const ConvLayer& L = convs[op.a];
        const double gf = 2.0 * L.macs_alg * n / 1e9;
        snprintf(line, sizeof line, "%-28s conv M/img=%-6d N=%-5d K=%-5d %dx%d s%d cfg%-2d %8.4f ms  %8.1f GFLOP %8.1f TFLOP/s\n",
                 L.name.c_str(), L.Ho * L.Wo, L.cout, L.K, L.KH, L.KW, L.sh, L.cfg, ms[oi], gf, ms[oi] > 0 ? gf / ms[oi] : 0.0);
      } else if (op.kind == Op::STEM1) {
        const double gf = 2.0 * convs[op.a].macs_alg * n / 1e9;
        snprintf(line, sizeof line, "%-28s %-60s %8.4f ms  %8.1f GFLOP %8.1f TFLOP/s\n", "conv2d_1a (direct, NCHW in)",
                 "3x3 s2 3->32 on the caller's tensor, exact f32 (MFMA / VALU)", ms[oi], gf, ms[oi] > 0 ? gf / ms[oi] : 0.0);
      } else {
        static const char* kn[] = {"pack", "conv", "maxpool", "avgpool", "l2norm", "copyout", "maxpool_ceil", "stem1", "dwconv3x3",
                                   "upsample_add", "retina_stem (u8 frames -> conv0)", "dw3x3+pw1x1 fused"};
        snprintf(line, sizeof line, "%-28s %-8s %60s %8.4f ms\n", "", kn[op.kind], "", ms[oi]);
      }
      *report += line;
    }
    snprintf(line, sizeof line, "TOTAL %.4f ms for n=%d\n", total, n);
    *report += line;
  }
  return VNF_OK;
}

}  // namespace vnf
