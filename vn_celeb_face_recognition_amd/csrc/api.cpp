// extern "C" surface of libvnface.so (declared in include/vnface.h).  No exception crosses it.
#include <cstring>
#include <new>

#include "engine.h"
#include <cstdlib>

namespace vnf {
const char* last_error_cstr();
}
using namespace vnf;

#define API_GUARD_BEGIN try {
#define API_GUARD_END                                                  \
  }                                                                    \
  catch (const std::bad_alloc&) { return fail(VNF_E_INVALID, "host out of memory"); } \
  catch (const std::exception& ex) { return fail(VNF_E_INVALID, std::string("exception: ") + ex.what()); } \
  catch (...) { return fail(VNF_E_INVALID, "unknown exception"); }

extern "C" {

const char* vnf_last_error(void) { return last_error_cstr(); }
const char* vnf_version(void) { return "vnface 0.1 (gfx950)"; }

int vnf_init(int device_ordinal) {
  API_GUARD_BEGIN
  int n = 0;
  VNF_HIP(hipGetDeviceCount(&n));
  if (device_ordinal < 0 || device_ordinal >= n) return fail(VNF_E_INVALID, "no such device");
  VNF_HIP(hipSetDevice(device_ordinal));
  hipDeviceProp_t p;
  VNF_HIP(hipGetDeviceProperties(&p, device_ordinal));
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
    return fail(VNF_E_INVALID, std::string("libvnface is built for gfx950 only, device is ") + p.gcnArchName);
  return VNF_OK;
  API_GUARD_END
}

int vnf_destroy(vnf_handle h) {
  API_GUARD_BEGIN
  delete reinterpret_cast<HandleBase*>(h);
  return VNF_OK;
  API_GUARD_END
}

int vnf_encoder_create(int arch, const vnf_tensor_desc* weights, int n_weights, int compute_dtype, int max_batch,
                       vnf_handle* out) {
  API_GUARD_BEGIN
  if (!out || !weights || n_weights <= 0 || max_batch <= 0) return fail(VNF_E_INVALID, "bad argument");
  if (compute_dtype != VNF_F32 && compute_dtype != VNF_BF16 && compute_dtype != VNF_F16 && compute_dtype != VNF_F16X2)
    return fail(VNF_E_INVALID, "compute_dtype must be VNF_F32, VNF_BF16, VNF_F16 or VNF_F16X2");
  *out = nullptr;
  Encoder* e = new Encoder();
  e->kind = 1;
  e->arch = arch;
  e->dtype = compute_dtype;  // VNF_F32/BF16/F16/F16X2 == vnf::F32/BF16/F16/F16X2
  // the encoders keep split-f16 tensors PLANAR (8-channel units [8 hi][8 lo]; three MFMAs per product, split_f16.h);
  // VNF_SPLIT_LAYOUT=interleaved restores the (hi, lo)-pair layout of the first version (four MFMA-equivalents)
  if (compute_dtype == VNF_F16X2) {
    const char* lay = getenv("VNF_SPLIT_LAYOUT");
    if (!lay || std::string(lay) != "interleaved") e->dtype = F16P;
  }
  e->max_batch = max_batch;
  (void)hipGetDevice(&e->device);
  WeightMap wm(weights, n_weights);
  int r = arch == VNF_ARCH_IRV1 ? build_irv1(*e, wm) : arch == VNF_ARCH_IR100 ? build_ir100(*e, wm)
                                                                               : fail(VNF_E_INVALID, "unknown arch");
  if (r == VNF_OK) r = e->finalize();
  if (r != VNF_OK) {
    delete e;
    return r;
  }
  VNF_HIP(hipDeviceSynchronize());
  *out = reinterpret_cast<vnf_handle>(static_cast<HandleBase*>(e));
  return VNF_OK;
  API_GUARD_END
}

static Encoder* as_encoder(vnf_handle h) {
  HandleBase* b = reinterpret_cast<HandleBase*>(h);
  return (b && b->kind == 1) ? static_cast<Encoder*>(b) : nullptr;
}

int vnf_embed(vnf_handle h, const void* x, int n, int x_dtype, float* emb_out, void* stream) {
  API_GUARD_BEGIN
  Encoder* e = as_encoder(h);
  if (!e) return fail(VNF_E_INVALID, "not an encoder handle");
  if (n < 0 || (n > 0 && (!x || !emb_out))) return fail(VNF_E_INVALID, "bad argument");
  if (x_dtype != VNF_F32 && x_dtype != VNF_BF16 && x_dtype != VNF_F16) return fail(VNF_E_INVALID, "bad x_dtype");
  return e->run(x, n, x_dtype, emb_out, (hipStream_t)stream);
  API_GUARD_END
}

int vnf_encoder_tap(vnf_handle h, const char* name, int n, float* host_out, int64_t capacity, int64_t shape_out[4]) {
  API_GUARD_BEGIN
  Encoder* e = as_encoder(h);
  if (!e || !name) return fail(VNF_E_INVALID, "not an encoder handle");
  auto it = e->taps.find(name);
  if (it == e->taps.end()) return fail(VNF_E_INVALID, std::string("no such tap: ") + name);
  if (!e->buf_materialised(it->second.buf))
    return fail(VNF_E_INVALID, std::string("tap '") + name + "' is computed inside a fused kernel and never reaches memory in this "
                "compute dtype: create the encoder under VNF_FUSE=0 (or use a dtype without fused stacks) to read it");
  const Buf& b = e->bufs[it->second.buf];
  const int C = it->second.C;
  const int64_t total = (int64_t)n * C * b.H * b.W;
  if (shape_out) { shape_out[0] = n; shape_out[1] = C; shape_out[2] = b.H; shape_out[3] = b.W; }
  if (n > e->max_batch || total > capacity || !host_out) return fail(VNF_E_CAPACITY, "tap capacity");
  float* tmp = nullptr;
  VNF_HIP(hipMalloc(&tmp, (size_t)total * 4));
  hipError_t err = launch_nhwc_to_nchw_f32(b.ptr + (size_t)it->second.coff * dtype_size(e->dtype), b.C, e->dtype, tmp, n,
                                           b.H * b.W, C, 0);
  if (err == hipSuccess) err = hipMemcpy(host_out, tmp, (size_t)total * 4, hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (err != hipSuccess) return fail(VNF_E_HIP, hipGetErrorString(err));
  return VNF_OK;
  API_GUARD_END
}

int vnf_encoder_profile(vnf_handle h, const void* x, int n, int x_dtype, float* emb_out, void* stream, char* report,
                        int64_t capacity) {
  API_GUARD_BEGIN
  Encoder* e = as_encoder(h);
  if (!e || !report || capacity <= 0) return fail(VNF_E_INVALID, "bad argument");
  std::string rep;
  int r = e->run(x, n, x_dtype, emb_out, (hipStream_t)stream, &rep);
  if (r != VNF_OK) return r;
  strncpy(report, rep.c_str(), (size_t)capacity - 1);
  report[capacity - 1] = 0;
  return VNF_OK;
  API_GUARD_END
}

int vnf_encoder_flops(vnf_handle h, double* algorithmic, double* executed) {
  API_GUARD_BEGIN
  Encoder* e = as_encoder(h);
  if (!e) return fail(VNF_E_INVALID, "not an encoder handle");
  if (algorithmic) *algorithmic = 2.0 * e->macs_alg;
  if (executed) *executed = 2.0 * e->macs_exec;
  return VNF_OK;
  API_GUARD_END
}

int vnf_encoder_set_streams(vnf_handle h, int max_streams) {
  API_GUARD_BEGIN
  Encoder* e = as_encoder(h);
  if (!e) return fail(VNF_E_INVALID, "not an encoder handle");
  if (max_streams < 1 || max_streams > 4) return fail(VNF_E_INVALID, "vnf_encoder_set_streams: 1..4");
  if (e->max_streams != max_streams) {
    e->max_streams = max_streams;
    e->tune_dirty = true;  // the part size the layers see changed: the next vnf_embed picks the tiles again
  }
  return VNF_OK;
  API_GUARD_END
}

int vnf_encoder_set_contexts(vnf_handle h, int n) {
  API_GUARD_BEGIN
  Encoder* e = as_encoder(h);
  if (!e) return fail(VNF_E_INVALID, "not an encoder handle");
  if (n < 1 || n > 4) return fail(VNF_E_INVALID, "vnf_encoder_set_contexts: 1..4");
  e->n_ctx = n;
  e->next_ctx = 0;
  if (e->tune_lanes != n) {
    e->tune_lanes = n;
    e->tune_dirty = true;  // the next vnf_embed picks the tiles for `n` kernels sharing the GPU
  }
  return VNF_OK;
  API_GUARD_END
}

}  // extern "C"
