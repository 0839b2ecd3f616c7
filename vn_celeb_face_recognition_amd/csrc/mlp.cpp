// MLP identity classifier (models/mlp_model.py:5-15) + the argmax / probability part of
// identify_person (demo_image.py:125-129), fp32 end to end (the <=1e-4 log-prob gate):
//   dense_1 (512->2048) + ReLU and dense_2 (2048->C) run on the exact-f32 MFMA GEMM core
//   (a linear layer is a 1x1 convolution over a 1x1 image), then one wave per face does
//   log_softmax, argmax and exp.
#include "engine.h"

namespace vnf {

struct Mlp : HandleBase {
  Encoder enc;  // owns the packed layers and the three activation buffers
  int input_dim = 0, num_classes = 0, cpad = 0, max_batch = 0;
  int b_in = -1, b_h = -1, b_logit = -1;
  // the handle has ONE set of activation buffers: calls issued on different streams (the pipeline's rotating embedding
  // lanes) are ordered against each other through this event, so a later call cannot overwrite what an earlier one
  // is still reading
  hipEvent_t done = nullptr;
  ~Mlp() override { if (done) (void)hipEventDestroy(done); }
};

int add_linear(Encoder& e, const std::string& name, const float* w, const float* b, int cin, int cout, int cout_pad,
               int x_buf, int o_buf, int act);

}  // namespace vnf
using namespace vnf;

extern "C" int vnf_mlp_create(const vnf_tensor_desc* weights, int n_weights, int input_dim, int num_classes,
                              int max_batch, vnf_handle* out) {
  try {
    if (!out || !weights || input_dim <= 0 || num_classes <= 0 || max_batch <= 0 || input_dim % 8)
      return fail(VNF_E_INVALID, "vnf_mlp_create: bad argument");
    *out = nullptr;
    WeightMap wm(weights, n_weights);
    const float* w1 = wm.get("dense_1.weight", (int64_t)2048 * input_dim);
    const float* b1 = wm.get("dense_1.bias", 2048);
    const float* w2 = wm.get("dense_2.weight", (int64_t)num_classes * 2048);
    const float* b2 = wm.get("dense_2.bias", num_classes);
    if (!w1 || !b1 || !w2 || !b2) return fail(VNF_E_MISSING, "vnf_mlp_create: missing weight: " + wm.missing);
    Mlp* m = new Mlp();
    m->kind = 2;
    m->input_dim = input_dim; m->num_classes = num_classes; m->max_batch = max_batch;
    m->cpad = (num_classes + 7) / 8 * 8;
    Encoder& e = m->enc;
    e.dtype = F32; e.max_batch = max_batch; e.in_size = 1; e.arch = -1;
    m->b_in = e.add_buf(1, 1, input_dim);
    m->b_h = e.add_buf(1, 1, 2048);
    m->b_logit = e.add_buf(1, 1, m->cpad);
    int r = add_linear(e, "dense_1", w1, b1, input_dim, 2048, 2048, m->b_in, m->b_h, ACT_RELU);
    if (r == VNF_OK) r = add_linear(e, "dense_2", w2, b2, 2048, num_classes, m->cpad, m->b_h, m->b_logit, ACT_NONE);
    if (r == VNF_OK) r = e.finalize();
    if (r != VNF_OK) { delete m; return r; }
    VNF_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<vnf_handle>(static_cast<HandleBase*>(m));
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

extern "C" int vnf_classify(vnf_handle h, const float* emb, int f, float* logp_out, int32_t* argmax_out, float* prob_out,
                            void* stream) {
  try {
    HandleBase* hb = reinterpret_cast<HandleBase*>(h);
    if (!hb || hb->kind != 2) return fail(VNF_E_INVALID, "not an MLP handle");
    Mlp* m = static_cast<Mlp*>(hb);
    if (f < 0 || f > m->max_batch) return fail(VNF_E_CAPACITY, "vnf_classify: batch exceeds max_batch");
    if (f == 0) return VNF_OK;
    if (!emb) return fail(VNF_E_INVALID, "vnf_classify: bad argument");
    hipStream_t s = (hipStream_t)stream;
    Encoder& e = m->enc;
    if (m->done) VNF_HIP(hipStreamWaitEvent(s, m->done, 0));
    else VNF_HIP(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
    VNF_HIP(hipMemcpyAsync(e.bufs[m->b_in].ptr, emb, (size_t)f * m->input_dim * 4, hipMemcpyDeviceToDevice, s));
    int r = e.run(nullptr, f, VNF_F32, nullptr, s);
    if (r != VNF_OK) return r;
    VNF_HIP(launch_logsoftmax_argmax((const float*)e.bufs[m->b_logit].ptr, m->cpad, m->num_classes, f, logp_out,
                                     argmax_out, prob_out, s));
    VNF_HIP(hipEventRecord(m->done, s));
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}
