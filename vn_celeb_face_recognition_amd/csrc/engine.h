// Host-side engine shared by the encoder / classifier / detector handles of libvnface.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vnface.h"
#include "kernels.h"

namespace vnf {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define VNF_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return ::vnf::fail(VNF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

struct HandleBase {
  int kind = 0;  // 1 encoder, 2 mlp, 3 mtcnn
  int device = 0;
  std::vector<void*> allocs;  // device allocations owned by the handle
  virtual ~HandleBase();
  void* dalloc(size_t bytes);  // hipMalloc + bookkeeping (nullptr on failure, error set)
  void* upload(const void* host, size_t bytes);
};

// state_dict lookup -----------------------------------------------------------------------
struct WeightMap {
  std::unordered_map<std::string, const vnf_tensor_desc*> m;
  std::string missing;
  WeightMap(const vnf_tensor_desc* w, int n);
  const float* get(const std::string& name, int64_t numel);  // nullptr (+missing recorded) if absent / wrong size
  bool has(const std::string& name) const { return m.count(name) != 0; }
};

// host float -> storage dtype ----------------------------------------------------------------
void convert_to(int dtype, const float* src, void* dst, size_t n);

// ---------------------------------------------------------------------------------------------
// Encoder: a static plan of convolutions / pools over NHWC buffers.
struct Buf {
  int H, W, C;
  size_t elems_per_image() const { return (size_t)H * W * C; }
  char* ptr = nullptr;
};

struct ConvLayer {
  std::string name;
  int x_buf, x_coff, cin;  // cin = padded channels consumed
  int H, W, Ho, Wo, KH, KW, sh, sw, ph, pw;
  void* w = nullptr;
  float* bias = nullptr;
  float* slope = nullptr;
  int4* ktab = nullptr;
  int K, Kpad, cout, cout_pad, ncls = 1;
  int nseg = 0;
  struct { int c0, c1, buf, coff; } seg[4];
  int res_buf = -1, res_coff = 0;
  int act = ACT_NONE, out_f32 = 0;
  int cfg = -1;  // tile configuration chosen by Encoder::autotune (-1 = launcher heuristic)
  double macs_alg = 0, macs_exec = 0;  // per image
};

struct Op {
  enum Kind { PACK, CONV, MAXPOOL, AVGPOOL, L2NORM, COPYOUT, MAXPOOLC, STEM1, DWCONV, UPADD, RSTEM, DWPW } kind;
  int a = 0, b = 0, c = 0, d = 0, e = 0;  // meaning per kind (see engine.cpp)
};

struct DwLayer {   // depthwise 3x3 pad 1 + folded BN + LeakyReLU (fp32 plans: RetinaFace's MobileNetV1)
  int x_buf, o_buf, C, stride;
  float *w = nullptr, *bias = nullptr;   // [9][C], [C]
  float slope = 0.f;
};

struct DwPwLayer {   // conv_dw in one kernel (launch_dwpw): depthwise 3x3 + BN + leaky, pointwise 1x1 + BN + leaky
  int x_buf, o_buf, cin, cout, stride;
  float *dw = nullptr, *dbias = nullptr, *pw = nullptr, *pbias = nullptr;
  float slope = 0.f;
  std::string name;
};

struct Group { int first, last, chunk; };

// A run of plan ops that fused kernels replace when the compute dtype allows it (16-bit operands): the 40 convolutions
// of repeat_2 become one launch of block17_trunk_kernel (trunk17.hip), the 25 of repeat_1 five launches of
// block35_kernel (block35.hip).
struct FusedStack {
  int kind = 17;               // 17: persistent Block17 stack (trunk17.hip); 35: one fused launch per Block35 (block35.hip);
                               // 2: conv2d_2a + conv2d_2b + maxpool_3a in one launch (stem_mid.hip)
  int first = 0, last = 0;     // op range
  int in_buf = -1, out_buf = -1;
  int nblocks = 0;
  int conv0 = 0;               // index of the first of the 4*nblocks convolutions (reduce, 1x7, 7x1, up per block)
  void* wstream = nullptr;
  float* bias = nullptr;
  void* wtail = nullptr;       // kind 35 with ext: block35_tail_repack image of mixed_6a.branch1.0
  bool active = false;
  bool stack = false;          // kind 35: the five blocks in ONE launch, x resident in registers (trunk35.hip; bf16 / f16)
  double macs_alg = 0;         // per image
  // kind 2: conv2d_3b (the op after the pool) folded into the stem kernel; kind 35 (stack): mixed_6a.branch1.0 (the op
  // after the last block) computed from the register-resident output: ops [first, ext_last) become one launch
  int ext_last = 0, ext_conv = -1, ext_out_buf = -1;
  bool ext = false;
};  // ops [first,last) run per `chunk` images (L3 residency)

struct Tap { int buf, coff, C; };

struct Encoder : HandleBase {
  ~Encoder() override;  // side streams, fork/join and context events (device buffers: HandleBase)
  int arch, dtype, max_batch, in_size;
  std::vector<Buf> bufs;
  std::vector<ConvLayer> convs;
  std::vector<DwLayer> dws;
  std::vector<DwPwLayer> dwpws;
  float *rstem_wa = nullptr, *rstem_bias = nullptr;   // RetinaFace stem (Op::RSTEM): MFMA lane table [7][64], bias [8]
  std::vector<Op> ops;
  std::vector<Group> groups;
  std::vector<FusedStack> fused;
  int prepare_fused();  // build the weight streams of the fused stacks (finalize)
  std::unordered_map<std::string, Tap> taps;
  bool buf_materialised(int buf) const;  // false: only ops that an ACTIVE fused stack replaces would write it
  float* emb_raw = nullptr;  // (max_batch,512) fp32 before the final normalisation
  double macs_alg = 0, macs_exec = 0;

  int add_buf(int H, int W, int C);
  int finalize();  // allocate buffers, autotune tile configurations
  int autotune();
  ConvArgs conv_args(const ConvLayer& L, int n0, int nn) const;
  int run(const void* x, int n, int x_dtype, float* out, hipStream_t s, std::string* report = nullptr);
  int run_range(const void* x, int i0, int i1, int x_dtype, float* out, hipStream_t s, std::string* report);
  float* stem_wt = nullptr;  // IRv1: fp32 folded conv2d_1a weights + biases for the direct stem kernel (Op::STEM1)
  // activation-buffer contexts: consecutive vnf_embed calls rotate over n_ctx private buffer sets, so calls issued
  // on DIFFERENT streams may overlap on the GPU (the latency-bound tail of one batch under the throughput-bound
  // stem of the next); a set is re-used only after the event of its previous use.  Extra sets are allocated lazily.
  int n_ctx = 1, next_ctx = 0;
  std::vector<std::vector<char*>> ctx_bufs;
  std::vector<float*> ctx_emb;
  std::vector<hipEvent_t> ctx_ev;
  int select_ctx(hipStream_t s, int* used);
  int max_streams = 4;  // cap on run()'s batch split (1: never fork side streams)
  bool tune_dirty = true;  // tiles are (re)picked lazily at the next run(): create, set_streams and set_contexts only mark
  int tune_lanes = 1;   // concurrent copies the autotuner times each candidate as (set with the context count)
  int tune_batch = 0;   // batch size the autotuner times at (0: the part size run() uses at max_batch)
  hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t join_ev[4] = {nullptr, nullptr, nullptr, nullptr}, fork_ev = nullptr;
};

int build_irv1(Encoder& e, WeightMap& wm);
int build_ir100(Encoder& e, WeightMap& wm);
// MTCNN R-Net / O-Net as exact-f32 MFMA plans over NHWC4 candidate crops (input buffer 0 is written by the
// crop kernel; the last buffer holds the head outputs: 8 floats [a0,a1,reg0..3,-,-] / 16 floats [a0,a1,reg0..3,lm0..9])
// RetinaFace (mobilenet0.25) on the exact-f32 core for an H x W input: buffer 0 = NHWC4 mean-subtracted input (written by
// the caller), head_bufs[l] = (Hl, Wl, 32) fp32 [cls 4 | bbox 8 | landmark 20] of pyramid level l
int build_retina_mnet(Encoder& e, WeightMap& wm, int H, int W, int head_bufs[3]);
int build_rnet(Encoder& e, WeightMap& wm, bool front = false, bool mid = false);
int build_onet(Encoder& e, WeightMap& wm, bool front = false, bool mid = false);

}  // namespace vnf
