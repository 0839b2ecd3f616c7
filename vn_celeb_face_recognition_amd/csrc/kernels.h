// Internal launch interface between the engine (engine.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vnf {

// F16X2: split-f16 (hi, lo) pairs, split_f16.h -- fp32-class accuracy on the 16-bit MFMA (ids follow include/vnface.h)
// F16P: planar split-f16 (8-channel units [8 hi][8 lo], three MFMAs per product): what the ENCODERS run when the caller
// asks for VNF_F16X2; F16X2 itself (interleaved pairs) stays the storage of the R-Net / O-Net / RetinaFace plans
enum DType { F32 = 0, BF16 = 1, F16 = 2, F16X2 = 5, F16P = 6 };
inline int dtype_size(int dt) { return (dt == F32 || dt == F16X2 || dt == F16P) ? 4 : 2; }
inline int dtype_chan_align(int dt) { return dt == F16P ? 8 : 16 / dtype_size(dt); }   // channel granularity of slices / gathers

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2 };

struct ConvSeg {   // output columns [c0,c1) of the GEMM go to ptr (already channel-offset), pixel stride ld
  int c0, c1;
  void* ptr;
  int ld;
};

// One implicit-GEMM convolution over NHWC activations.
//   out[m][co] = act( sum_k x[pix(m) + tap(k)] * w[co][k] + bias[cls(m)][co] + res[m][co] )
// m = (n*Ho + ho)*Wo + wo; k = (kh*KW + kw)*Cin + c.
struct ConvArgs {
  int dtype;            // compute/storage dtype of x, w, res, out (unless out_f32)
  const void* x;        // channel-offset base of the input slice
  int ldx;              // pixel stride of x in elements
  int H, W, Cin;        // Cin = channels consumed (multiple of the 16-byte chunk)
  int Ho, Wo;
  int KH, KW, sh, sw, ph, pw;
  const void* w;        // packed [Cout_pad][Kpad], k-order (kh,kw,c)
  int K, Kpad;
  const float* bias;    // [ncls][Cout_pad]
  int ncls;             // 1, or 9 for the border-class bias of a folded pre-conv BatchNorm
  int cout_pad;         // row stride of bias / slope tables
  const int4* ktab;     // per 16-byte k-chunk: {element offset, dh, dw, valid}
  int M, Cout;
  int nseg;
  ConvSeg seg[4];
  const void* res;      // optional residual [M][ldres] (channel-offset base)
  int ldres;
  int act;
  const float* slope;   // PReLU slopes [Cout_pad]
  int out_f32;          // store fp32 instead of dtype
  int cfg;              // tile configuration id (conv_cfg_ok), -1 = heuristic
};

hipError_t launch_conv(const ConvArgs& a, hipStream_t s);
int conv_num_cfgs();
bool conv_cfg_ok(const ConvArgs& a, int cfg);  // is tile configuration `cfg` usable for this convolution

// NCHW (n,3,S,S) of x_dtype -> NHWC8 of dtype (channels 3..7 zero)
hipError_t launch_pack_input(const void* x, int x_dtype, void* out, int dtype, int n, int hw, hipStream_t s);

// IRv1 stem: NCHW (n,3,160,160) of x_dtype -> conv2d_1a (3x3 s2, folded BN, ReLU) NHWC (n,79,79,32) of dtype;
// wt = fp32 [27][32] folded weights (k = (c*3+kh)*3+kw) followed by 32 biases
hipError_t launch_stem_conv1a(const void* x, int x_dtype, void* y, int ldy, int dtype, int n, const float* wt, hipStream_t s);

// 3x3 stride-2 max pool, floor mode, NHWC slice -> NHWC slice
hipError_t launch_maxpool3s2(const void* x, int ldx, void* y, int ldy, int dtype, int n, int H, int W, int C,
                             hipStream_t s);

// k x k stride-2 max pool with ceil_mode=True (MTCNN R/O-Net), NHWC slice -> NHWC slice
hipError_t launch_maxpool_ceil(const void* x, int ldx, void* y, int ldy, int dtype, int n, int H, int W, int C, int k,
                               hipStream_t s);

// global average pool NHWC (n,HW,C) -> (n,C)
hipError_t launch_avgpool(const void* x, int ldx, void* y, int dtype, int n, int HW, int C, hipStream_t s);

// rows of fp32 (n,C): y = x / max(||x||_2, 1e-12)
hipError_t launch_l2norm(const float* x, float* y, int n, int C, hipStream_t s);

// rows of fp32 logits (n, ld >= C): log_softmax over the first C columns, argmax (first
// occurrence) and exp(logp[argmax])
hipError_t launch_logsoftmax_argmax(const float* logits, int ld, int C, int n, float* logp, int32_t* amax, float* prob,
                                    hipStream_t s);

// depthwise 3x3 convolution, padding 1, stride 1 or 2, fp32 NHWC (C % 4 == 0): y = leaky(sum_t x[tap t] * w[t][c] + bias[c])
// (retina_face_utils/components.py:30-40 conv_dw, first half; BatchNorm folded into w / bias)
// (split: the tensors hold split-f16 pairs in their 32-bit elements -- F16X2 plans -- instead of fp32; same for the
// three launchers below)
hipError_t launch_dwconv3x3(const float* x, float* y, int n, int H, int W, int C, int stride, const float* w9c,
                            const float* bias, float slope, bool split, hipStream_t s);

// RetinaFace stem: u8 RGB frames (n,H,W,3) -> (x - (104,117,123)) -> 3x3 stride-2 pad-1 conv 3->8 + folded BN + LeakyReLU,
// NHWC8 fp32 out (retina_face.py:158-164 + components.py:103 conv_bn(3, 8, 2)); wa = [7][64] MFMA lane table
hipError_t launch_retina_stem(const uint8_t* frames, int n, int H, int W, const float* wa, const float* bias, float slope, float* y,
                              bool split, hipStream_t s);
// conv_dw in one pass (components.py:30-40): depthwise 3x3 pad 1 (stride 1|2) + BN + LeakyReLU + pointwise 1x1 + BN +
// LeakyReLU, fp32 NHWC; (cin, cout) in {(8,16), (16,32), (32,32), (32,64)}
bool dwpw_supported(int cin, int cout);
hipError_t launch_dwpw(const float* x, float* y, int n, int H, int W, int cin, int cout, int stride, const float* dw, const float* dbias,
                       float dslope, const float* pw, const float* pbias, float pslope, bool split, hipStream_t s);

// y[n][h][w][c] += x[n][floor(h * Hs/H)][floor(w * Ws/W)][c]: F.interpolate(mode="nearest") + add
// (retina_face_utils/components.py:88-94), fp32 NHWC
hipError_t launch_upsample_add(const float* x, int Hs, int Ws, float* y, int H, int W, int C, int n, bool split, hipStream_t s);

// NHWC slice (dtype) -> NCHW fp32 (for taps / debugging)
hipError_t launch_nhwc_to_nchw_f32(const void* x, int ldx, int dtype, float* y, int n, int HW, int C, hipStream_t s);

}  // namespace vnf
