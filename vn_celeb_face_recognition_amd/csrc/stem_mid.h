// Launch interface of the fused conv2d_2a -> conv2d_2b -> maxpool_3a kernel (stem_mid.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vnf {

constexpr int SM_WFRAG_BYTES = 54 * 1024;   // 6 channel tiles x 9 taps of 1-KiB MFMA A-fragments
constexpr int SM_BIAS = 96;                 // fp32: 32 (conv2d_2a) | 64 (conv2d_2b)

struct StemMidArgs {
  const void* x;       // (n, 79, 79, ldx) 16-bit NHWC conv2d_1a output (32 channels used)
  void* y;             // (n, 38, 38, ldy) pooled output, 64 channels
  int ldx, ldy, n;
  const void* wfrag;   // stem_mid_repack output
  const float* bias;   // [SM_BIAS]
  // optional: conv2d_3b (1x1, 64 -> 80, folded BN, ReLU) applied to every pooled row before it leaves the kernel; y is
  // then the (n, 38, 38, ldy) conv2d_3b output and the pooled tensor never reaches memory
  const void* w3b;     // packed engine weights [>= 80 rows][k3b_pad], 16-bit; nullptr: no fusion
  const float* b3b;    // [80]
  int k3b_pad;
  long long* dbg = nullptr;   // in-kernel timer buffer of the instrumented launch (tools), else null
};

struct StemMidPack {    // packed engine weights [rows][kpad], k = (kh, kw, c): conv2d_2a (32 x 288), conv2d_2b (64 x 288)
  const void* w[2];
  int kpad[2];
};

// planar split-f16 twin (stem_mids.hip): x / y are F16P tensors, weights F16P-packed; 108 fragments; conv2d_3b always fused
constexpr int SMS_WFRAG_BYTES = 108 * 1024;
hipError_t stem_mids_repack(const StemMidPack& p, void* out, hipStream_t s);
hipError_t launch_stem_mids(const StemMidArgs& a, hipStream_t s);

hipError_t stem_mid_repack(const StemMidPack& p, void* out, hipStream_t s);
hipError_t launch_stem_mid(const StemMidArgs& a, int dtype, hipStream_t s);

}  // namespace vnf
