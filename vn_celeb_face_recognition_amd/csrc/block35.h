// Launch interface of the fused Block35 kernel (block35.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vnf {

// weight image of one block (bytes): 48 + 3*18 + 48 fragments of 1 KiB, MFMA A-fragment order
constexpr int B35_W1 = 0, B35_W2 = 48 * 1024, B35_W3 = 66 * 1024, B35_W4 = 84 * 1024, B35_W5 = 102 * 1024;
constexpr int B35_FRAGS = 150;
constexpr int B35_BIASOFF = B35_FRAGS * 1024;          // 2 KiB: the block's fp32 biases follow the weight fragments
constexpr int B35_WIMG_BYTES = (B35_FRAGS + 2) * 1024;
constexpr int B35_BIAS = 448;  // fp32 per block: 96 reduce | 32 branch1.1 | 32 branch2.1 | 32 branch2.2 | 256 up

struct Block35Args {
  const void* x;      // (n, 289, ldx) 16-bit NHWC block input (residual source)
  void* y;            // (n, 289, ldy) block output
  int ldx, ldy, n;
  const void* wimg;   // block35_repack output of this block (weights + biases)
  const void* zero;   // >= 16 zero bytes (padding rows of the x tiles)
};

struct Block35Pack {   // packed engine weights [rows][kpad] (k = (kh, kw, c)) of the block's five convolutions
  const void* w[5];    // reduce (96 x 256), branch1.1, branch2.1, branch2.2 (32 x 288 each), up (256 x 96)
  int kpad[5];
  const float* bias;   // device [B35_BIAS]: the five convolutions' biases in that order
};

// planar split-f16 twin (block35s.hip): x / y are F16P tensors, the pack's weights F16P-packed; 300 fragments + 2 KiB biases
constexpr int B35S_WIMG_BYTES = (300 + 2) * 1024;
hipError_t block35s_repack(const Block35Pack& p, void* out, hipStream_t s);
hipError_t launch_block35s(const Block35Args& a, hipStream_t s);

hipError_t block35_repack(const Block35Pack& p, void* out, hipStream_t s);
hipError_t launch_block35(const Block35Args& a, int dtype, hipStream_t s);

// the whole repeat_1 stack in one launch, x resident in registers (trunk35.hip; bf16 / f16 plans)
struct Block35StackArgs {
  const void* x;      // (n, 289, ldx) input of the first block
  void* y;            // (n, 289, ldy) output of the last block (may alias x)
  int ldx, ldy, n, nblocks;
  const void* wimg;   // nblocks block35_repack images, B35_WIMG_BYTES apart
  // optional: mixed_6a.branch1.0 (1x1, 256 -> 192, folded BN, ReLU: inception_resnet_v1.py:137) on the stack's output while it
  // is still in registers: wtail = block35_tail_repack image, ytail = (n, 289, ldyt) output; nullptr: no fusion
  const void* wtail = nullptr;
  void* ytail = nullptr;
  int ldyt = 0;
  long long* dbg = nullptr;  // in-kernel stamp buffer of the instrumented launch (tools), else null
};
hipError_t launch_block35_stack(const Block35StackArgs& a, int dtype, hipStream_t s);
constexpr int B35_TAIL_BYTES = 97 * 1024;   // 96 weight fragments (k-step x 12 channel tiles) + 1 KiB of fp32 biases
hipError_t block35_tail_repack(const void* w, int kpad, const float* bias, void* out, hipStream_t s);
const char* conv_zero_page();  // conv_igemm.hip: per-device page of zero bytes

}  // namespace vnf
