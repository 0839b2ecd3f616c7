// Launch interface of the persistent Block17 stack kernel (trunk17.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vnf {

constexpr int T17_FRAGS = 168;   // 1-KiB weight fragments per wave per block: 56 reduce + 28 (1x7) + 28 (7x1) + 56 up
constexpr int T17_BIAS = 1408;   // fp32 biases per block: 256 reduce | 128 (1x7) | 128 (7x1) | 896 up
constexpr int T17_MAX_BLOCKS = 10;

struct Trunk17Args {
  const void* x;        // (n, 64, ldx) 16-bit NHWC input (first 896 channels of each pixel row)
  void* y;              // (n, 64, ldy) output
  int ldx, ldy, n, nblocks;
  const void* wstream;  // trunk17_repack output
  const float* bias;    // [nblocks][T17_BIAS]
};

// packed per-convolution weights [rows][kpad] of every block (engine layout, k = (kh, kw, c)), device pointers
struct Trunk17Pack {
  const void* w[T17_MAX_BLOCKS][4];   // reduce (256 rows, K 896), 1x7 (128, 896), 7x1 (128, 896), up (896, 256)
  int kpad[4];
  int nblocks;
};

// planar split-f16 twin (trunk17s.hip): x / y are F16P tensors (4 bytes per value), the pack's weights F16P-packed
size_t trunk17s_stream_bytes(int nblocks);
hipError_t trunk17s_repack(const Trunk17Pack& p, void* out, hipStream_t s);
hipError_t launch_trunk17s(const Trunk17Args& a, hipStream_t s);

size_t trunk17_stream_bytes(int nblocks);
hipError_t trunk17_repack(const Trunk17Pack& p, void* out, hipStream_t s);
hipError_t launch_trunk17(const Trunk17Args& a, int dtype, hipStream_t s);

}  // namespace vnf
