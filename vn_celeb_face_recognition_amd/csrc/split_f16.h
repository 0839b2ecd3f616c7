// Split-f16 storage type of the "f16x2" compute path: one fp32 value kept as an (hi, lo) pair of IEEE halves with
// hi = rne_f16(v), lo = rne_f16(v - hi), i.e. ~22 significant bits in 32 bits of storage.  Products of two such
// values expand to hi*hi' + hi*lo' + lo*hi' (+ lo*lo'), which the 16-bit MFMA (v_mfma_f32_16x16x32_f16, fp32
// accumulate) evaluates exactly per product: fp32-class accuracy (the <=1e-4 embedding gate) at the 16-bit matrix
// rate instead of the 16x slower v_mfma_f32_16x16x4_f32.  Range is that of f16 (|v| < 65504).
#pragma once

namespace vnf {

struct sf16 {
  _Float16 hi, lo;
  sf16() = default;
  __host__ __device__ explicit sf16(float v) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
  }
  __host__ __device__ explicit operator float() const { return (float)hi + (float)lo; }
};
static_assert(sizeof(sf16) == 4, "sf16 is one dword");

// Planar split-f16 ("F16P", the encoders' f16x2 storage): the same (hi, lo) halves, but laid out per group of 8
// consecutive channels (or k values) as [hi0 .. hi7][lo0 .. lo7] -- 32 bytes, the hi plane and the lo plane each one
// 16-byte MFMA operand chunk.  The product of two such tensors is then three plain MFMAs on chunks as they are loaded,
//     W_hi . X_hi  +  W_hi . X_lo  +  W_lo . X_hi          (lo . lo' is below 2^-22 of the product and is dropped)
// with no lane or register shuffling at all (the interleaved sf16 form needs four MFMA-equivalents and a rotation of
// every activation dword).  pf16 is a TAG for templates: sizeof == 4 bytes per value so that element offsets,
// strides and buffer sizes are those of any 4-byte dtype; single elements are never addressed through it -- tensors
// are read and written in 8-channel units (load8 / store8), channel counts and offsets are multiples of 8.
struct pf16 {
  unsigned bits;
};
static_assert(sizeof(pf16) == 4, "pf16 stands for 4 bytes per value");

}  // namespace vnf
