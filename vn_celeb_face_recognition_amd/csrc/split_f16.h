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

}  // namespace vnf
