// Microbenchmark (GPU box): issue rate of v_fma_f32 / v_pk_fma_f32 with VGPR and SGPR operands, 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ void k(float* out, const float* __restrict__ wv, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float2_t acc[8];
  float2_t x = {(float)t * 1e-6f, (float)t * 2e-6f};
  for (int j = 0; j < 8; ++j) acc[j] = float2_t{(float)j, (float)j + 0.5f};
  const float s0 = wv[0], s1 = wv[1];           // uniform: SGPRs
  float2_t sv = {s0, s1};
  float2_t vv = {wv[t & 63], wv[(t & 63) + 1]};  // per-lane: VGPRs
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) acc[j] = __builtin_elementwise_fma(x, vv, acc[j]);                       // pk, VGPR operands
        if (MODE == 1) acc[j] = __builtin_elementwise_fma(x, sv, acc[j]);                       // pk, SGPR pair operand
        if (MODE == 2) { acc[j].x = fmaf(x.x, vv.x, acc[j].x); acc[j].y = fmaf(x.y, vv.y, acc[j].y); }   // scalar fma, VGPR
        if (MODE == 3) { acc[j].x = fmaf(x.x, s0, acc[j].x); acc[j].y = fmaf(x.y, s1, acc[j].y); }       // scalar fma, SGPR
        if (MODE == 4) acc[j] = __builtin_elementwise_fma(float2_t{x.x, x.x}, sv, acc[j]);      // pk, splat + SGPR (the P-Net form)
      }
  }
  float2_t r = acc[0];
  for (int j = 1; j < 8; ++j) r += acc[j];
  out[t] = r.x + r.y;
}
int main() {
  float *o, *w; CK(hipMalloc(&o, 1 << 26)); CK(hipMalloc(&w, 4096)); CK(hipMemset(w, 0, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 16, iters = 64;     // 16 workgroups of 256 per CU
  const char* names[5] = {"v_pk_fma_f32 VGPR", "v_pk_fma_f32 SGPR pair", "v_fma_f32 VGPR", "v_fma_f32 SGPR", "v_pk_fma_f32 splat x SGPR pair"};
  for (int m = 0; m < 5; ++m) {
    for (int it = 0; it < 6; ++it) {
      if (it == 1) (void)hipEventRecord(e0, 0);
      switch (m) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, o, w, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, o, w, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, o, w, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, o, w, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, o, w, iters); break;
      }
    }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * 256 * iters * 64 * 2 * 5;   // FMAs in the 5 timed launches
    printf("%-34s %.1f us/launch  %.1f TFLOP/s\n", names[m], ms * 200.f, 2 * fma / (ms * 1e-3) / 1e12);
  }
  return 0;
}
