// Microbenchmark (GPU box): cost of a chain of dependent small kernels in one stream, plain launches vs hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  float* d; CK(hipMalloc(&d, 1 << 24));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 130, REP = 50;
  for (int blocks : {1, 256, 2048}) {
    for (int r = 0; r < 3; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, d, blocks * 256);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, d, blocks * 256);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    double cpu = std::chrono::duration<double, std::micro>(t1 - t0).count() / (REP * N);
    double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / (REP * N);
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, d, blocks * 256);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto g0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
    auto g1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto g2 = std::chrono::steady_clock::now();
    double gcpu = std::chrono::duration<double, std::micro>(g1 - g0).count() / (REP * N);
    double gtot = std::chrono::duration<double, std::micro>(g2 - g0).count() / (REP * N);
    printf("blocks %4d: plain cpu %.2f us/launch, end-to-end %.2f us/kernel | graph cpu %.2f, end-to-end %.2f us/kernel\n", blocks, cpu, tot, gcpu, gtot);
  }
  return 0;
}
