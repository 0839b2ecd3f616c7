// Microbenchmark (GPU box): block17_trunk_split_kernel alone on random data, for A/B of its build variants
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DT17S_RING=8 -DT17S_P1=0 tools/micro/trunk17s_bench.hip -o t.bin
#include "../../vn_celeb_face_recognition_amd/csrc/trunk17s.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace vnf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const int n = 256, nb = 10;
  const size_t xbytes = (size_t)n * 64 * 896 * 4, wbytes = trunk17s_stream_bytes(nb);
  std::vector<_Float16> hx(xbytes / 2), hw(wbytes / 2);
  std::mt19937 g(1);
  std::normal_distribution<float> d(0.f, 1.f);
  for (auto& v : hx) v = (_Float16)(d(g) * 0.5f);
  for (auto& v : hw) v = (_Float16)(d(g) * 0.02f);
  void *x, *y, *w; float* bias;
  CK(hipMalloc(&x, xbytes)); CK(hipMalloc(&y, xbytes)); CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&bias, nb * T17_BIAS * 4));
  CK(hipMemcpy(x, hx.data(), xbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, nb * T17_BIAS * 4));
  Trunk17Args a; a.x = x; a.y = y; a.ldx = a.ldy = 896; a.n = n; a.nblocks = nb; a.wstream = w; a.bias = bias;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(launch_trunk17s(a, 0));
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) CK(launch_trunk17s(a, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 10 < best) best = ms / 10;
  }
  printf("RING %d P1 %d: %.4f ms per launch (%d images, %d blocks), %.1f TFLOP/s algorithmic\n", T17S_RING, T17S_P1, best, n, nb, 225.5 / best);
  return 0;
}
