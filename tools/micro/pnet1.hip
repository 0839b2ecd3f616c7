// Microbenchmark (GPU box): what bounds a conv1(3->10,3x3)+PReLU+maxpool(2,2) kernel of the P-Net shape -- variants
// with the global loads or the arithmetic removed, one level of Hs x Ws per image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));
struct W { const float *w1, *b1, *a1; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(const float* __restrict__ lvl, int Hs, int Ws, int tot_px, W w, float* __restrict__ p1) {
  const int Hc = Hs - 2, Wc = Ws - 2, Hp = (Hc + 1) / 2, Wp = (Wc + 1) / 2, tot_p1 = Hp * Wp;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int img = blockIdx.y;
  const int py = idx / Wp, px = idx - py * Wp;
  float in[3][4][4];
  const float* src = lvl + ((size_t)img * 3) * tot_px;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int yy = min(2 * py + dy, Hs - 1), xx = min(2 * px + dx, Ws - 1);
        if (MODE == 1 || MODE == 5) in[c][dy][dx] = (float)(yy * 3 + xx + c) * 1e-3f;
        else in[c][dy][dx] = src[(size_t)c * tot_px + yy * Ws + xx];
      }
  float best[10];
  __shared__ float sw[272];
  if (MODE == 4 || MODE == 9) {
    for (int i = threadIdx.x; i < 270; i += blockDim.x) sw[i] = w.w1[i];
    __syncthreads();
  }
  if (MODE == 2) {
    float sacc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) sacc += in[c][dy][dx];
#pragma unroll
    for (int co = 0; co < 10; ++co) best[co] = sacc + co;
  } else if (MODE >= 7) {
    constexpr int D = MODE == 7 ? 1 : (MODE == 8 ? 3 : 2);
    const float* wsrc = MODE == 9 ? sw : w.w1;
    float2_t acc2[4][5];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc2[q][j] = float2_t{w.b1[2 * j], w.b1[2 * j + 1]};
    float2_t wv[D + 1][5];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int j = 0; j < 5; ++j) wv[d][j] = float2_t{wsrc[d * 10 + 2 * j], wsrc[d * 10 + 2 * j + 1]};
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int c = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
      if (tap + D < 27) {
#pragma unroll
        for (int j = 0; j < 5; ++j) wv[(tap + D) % (D + 1)][j] = float2_t{wsrc[(tap + D) * 10 + 2 * j], wsrc[(tap + D) * 10 + 2 * j + 1]};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = in[c][(q >> 1) + kh][(q & 1) + kw];
        const float2_t v2 = {v, v};
#pragma unroll
        for (int j = 0; j < 5; ++j) acc2[q][j] = __builtin_elementwise_fma(v2, wv[tap % (D + 1)][j], acc2[q][j]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(acc2[q][j]));
      __builtin_amdgcn_sched_barrier(0);
    }
    float a1r[10];
#pragma unroll
    for (int co = 0; co < 10; ++co) a1r[co] = w.a1[co];
#pragma unroll
    for (int co = 0; co < 10; ++co) best[co] = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = 2 * py + (q >> 1) < Hc && 2 * px + (q & 1) < Wc;
#pragma unroll
      for (int co = 0; co < 10; ++co) {
        const float av = acc2[q][co >> 1][co & 1];
        const float a = av > 0.f ? av : av * a1r[co];
        best[co] = ok ? fmaxf(best[co], a) : best[co];
      }
    }
  } else {
    float2_t acc2[4][5];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc2[q][j] = float2_t{w.b1[2 * j], w.b1[2 * j + 1]};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float* ww = (MODE == 4 ? sw : w.w1) + (MODE == 5 || MODE == 6 ? 0 : ((c * 3 + kh) * 3 + kw) * 10);
          float2_t wv[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) wv[j] = float2_t{ww[2 * j], ww[2 * j + 1]};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = in[c][(q >> 1) + kh][(q & 1) + kw];
            const float2_t v2 = {v, v};
#pragma unroll
            for (int j = 0; j < 5; ++j) acc2[q][j] = __builtin_elementwise_fma(v2, wv[j], acc2[q][j]);
          }
        }
#pragma unroll
    for (int co = 0; co < 10; ++co) best[co] = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = 2 * py + (q >> 1) < Hc && 2 * px + (q & 1) < Wc;
#pragma unroll
      for (int co = 0; co < 10; ++co) {
        const float av = acc2[q][co >> 1][co & 1];
        const float a = av > 0.f ? av : av * w.a1[co];
        best[co] = ok ? fmaxf(best[co], a) : best[co];
      }
    }
  }
  if (idx >= tot_p1) return;
  float* o = p1 + ((size_t)img * 10) * tot_p1 + idx;
  if (MODE == 3) { o[0] = best[0] + best[1] + best[2] + best[3] + best[4] + best[5] + best[6] + best[7] + best[8] + best[9]; return; }
#pragma unroll
  for (int co = 0; co < 10; ++co) o[(size_t)co * tot_p1] = best[co];
}

int main() {
  const int Hs = 520, Ws = 463, B = 16, tot_px = Hs * Ws;   // ~240k px per image, like the 9 levels of a 1080p frame
  const int Hp = (Hs - 1) / 2, Wp = (Ws - 1) / 2, tot_p1 = Hp * Wp;
  float *lvl, *p1, *wd;
  CK(hipMalloc(&lvl, (size_t)B * 3 * tot_px * 4));
  CK(hipMalloc(&p1, (size_t)B * 10 * tot_p1 * 4));
  CK(hipMalloc(&wd, 4096));
  std::vector<float> h(1024, 0.01f);
  CK(hipMemcpy(wd, h.data(), 4096, hipMemcpyHostToDevice));
  CK(hipMemset(lvl, 0, (size_t)B * 3 * tot_px * 4));
  W w{wd, wd + 300, wd + 320};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int mode) {
    dim3 g((tot_p1 + 255) / 256, B), b(256);
    for (int it = 0; it < 13; ++it) {
      if (it == 3) (void)hipEventRecord(e0, 0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 8: hipLaunchKernelGGL(k<8>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
        case 9: hipLaunchKernelGGL(k<9>, g, b, 0, 0, lvl, Hs, Ws, tot_px, w, p1); break;
      }
    }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const char* extra[3] = {"explicit taps, scalar weights 1 ahead", "explicit taps, scalar weights 3 ahead", "explicit taps, LDS weights 2 ahead"};
    if (mode >= 7) { printf("mode %d (%s): %.1f us per launch\n", mode, extra[mode - 7], ms * 100.f); return; }
    printf("mode %d (%s): %.1f us per launch\n", mode, mode == 0 ? "full" : mode == 1 ? "no loads" : mode == 2 ? "no FMAs" : mode == 3 ? "one store" : mode == 4 ? "weights in LDS" : mode == 5 ? "no loads, one tap of weights" : "loads, one tap of weights", ms * 100.f);
  };
  for (int m = 0; m < 10; ++m) if (m != 4 && m != 3) run(m);
  return 0;
}
