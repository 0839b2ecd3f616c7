// Training of the MLP identity classifier on precomputed embeddings: the arithmetic of
// /root/reference/trainer/classification_trainer.py:9-40 (one optimisation step: forward, NLL loss, backward,
// optimizer.step) for models/mlp_model.py:4-15 with torch.optim.Adam (cfg/train_cfg_emb_classify.json: lr 1e-4,
// weight_decay 1e-4, betas (0.9, 0.999), eps 1e-8), fp32 end to end on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32 == an fp32 fma chain).
//
//   h  = relu(x W1^T + b1) * mask          mask: the caller's dropout draw, 0 or 1/(1-p) per element (F.dropout)
//   z  = h W2^T + b2 ;  logp = log_softmax(z) ;  loss = -mean_b logp[b, t_b]              (losses/__init__.py: NLLLoss)
//   dz = (softmax(z) - onehot(t)) / B ;  dW2 = dz^T h ; db2 = sum_b dz ; dh = (dz W2) * mask * (pre > 0)
//   dW1 = dh^T x ; db1 = sum_b dh ;  Adam with L2 weight decay folded into the gradient (torch.optim.Adam)
//
// Every product is one NT GEMM C[M][N] = sum_k A[m][k] B[n][k] (both operands k-contiguous); the three products that
// are not in that form get their operand transposed by a small tile-transpose kernel first.  Sizes are tiny (batch 64:
// ~1.2 GFLOP per step), so the kernels are simple 64x64 LDS-tiled MFMA loops with bounds checks, not the inference core.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

namespace vnf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// C[M][N] (ldc) = A[M][K] (lda) x B[N][K] (ldb)^T (+ bias[n]) (ReLU) (* mask[m][n]); pre (optional) receives the value
// before ReLU/mask.  64x64 tile per workgroup of 4 waves (2x2 tiles of 32x32), K step 16.
__global__ void __launch_bounds__(256) gemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                      float* __restrict__ C, int ldc, int M, int N, int K,
                                                      const float* __restrict__ bias, int relu, const float* __restrict__ mask,
                                                      float* __restrict__ pre) {
  __shared__ float sA[64][17], sB[64][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // 64 rows x 16 k = 1024 elements per operand, 4 per thread
      const int idx = tid + 256 * q, r = idx >> 4, c = idx & 15;
      sA[r][c] = (m0 + r < M && k0 + c < K) ? A[(size_t)(m0 + r) * lda + k0 + c] : 0.f;
      sB[r][c] = (n0 + r < N && k0 + c < K) ? B[(size_t)(n0 + r) * ldb + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sA[wm + 16 * i + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = sB[wn + 16 * j + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[row = 4*(lane>>4) + r][col = lane & 15]: row from A (m), column from B (n)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + r, n = n0 + wn + 16 * j + (lane & 15);
        if (m < M && n < N) {
          float v = acc[i][j][r] + (bias ? bias[n] : 0.f);
          if (pre) pre[(size_t)m * ldc + n] = v;
          if (relu) v = fmaxf(v, 0.f);
          if (mask) v *= mask[(size_t)m * ldc + n];
          C[(size_t)m * ldc + n] = v;
        }
      }
}

// out[c][r] = in[r][c]
__global__ void transpose_kernel(const float* __restrict__ in, int R, int Cc, float* __restrict__ out) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    t[i][threadIdx.x] = (r < R && c < Cc) ? in[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < Cc && r < R) out[(size_t)c * R + r] = t[threadIdx.x][i];
  }
}

// one wave per row: log_softmax, NLL term, argmax match, dz = (softmax - onehot) / B
__global__ void softmax_nll_kernel(const float* __restrict__ z, int C, int Bn, const int64_t* __restrict__ target,
                                   float* __restrict__ dz, float* __restrict__ loss_rows, int* __restrict__ hit_rows, float inv_b) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Bn) return;
  const float* x = z + (size_t)row * C;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > m) { m = v; mi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o);
    const int oi = __shfl_xor(mi, o);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float ls = logf(s);
  // a label outside [0, C) (torch's NLLLoss asserts on it; the host layer raises before the launch) never indexes the
  // row: its loss term is NaN, so the step's loss shows it, and no out-of-range address is formed
  const long long tl = (long long)target[row];
  const bool tok = tl >= 0 && tl < (long long)C;
  const int t = tok ? (int)tl : -1;
  if (dz)
    for (int c = lane; c < C; c += 64) dz[(size_t)row * C + c] = (expf((x[c] - m) - ls) - (c == t ? 1.f : 0.f)) * inv_b;
  if (lane == 0) {
    loss_rows[row] = tok ? -((x[t] - m) - ls) : __builtin_nanf("");
    hit_rows[row] = mi == t ? 1 : 0;
  }
}

// loss = mean(loss_rows), hits = sum(hit_rows); one workgroup
__global__ void reduce_rows_kernel(const float* __restrict__ loss_rows, const int* __restrict__ hit_rows, int Bn,
                                   float* __restrict__ loss_out, int* __restrict__ hits_out) {
  __shared__ float sl[256];
  __shared__ int sh[256];
  float l = 0.f;
  int h = 0;
  for (int i = threadIdx.x; i < Bn; i += 256) { l += loss_rows[i]; h += hit_rows[i]; }
  sl[threadIdx.x] = l; sh[threadIdx.x] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sh[threadIdx.x] += sh[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (loss_out) *loss_out = sl[0] / (float)Bn;
    if (hits_out) *hits_out = sh[0];
  }
}

// column sums of g[Bn][N] -> out[N]
__global__ void colsum_kernel(const float* __restrict__ g, int Bn, int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int b = 0; b < Bn; ++b) s += g[(size_t)b * N + n];
  out[n] = s;
}

// dh = dh * mask * (pre > 0)
__global__ void relu_mask_grad_kernel(float* __restrict__ dh, const float* __restrict__ pre, const float* __restrict__ mask, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = pre[i] > 0.f ? dh[i] : 0.f;
  if (mask) v *= mask[i];
  dh[i] = v;
}

// torch.optim.Adam (no amsgrad, coupled weight decay), in the operation order of torch/optim/adam.py _single_tensor_adam:
//   g = g + wd*p ; m.lerp_(g, 1-b1) ; v = v*b2 + ((1-b2)*g)*g ; p += (-step_size) * (m / (sqrt(v)/bc2_sqrt + eps))
// step_size = lr / (1 - b1^t) and bc2_sqrt = sqrt(1 - b2^t) are formed on the host in double, as Python does.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float b1, float b2, float eps, float wd, float step_size, float bc2_sqrt) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float grad = g[i] + wd * p[i];
  const float mi = m[i] + (1.f - b1) * (grad - m[i]);
  const float vi = v[i] * b2 + ((1.f - b2) * grad) * grad;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] + (-step_size) * (mi / denom);
}

struct MlpTrainer : HandleBase {
  int D = 0, C = 0, H = 2048, max_batch = 0;
  long long step = 0;
  float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, wd = 0.f;
  // parameters, gradients, Adam moments: W1 [H][D], b1 [H], W2 [C][H], b2 [C]
  float *p[4] = {nullptr, nullptr, nullptr, nullptr}, *g[4] = {nullptr, nullptr, nullptr, nullptr};
  float *m[4] = {nullptr, nullptr, nullptr, nullptr}, *v[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t numel[4] = {0, 0, 0, 0};
  // activations / scratch
  float *h = nullptr, *pre = nullptr, *z = nullptr, *dz = nullptr, *dh = nullptr, *dzT = nullptr, *hT = nullptr, *dhT = nullptr,
        *xT = nullptr, *w2T = nullptr, *loss_rows = nullptr;
  int* hit_rows = nullptr;
};

static const char* kParamNames[4] = {"dense_1.weight", "dense_1.bias", "dense_2.weight", "dense_2.bias"};

static hipError_t gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, const float* bias,
                          int relu, const float* mask, float* pre, hipStream_t s) {
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, relu,
                     mask, pre);
  return hipGetLastError();
}
static hipError_t transpose(const float* in, int R, int Cc, float* out, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, s, in, R, Cc, out);
  return hipGetLastError();
}

}  // namespace vnf
using namespace vnf;

extern "C" int vnf_mlp_trainer_create(const vnf_tensor_desc* weights, int n_weights, int input_dim, int num_classes, int max_batch,
                                      float beta1, float beta2, float eps, float weight_decay, vnf_handle* out) {
  try {
    if (!out || !weights || input_dim <= 0 || num_classes <= 0 || max_batch <= 0) return fail(VNF_E_INVALID, "vnf_mlp_trainer_create: bad argument");
    *out = nullptr;
    WeightMap wm(weights, n_weights);
    MlpTrainer* t = new MlpTrainer();
    t->kind = 4;
    t->D = input_dim; t->C = num_classes; t->max_batch = max_batch;
    t->b1 = beta1; t->b2 = beta2; t->eps = eps; t->wd = weight_decay;
    (void)hipGetDevice(&t->device);
    const size_t H = t->H, D = input_dim, C = num_classes, B = max_batch;
    const size_t ne[4] = {H * D, H, C * H, C};
    for (int i = 0; i < 4; ++i) {
      t->numel[i] = ne[i];
      const float* src = wm.get(kParamNames[i], (int64_t)ne[i]);
      if (!src) { delete t; return fail(VNF_E_MISSING, "vnf_mlp_trainer_create: missing weight: " + wm.missing); }
      t->p[i] = (float*)t->upload(src, ne[i] * 4);
      t->g[i] = (float*)t->dalloc(ne[i] * 4);
      t->m[i] = (float*)t->dalloc(ne[i] * 4);
      t->v[i] = (float*)t->dalloc(ne[i] * 4);
      if (!t->p[i] || !t->g[i] || !t->m[i] || !t->v[i]) { delete t; return VNF_E_HIP; }
      hipError_t me = hipMemset(t->m[i], 0, ne[i] * 4);
      if (me == hipSuccess) me = hipMemset(t->v[i], 0, ne[i] * 4);
      if (me != hipSuccess) { delete t; return fail(VNF_E_HIP, std::string("vnf_mlp_trainer_create: hipMemset: ") + hipGetErrorString(me)); }
    }
    float** bufs[] = {&t->h, &t->pre, &t->z, &t->dz, &t->dh, &t->dzT, &t->hT, &t->dhT, &t->xT, &t->w2T, &t->loss_rows};
    const size_t sz[] = {B * H, B * H, B * C, B * C, B * H, C * B, H * B, H * B, D * B, H * C, B};
    for (int i = 0; i < 11; ++i) {
      *bufs[i] = (float*)t->dalloc(sz[i] * 4);
      if (!*bufs[i]) { delete t; return VNF_E_HIP; }
    }
    t->hit_rows = (int*)t->dalloc(B * 4);
    if (!t->hit_rows) { delete t; return VNF_E_HIP; }
    const hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) { delete t; return fail(VNF_E_HIP, std::string("vnf_mlp_trainer_create: ") + hipGetErrorString(se)); }
    *out = reinterpret_cast<vnf_handle>(static_cast<HandleBase*>(t));
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

static MlpTrainer* as_trainer(vnf_handle h) {
  HandleBase* b = reinterpret_cast<HandleBase*>(h);
  return (b && b->kind == 4) ? static_cast<MlpTrainer*>(b) : nullptr;
}

// forward (+ loss / hits); train != 0: backward + Adam step with learning rate lr.
extern "C" int vnf_mlp_train_step(vnf_handle h, const float* emb, const int64_t* target, int b, const float* dropout_mask, float lr,
                                  int train, float* loss_out, int32_t* hits_out, void* stream) {
  try {
    MlpTrainer* t = as_trainer(h);
    if (!t) return fail(VNF_E_INVALID, "not an MLP trainer handle");
    if (b <= 0 || b > t->max_batch) return fail(VNF_E_CAPACITY, "vnf_mlp_train_step: batch exceeds max_batch");
    if (!emb || !target) return fail(VNF_E_INVALID, "vnf_mlp_train_step: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int D = t->D, C = t->C, H = t->H;
    // forward
    VNF_HIP(gemm_nt(emb, D, t->p[0], D, t->h, H, b, H, D, t->p[1], 1, train ? dropout_mask : nullptr, t->pre, s));
    VNF_HIP(gemm_nt(t->h, H, t->p[2], H, t->z, C, b, C, H, t->p[3], 0, nullptr, nullptr, s));
    hipLaunchKernelGGL(softmax_nll_kernel, dim3((b + 3) / 4), dim3(256), 0, s, t->z, C, b, target, train ? t->dz : nullptr, t->loss_rows,
                       t->hit_rows, 1.f / (float)b);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, s, t->loss_rows, t->hit_rows, b, loss_out, hits_out);
    VNF_HIP(hipGetLastError());
    if (!train) return VNF_OK;
    // backward
    VNF_HIP(transpose(t->dz, b, C, t->dzT, s));                                                   // [C][b]
    VNF_HIP(transpose(t->h, b, H, t->hT, s));                                                     // [H][b]
    VNF_HIP(gemm_nt(t->dzT, b, t->hT, b, t->g[2], H, C, H, b, nullptr, 0, nullptr, nullptr, s));  // dW2 [C][H]
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, s, t->dz, b, C, t->g[3]);
    VNF_HIP(transpose(t->p[2], C, H, t->w2T, s));                                                 // [H][C]
    VNF_HIP(gemm_nt(t->dz, C, t->w2T, C, t->dh, H, b, H, C, nullptr, 0, nullptr, nullptr, s));    // dh [b][H]
    const size_t nh = (size_t)b * H;
    hipLaunchKernelGGL(relu_mask_grad_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, t->dh, t->pre, dropout_mask, nh);
    VNF_HIP(transpose(t->dh, b, H, t->dhT, s));                                                   // [H][b]
    VNF_HIP(transpose(emb, b, D, t->xT, s));                                                      // [D][b]
    VNF_HIP(gemm_nt(t->dhT, b, t->xT, b, t->g[0], D, H, D, b, nullptr, 0, nullptr, nullptr, s));  // dW1 [H][D]
    hipLaunchKernelGGL(colsum_kernel, dim3((H + 255) / 256), dim3(256), 0, s, t->dh, b, H, t->g[1]);
    // Adam
    t->step += 1;
    const double bc1 = 1.0 - std::pow((double)t->b1, (double)t->step), bc2 = 1.0 - std::pow((double)t->b2, (double)t->step);
    const float step_size = (float)((double)lr / bc1), bc2s = (float)std::sqrt(bc2);
    for (int i = 0; i < 4; ++i)
      hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((t->numel[i] + 255) / 256)), dim3(256), 0, s, t->p[i], t->g[i], t->m[i], t->v[i],
                         t->numel[i], t->b1, t->b2, t->eps, t->wd, step_size, bc2s);
    VNF_HIP(hipGetLastError());
    return VNF_OK;
  } catch (const std::exception& ex) {
    return fail(VNF_E_INVALID, std::string("exception: ") + ex.what());
  }
}

// kind: 0 parameter, 1 Adam exp_avg, 2 Adam exp_avg_sq; name: one of the four state_dict keys.  Synchronous copies.
static float* trainer_buf(MlpTrainer* t, const char* name, int kind, size_t* numel) {
  for (int i = 0; i < 4; ++i)
    if (name && !strcmp(name, kParamNames[i])) {
      *numel = t->numel[i];
      return kind == 0 ? t->p[i] : kind == 1 ? t->m[i] : kind == 2 ? t->v[i] : nullptr;
    }
  return nullptr;
}

extern "C" int vnf_mlp_trainer_get(vnf_handle h, const char* name, int kind, float* host_out, int64_t numel) {
  MlpTrainer* t = as_trainer(h);
  if (!t) return fail(VNF_E_INVALID, "not an MLP trainer handle");
  size_t n = 0;
  float* src = trainer_buf(t, name, kind, &n);
  if (!src || !host_out || (int64_t)n != numel) return fail(VNF_E_INVALID, "vnf_mlp_trainer_get: unknown tensor or size mismatch");
  VNF_HIP(hipDeviceSynchronize());
  VNF_HIP(hipMemcpy(host_out, src, n * 4, hipMemcpyDeviceToHost));
  return VNF_OK;
}

extern "C" int vnf_mlp_trainer_set(vnf_handle h, const char* name, int kind, const float* host_in, int64_t numel) {
  MlpTrainer* t = as_trainer(h);
  if (!t) return fail(VNF_E_INVALID, "not an MLP trainer handle");
  size_t n = 0;
  float* dst = trainer_buf(t, name, kind, &n);
  if (!dst || !host_in || (int64_t)n != numel) return fail(VNF_E_INVALID, "vnf_mlp_trainer_set: unknown tensor or size mismatch");
  VNF_HIP(hipDeviceSynchronize());
  VNF_HIP(hipMemcpy(dst, host_in, n * 4, hipMemcpyHostToDevice));
  return VNF_OK;
}

extern "C" int vnf_mlp_trainer_step_count(vnf_handle h, int64_t* step_io, int set) {
  MlpTrainer* t = as_trainer(h);
  if (!t || !step_io) return fail(VNF_E_INVALID, "not an MLP trainer handle");
  if (set) t->step = *step_io; else *step_io = t->step;
  return VNF_OK;
}
