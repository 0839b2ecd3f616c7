// Face alignment on the device: crop rectangle, landmark shift, Umeyama similarity, and the
// fixed-point bilinear affine warp of cv2.warpAffine (8-bit, INTER_LINEAR, BORDER_CONSTANT 0),
// optionally fused with the (x-127.5)/128 normalisation into an NCHW tensor for the encoder.
//
// Reference semantics (file:line under /root/reference):
//   demo_image.py:179-182   crop = [max(int(x1),0), min(int(x2+1),W)) x [max(int(y1),0), min(int(y2+1),H))
//   demo_image.py:236-239   landmarks are moved by the FLOAT box corner, not the integer crop origin
//   align_face.py:52-54     SimilarityTransform.estimate(landmarks -> template)  (Umeyama)
//   align_face.py:55        cv2.warpAffine(crop, M, (S,S), borderValue=0): the CROP is the source image
//   data_loader/__init__.py:27-34  (x - 127.5) / 128, HWC -> CHW
// HBM-bound integer/byte work: one thread per output pixel, 3 channels, no tiling needed -- the
// four source taps of neighbouring pixels share cache lines; output rows are written coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "engine.h"

namespace vnf {

struct FaceXf {
  int frame, x1, y1, cw, ch;
  double m[6];  // inverse map dst -> crop coordinates, as cv::warpAffine computes it
};

__device__ __forceinline__ int sat_int(double v) {
  v = rint(v);  // round half to even == cvRound
  if (v < -2147483648.0) return INT32_MIN;
  if (v > 2147483647.0) return INT32_MAX;
  return (int)v;
}

struct Tmpl5 { float v[10]; };  // the 5-point template travels as a kernel argument: no H2D copy on the launch path

// crop rectangle + inverse similarity of face i (one thread per workgroup computes it into LDS)
__device__ __forceinline__ void face_xf(const int32_t* __restrict__ frame_idx, const float* __restrict__ boxes,
                                        const float* __restrict__ points, const Tmpl5& tmpl, int i, int B, int H, int W,
                                        FaceXf& o) {
  const float* b = boxes + 4 * i;
  const int fr = frame_idx ? frame_idx[i] : 0;
  o.frame = min(max(fr, 0), B - 1);
  // python int() truncates toward zero
  const int x1 = max((int)b[0], 0), y1 = max((int)b[1], 0);
  const int x2 = min((int)(b[2] + 1.0f), W), y2 = min((int)(b[3] + 1.0f), H);
  // a box entirely outside the frame (or a frame index outside the batch) is an empty crop: every tap is border
  // (value 0), as warping a zero-size source would be
  o.x1 = min(x1, W); o.y1 = min(y1, H);
  o.cw = (fr == o.frame) ? max(x2 - x1, 0) : 0;
  o.ch = max(y2 - y1, 0);
  // landmarks relative to the float box corner (fp32 subtraction, as numpy does on float32 arrays)
  double px[5], py[5], qx[5], qy[5];
  double pmx = 0, pmy = 0, qmx = 0, qmy = 0;
  for (int k = 0; k < 5; ++k) {
    px[k] = (double)(points[10 * i + 2 * k] - b[0]);
    py[k] = (double)(points[10 * i + 2 * k + 1] - b[1]);
    qx[k] = (double)tmpl.v[2 * k];
    qy[k] = (double)tmpl.v[2 * k + 1];
    pmx += px[k]; pmy += py[k]; qmx += qx[k]; qmy += qy[k];
  }
  pmx /= 5; pmy /= 5; qmx /= 5; qmy /= 5;
  // Umeyama in closed form for 2-D: A = dd^T sd / n; rotation+scale = ((a11+a22), (a21-a12)) / var(src)
  double a11 = 0, a12 = 0, a21 = 0, a22 = 0, var = 0;
  for (int k = 0; k < 5; ++k) {
    const double sx = px[k] - pmx, sy = py[k] - pmy, dx = qx[k] - qmx, dy = qy[k] - qmy;
    a11 += dx * sx; a12 += dx * sy; a21 += dy * sx; a22 += dy * sy;
    var += sx * sx + sy * sy;
  }
  a11 /= 5; a12 /= 5; a21 /= 5; a22 /= 5; var /= 5;
  // coincident landmarks (var == 0) have no similarity: a singular map (everything samples one point) instead of NaN
  const double sc = var > 0 ? (a11 + a22) / var : 0.0, ss = var > 0 ? (a21 - a12) / var : 0.0;
  double M[6] = {sc, -ss, qmx - (sc * pmx - ss * pmy), ss, sc, qmy - (ss * pmx + sc * pmy)};
  // cv::warpAffine: invert the forward matrix in double
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  for (int k = 0; k < 6; ++k) o.m[k] = M[k];
}

template <typename TN>
__global__ void __launch_bounds__(256) align_warp_kernel(const uint8_t* __restrict__ frames, int B, int H, int W,
                                                         const int32_t* __restrict__ frame_idx, const float* __restrict__ boxes,
                                                         const float* __restrict__ points, const Tmpl5 tmpl, int S,
                                                         uint8_t* __restrict__ out_u8, TN* __restrict__ out_norm) {
  // blockIdx.y = face; its workgroups (blockIdx.x) each derive the face's transform (a few hundred fp64 operations on
  // one lane) instead of reading a table a setup launch would have to allocate and fill
  __shared__ FaceXf s_xf;
  const int f = blockIdx.y;
  if (threadIdx.x == 0) face_xf(frame_idx, boxes, points, tmpl, f, B, H, W, s_xf);
  __syncthreads();
  const FaceXf t = s_xf;
  const int per_face = S * S;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < per_face; q += gridDim.x * blockDim.x) {
    const int x = q % S, y = q / S;
    const size_t i = (size_t)f * per_face + q;
    constexpr int AB_BITS = 10, INTER_BITS = 5, AB_SCALE = 1 << AB_BITS, TAB = 1 << INTER_BITS;
    constexpr int round_delta = AB_SCALE / TAB / 2;
    const int adelta = sat_int(t.m[0] * x * AB_SCALE);
    const int bdelta = sat_int(t.m[3] * x * AB_SCALE);
    const int X0 = sat_int((t.m[1] * y + t.m[2]) * AB_SCALE) + round_delta;
    const int Y0 = sat_int((t.m[4] * y + t.m[5]) * AB_SCALE) + round_delta;
    const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS);
    const int Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
    int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
    sx = min(max(sx, -32768), 32767);  // saturate_cast<short>
    sy = min(max(sy, -32768), 32767);
    const int fx = X & (TAB - 1), fy = Y & (TAB - 1);
    const int w00 = (TAB - fy) * (TAB - fx) * 32, w01 = (TAB - fy) * fx * 32;
    const int w10 = fy * (TAB - fx) * 32, w11 = fy * fx * 32;
    const uint8_t* base = frames + ((size_t)t.frame * H + t.y1) * (size_t)W * 3 + (size_t)t.x1 * 3;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = 0;
    auto tap = [&](int yy, int xx, int w) {
      if (w != 0 && (unsigned)yy < (unsigned)t.ch && (unsigned)xx < (unsigned)t.cw) {
        const uint8_t* p = base + ((size_t)yy * W + xx) * 3;
        v[0] += p[0] * w; v[1] += p[1] * w; v[2] += p[2] * w;
      }
    };
    tap(sy, sx, w00); tap(sy, sx + 1, w01); tap(sy + 1, sx, w10); tap(sy + 1, sx + 1, w11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int r = (v[c] + (1 << 14)) >> 15;
      r = min(max(r, 0), 255);
      if (out_u8) out_u8[i * 3 + c] = (uint8_t)r;
      if (out_norm) out_norm[((size_t)f * 3 + c) * S * S + (size_t)y * S + x] = (TN)(((float)r - 127.5f) / 128.0f);
    }
  }
}

}  // namespace vnf

using namespace vnf;

extern "C" int vnf_align(const uint8_t* frames, int b, int height, int width, const int32_t* frame_idx,
                         const float* boxes, const float* points, int n, const float* template5x2, int s,
                         uint8_t* faces_u8, void* faces_norm, int norm_dtype, void* stream) {
  if (n == 0) return VNF_OK;
  if (!frames || !boxes || !points || !template5x2 || n < 0 || s <= 0 || b <= 0 || (!faces_u8 && !faces_norm))
    return fail(VNF_E_INVALID, "vnf_align: bad argument");
  hipStream_t st = (hipStream_t)stream;
  Tmpl5 tmpl;  // travels as a kernel argument; no workspace, nothing allocated or copied on the launch path
  for (int i = 0; i < 10; ++i) tmpl.v[i] = template5x2[i];
  if (n > 65535) return fail(VNF_E_CAPACITY, "vnf_align: at most 65535 faces per call");
  const int per_face = (s * s + 255) / 256;
  const dim3 grid(per_face < 8 ? per_face : 8, n);
  switch (faces_norm ? norm_dtype : VNF_F32) {
    case VNF_F32:
      hipLaunchKernelGGL(align_warp_kernel<float>, grid, dim3(256), 0, st, frames, b, height, width, frame_idx, boxes, points,
                         tmpl, s, faces_u8, (float*)faces_norm);
      break;
    case VNF_BF16:
      hipLaunchKernelGGL(align_warp_kernel<__bf16>, grid, dim3(256), 0, st, frames, b, height, width, frame_idx, boxes, points,
                         tmpl, s, faces_u8, (__bf16*)faces_norm);
      break;
    case VNF_F16:
      hipLaunchKernelGGL(align_warp_kernel<_Float16>, grid, dim3(256), 0, st, frames, b, height, width, frame_idx, boxes,
                         points, tmpl, s, faces_u8, (_Float16*)faces_norm);
      break;
    default:
      return fail(VNF_E_INVALID, "vnf_align: bad norm_dtype");
  }
  VNF_HIP(hipGetLastError());
  return VNF_OK;
}
