/*
 * vnface.h -- C ABI of libvnface.so: the MI355X (gfx950) detect -> align -> embed -> classify
 * hot path of votnhan/VN_celeb_face_recognition, written from scratch in HIP.
 *
 * Each entry point replaces one reference interface (file:line under /root/reference):
 *
 *   vnf_encoder_create / vnf_embed   models/inception_resnet_v1.py:202,272-303
 *                                    (InceptionResnetV1.__init__ / forward) and
 *                                    models/iresnet_encoder.py:139-159,194-196 (iresnet100)
 *   vnf_mlp_create / vnf_classify    models/mlp_model.py:5-15 (MLPModel) +
 *                                    demo_image.py:113-137 (argmax / exp / threshold part of
 *                                    identify_person)
 *   vnf_mtcnn_create / vnf_mtcnn_detect
 *                                    models/mtcnn.py:200-227,318-361,511-513 (MTCNN.__init__,
 *                                    detect, inference) and
 *                                    models/mtcnn_utils/detect_face.py:25-185 (detect_face)
 *   vnf_align                        demo_image.py:174-199,236-239,283-295 +
 *                                    align_face.py:51-57 (crop, move landmarks, Umeyama,
 *                                    cv2.warpAffine) + data_loader/__init__.py:27-34,52-56
 *                                    (transforms_default, fused)
 *
 * Conventions
 *   - every function returns 0 on success or a negative VNF_E_* code and never throws;
 *     vnf_last_error() returns a thread-local message for the last failure;
 *   - the caller owns all input/output buffers; the library owns handles, packed weights and
 *     workspaces (allocated at create time, sized by max_batch; nothing is allocated on the
 *     launch path -- the one exception is documented at vnf_encoder_set_contexts);
 *   - a handle is bound to the device that was current at create time and is NOT thread-safe
 *     (one host thread per GPU / rank);
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); calls do not
 *     synchronise unless stated;
 *   - weights are handed over as host fp32 arrays keyed by their reference state_dict names
 *     (the Python side takes them from torch.load(...); a C caller fills the same table).
 */
#ifndef VNFACE_H
#define VNFACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNF_OK 0
#define VNF_E_INVALID (-1)   /* bad argument / shape */
#define VNF_E_MISSING (-2)   /* a required weight tensor is absent */
#define VNF_E_HIP (-3)       /* HIP runtime error */
#define VNF_E_CAPACITY (-4)  /* batch / candidate count exceeds the handle's capacity */

/* element types */
#define VNF_F32 0
#define VNF_BF16 1
#define VNF_F16 2
#define VNF_I64 3
#define VNF_U8 4
#define VNF_F16X2 5 /* compute dtype only: fp32 values kept as (hi, lo) pairs of halves (split-f16) */

/* encoder architectures */
#define VNF_ARCH_IRV1 0   /* InceptionResnetV1, 160x160 input, L2-normalised 512-d output */
#define VNF_ARCH_IR100 1  /* IResNet-100 (ArcFace), 112x112 input, 512-d BN1d features */

typedef struct vnf_handle_s* vnf_handle;

typedef struct {
  const char* name;  /* reference state_dict key, e.g. "repeat_1.0.branch0.conv.weight" */
  const void* data;  /* host pointer, contiguous */
  int32_t dtype;     /* VNF_F32 (VNF_I64 entries such as num_batches_tracked are ignored) */
  int32_t ndim;
  int64_t shape[4];
} vnf_tensor_desc;

/* library / device ------------------------------------------------------------------------- */
int vnf_init(int device_ordinal);          /* hipSetDevice + capability check (gfx950) */
const char* vnf_last_error(void);
const char* vnf_version(void);
int vnf_destroy(vnf_handle h);

/* encoders --------------------------------------------------------------------------------- */
/* compute_dtype: VNF_BF16 | VNF_F16 (MFMA 16x16x32 on 16-bit storage, fp32 accumulate);
 * VNF_F16X2 (split-f16: every weight and activation is an (hi, lo) pair of halves, products
 * expanded on the same 16-bit MFMA -- meets the <=1e-4 embedding gate at several times the
 * rate of the exact path); VNF_F32 (exact-f32 MFMA 16x16x4, bit-for-bit an fp32 fma chain). */
int vnf_encoder_create(int arch, const vnf_tensor_desc* weights, int n_weights, int compute_dtype,
                       int max_batch, vnf_handle* out);
/* x: device pointer, (N,3,S,S) NCHW, already normalised, dtype VNF_F32 | VNF_BF16 | VNF_F16.
 * emb_out: device pointer, (N,512) fp32. */
int vnf_embed(vnf_handle h, const void* x, int n, int x_dtype, float* emb_out, void* stream);
/* debugging / staged parity: copy an internal NHWC activation to a host fp32 NCHW array.
 * Synchronises the stream.  name is a reference module name ("conv2d_4b", "repeat_2", ...). */
int vnf_encoder_tap(vnf_handle h, const char* name, int n, float* host_out, int64_t capacity,
                    int64_t shape_out[4]);
/* vnf_embed with per-launch device timing (HIP events between ops; synchronises).  Writes a
 * text table (one line per plan op: shape, ms, TFLOP/s) into report. */
int vnf_encoder_profile(vnf_handle h, const void* x, int n, int x_dtype, float* emb_out, void* stream,
                        char* report, int64_t capacity);
/* FLOPs of one image through the loaded encoder as the kernels execute it (padded K / channels
 * included) and as the algorithm defines it; used by bench.py for the roofline line. */
int vnf_encoder_flops(vnf_handle h, double* algorithmic, double* executed);

/* vnf_embed cuts batches of >= 192 images over two internal HIP streams (one half each) to hide the small
 * layers' launch latencies.  A caller that already runs other work beside the encoder (the pipeline's detection
 * stream) sets 1: the forks then only take turns with that work.  Default 4 = the library decides. */
int vnf_encoder_set_streams(vnf_handle h, int max_streams);

/* Throughput mode for streams of independent batches (find_embedding.py's directory walk, the benchmark loop):
 * consecutive vnf_embed calls rotate over n (1..4) private activation-buffer sets, so calls the caller issues on
 * DIFFERENT streams overlap on the GPU; a set is re-used only after the event recorded at its previous use.
 * Each call stays ordered on the stream it was given.  Extra sets are allocated at first use (~2 GB for IRv1 at
 * max_batch 256).  Default 1. */
int vnf_encoder_set_contexts(vnf_handle h, int n);

/* classifier ------------------------------------------------------------------------------- */
int vnf_mlp_create(const vnf_tensor_desc* weights, int n_weights, int input_dim, int num_classes,
                   int max_batch, vnf_handle* out);
/* emb: device (F,input_dim) fp32.  logp_out: device (F,C) fp32 log-probabilities (may be NULL).
 * argmax_out: device (F,) int32; prob_out: device (F,) fp32 = exp(logp[argmax]) (may be NULL). */
int vnf_classify(vnf_handle h, const float* emb, int f, float* logp_out, int32_t* argmax_out,
                 float* prob_out, void* stream);

/* classifier training --------------------------------------------------- */
/* One optimisation step of trainer/classification_trainer.py:13-21 for models/mlp_model.py with
 * torch.optim.Adam semantics (coupled weight decay, no amsgrad): weights = the four MLPModel state_dict
 * tensors (initial values), fp32 end to end.  The handle owns parameters, gradients and Adam moments. */
int vnf_mlp_trainer_create(const vnf_tensor_desc* weights, int n_weights, int input_dim, int num_classes,
                           int max_batch, float beta1, float beta2, float eps, float weight_decay,
                           vnf_handle* out);
/* emb: device (b,input_dim) fp32; target: device (b,) int64; dropout_mask: device (b,2048) fp32 holding
 * F.dropout's factor per hidden unit (0 or 1/(1-p); NULL = no dropout; ignored when train == 0).
 * train != 0: forward, NLL loss, backward, Adam step with learning rate lr; train == 0: forward + loss only
 * (classification_trainer.py:48-56).  loss_out: device fp32 scalar (mean NLL of the batch); hits_out: device
 * int32 scalar (argmax == target count, losses/metrics.py:3-7).  Enqueued on `stream`, no synchronisation. */
int vnf_mlp_train_step(vnf_handle h, const float* emb, const int64_t* target, int b,
                       const float* dropout_mask, float lr, int train, float* loss_out,
                       int32_t* hits_out, void* stream);
/* checkpoint access (trainer/base_trainer.py:83-105): name = a state_dict key, kind 0 = parameter,
 * 1 = Adam exp_avg, 2 = Adam exp_avg_sq; host fp32 arrays of exactly numel elements.  Synchronise. */
int vnf_mlp_trainer_get(vnf_handle h, const char* name, int kind, float* host_out, int64_t numel);
int vnf_mlp_trainer_set(vnf_handle h, const char* name, int kind, const float* host_in, int64_t numel);
int vnf_mlp_trainer_step_count(vnf_handle h, int64_t* step_io, int set);  /* Adam's step counter */

/* detector --------------------------------------------------------------------------------- */
typedef struct {
  int32_t min_face_size;   /* mtcnn.py:201 */
  float thresholds[3];     /* mtcnn.py:202 */
  float factor;            /* mtcnn.py:202 */
  int32_t select_largest;  /* mtcnn.py:203: order boxes by area, descending */
  int32_t max_batch;       /* frames per call */
  int32_t max_height, max_width;
  int32_t max_candidates;  /* rows per frame of the stage-2 / stage-3 candidate tables (survivors of the cross-scale NMS,
                            * of the R-Net filter and of the O-Net filter): 0 or anything <= 2048 = 2048.  The reference
                            * has no cap (detect_face.py:79-93,203-218).  Here stage 1 is sized by the pyramid itself
                            * (every P-Net cell has a slot: it cannot overflow) and every NMS moves from LDS to
                            * global-memory scratch when a list outgrows the LDS tables, so this is the ONLY bound: a
                            * frame with more stage-1 survivors fails the call with VNF_E_CAPACITY (never a silent
                            * truncation) and the host layer re-creates the handle with a larger table and retries */
} vnf_mtcnn_cfg;

int vnf_mtcnn_create(const vnf_tensor_desc* pnet, int n_pnet, const vnf_tensor_desc* rnet, int n_rnet,
                     const vnf_tensor_desc* onet, int n_onet, const vnf_mtcnn_cfg* cfg, vnf_handle* out);
/* frames: device (B,H,W,3) uint8 RGB.  Results stay on the device for vnf_align and are also
 * copied to the caller's host arrays.  The reference synchronises at both stage boundaries to shape its tensors
 * (detect_face.py:96-146); this call synchronises the stream ONCE, at the end, for the results: stages 2 and 3 are
 * launched with the previous call's candidate counts (plus head room) as launch bounds, every kernel reads the true
 * counts from device memory, and the final read-back tells whether the bounds covered them.  If not -- or on the first
 * call of a frame size -- stage 1's counts are read and stages 2 / 3 run with exact bounds (one more synchronisation);
 * the results are identical either way (VNF_MTCNN_SPEC=0 always takes the second path):
 *   counts[B]            faces per frame
 *   boxes[max_out*4]     x1,y1,x2,y2 fp32, frames concatenated in order
 *   probs[max_out]
 *   points[max_out*10]   (5,2) landmarks
 * n_out receives the total number of faces; VNF_E_CAPACITY if it exceeds max_out. */
int vnf_mtcnn_detect(vnf_handle h, const uint8_t* frames, int b, int height, int width,
                     int32_t* counts, float* boxes, float* probs, float* points, int max_out,
                     int32_t* n_out, void* stream);

/* Device-resident copy of the LAST vnf_mtcnn_detect on this handle (same order as its host arrays): writes up to
 * max_out faces into caller-owned device buffers on `stream` (any of the four may be NULL).  This is what lets
 * vnf_align consume the detections without the host round trip of demo_image.py:283-295 (boxes and landmarks go
 * detector -> host -> OpenCV there).  Valid until the next vnf_mtcnn_detect on the handle. */
int vnf_mtcnn_results_device(vnf_handle h, int32_t* frame_idx, float* boxes, float* probs, float* points,
                             int max_out, void* stream);

/* measurement hook (bench.py roofline): one detection with HIP events between the cascade's stages on
 * `stream`; report receives one text line per stage, "name milliseconds algorithmic_bytes" (pyramid,
 * pnet_conv1_pool, pnet_conv2, pnet_conv3_heads, nms_stage1, host_sync_1, crop_resize_24, rnet, ...).
 * Synchronises. */
int vnf_mtcnn_stage_times(vnf_handle h, const uint8_t* frames, int b, int height, int width,
                          char* report, int64_t capacity, void* stream);

/* staged parity hook: runs the cascade on frame 0 and copies one pyramid level (3,Hs,Ws), its
 * P-Net face-probability map (oh,ow) and regression map (4,oh,ow) to host arrays.
 * dims receives {Hs, Ws, oh, ow}.  Synchronises. */
int vnf_mtcnn_debug_pnet(vnf_handle h, const uint8_t* frames, int height, int width, int level,
                         float* level_out, float* prob_out, float* reg_out, int32_t dims[4], void* stream);

/* staged parity hook for the O-stage decode alone (detect_face.py:148-169, mtcnn.py:334-340): threshold,
 * landmark decode, bbreg, "Min" NMS and the final area ordering on a caller-made single-frame table --
 * boxes (n,4) host fp32 (before bbreg), onet_out (n,15) host fp32 [prob, reg0..3, lm_x0..4, lm_y0..4];
 * fin_out (max_out,15) host rows [x1,y1,x2,y2,score, (x,y) x 5].  Lets a test inject exactly tied scores.
 * Synchronises. */
int vnf_mtcnn_debug_stage3(vnf_handle h, const float* boxes, const float* onet_out, int n, float* fin_out,
                           int max_out, int32_t* n_out, void* stream);

/* RetinaFace detector (replaces /root/reference/models/retina_face.py:56-232, the mobilenet0.25 configuration of
 * cfg/detection/retina_face.json) ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t height, width;  /* the exact frame size the handle serves (priors and the plan's buffers are sized for it;
                           * retina_face.py:180-183 rebuilds its PriorBox per image, here one handle per frame size) */
  int32_t max_batch;      /* frames per call */
  float conf_thres;       /* retina_face.py:191-195 (cfg 0.02) */
  int32_t topk_bf_nms;    /* :198-201 (cfg 5000) */
  float nms_thres;        /* :204-206 py_cpu_nms (cfg 0.4) */
  int32_t keep_top_k;     /* :209-210 (cfg 750; <= 768) */
  float vis_thres;        /* :213-216 (cfg 0.6) */
  int32_t compute_dtype;  /* VNF_F32 (0, default): the network on the exact-f32 MFMA; VNF_F16X2: split-f16 storage and
                           * products (~22 significant bits, ~15 % faster; scores move by up to ~3e-5) */
} vnf_retina_cfg;

/* weights: the RetinaFace state_dict (body.* / fpn.* / ssh{1,2,3}.* / ClassHead.* / BboxHead.* / LandmarkHead.*, without the
 * "module." prefix retina_face.py:117-127 strips), fp32 host arrays; BatchNorm is folded (eps 1e-5) at creation. */
int vnf_retina_create(const vnf_tensor_desc* weights, int n_weights, const vnf_retina_cfg* cfg, vnf_handle* out);
/* Same contract as vnf_mtcnn_detect: frames = device (B,H,W,3) uint8 RGB; per frame the rows that pass vis_thres in
 * descending score order.  VNF_E_CAPACITY when a frame has more than 16384 anchors above conf_thres or the total
 * exceeds max_out.  Synchronises the stream twice (counts, then rows). */
int vnf_retina_detect(vnf_handle h, const uint8_t* frames, int b, int height, int width,
                      int32_t* counts, float* boxes, float* probs, float* points, int max_out,
                      int32_t* n_out, void* stream);
int vnf_retina_results_device(vnf_handle h, int32_t* frame_idx, float* boxes, float* probs, float* points,
                              int max_out, void* stream);
/* staged parity hook: the raw head maps of pyramid level 0..2 of the last detection as a host (b,fh,fw,32) fp32
 * array, columns [class logits 2x2 | bbox 2x4 | landmarks 2x10]; dims receives {fh, fw}.  Synchronises. */
int vnf_retina_debug_heads(vnf_handle h, int level, int b, float* host_out, int64_t capacity, int32_t dims[2]);

/* alignment -------------------------------------------------------------------------------- */
/* For each of n faces: crop rectangle from its box (demo_image.py:179-182), landmarks moved by
 * the float box corner (236-239), Umeyama similarity landmarks -> template (align_face.py:52-54),
 * fixed-point bilinear warp into S x S (cv2.warpAffine, borderValue 0, the crop being the
 * source image), then optionally (x-127.5)/128 to NCHW.
 *   frames: device (B,H,W,3) u8; frame_idx: device (n,) int32; boxes: device (n,4) fp32;
 *   points: device (n,10) fp32; template5x2: host 10 floats.
 *   faces_u8: device (n,S,S,3) u8 or NULL; faces_norm: device (n,3,S,S) of norm_dtype or NULL. */
int vnf_align(const uint8_t* frames, int b, int height, int width, const int32_t* frame_idx,
              const float* boxes, const float* points, int n, const float* template5x2, int s,
              uint8_t* faces_u8, void* faces_norm, int norm_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VNFACE_H */
