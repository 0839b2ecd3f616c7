// Entry points declared in include/vnface.h whose kernels are not in this build yet.
// They fail loudly (no CPU fallback).  Each moves to its own translation unit when implemented.
#include "engine.h"
using namespace vnf;
extern "C" {
int vnf_mtcnn_create(const vnf_tensor_desc*, int, const vnf_tensor_desc*, int, const vnf_tensor_desc*, int, const vnf_mtcnn_cfg*, vnf_handle*) { return fail(VNF_E_INVALID, "vnf_mtcnn_create: not implemented in this build"); }
int vnf_mtcnn_detect(vnf_handle, const uint8_t*, int, int, int, int32_t*, float*, float*, float*, int, int32_t*, void*) { return fail(VNF_E_INVALID, "vnf_mtcnn_detect: not implemented in this build"); }
}
