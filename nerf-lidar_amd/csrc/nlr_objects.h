// Internal interface of the dynamic-object branch (nlr_objects.hip) used by the render loop (nlr_api.hip).
#pragma once
#include "nlr_common.h"

#define NLR_OBJ_MAX_CLASSES 8
#define NLR_OBJ_MAX_DEPTH 4
#define NLR_OBJ_MAX_DEG 4

struct NlrObjects;
int nlr_objects_apply_impl(const NlrObjects *o, const NlrRays *rays, const float *tdist, const float *box_params, uint32_t N, uint32_t S,
                           uint32_t n_obj, float *density, float *rgb, float *semantic, uint32_t K, int32_t *winner_out, void *workspace,
                           size_t workspace_bytes, hipStream_t st);
