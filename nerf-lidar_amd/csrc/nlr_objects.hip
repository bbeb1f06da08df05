// Dynamic-object branch (scope row f-1): which box, if any, owns each sample.
//
//   ZI/models.py:401-477 loops over tracks and lets every later track overwrite the samples of earlier ones, so a sample
//   belongs to the LAST track whose box contains its interval midpoint.  ZI/obj_utils.py:203-216 (box_pts) decides "inside"
//   by |p_o| < 1 on all three axes of p_o = scale * (R(p_w) + t_w_o), R = rotate_yaw_z with the reference's quirk
//   (y' from the already rotated x', obj_utils.py:106-107).
// The per-(ray, track) constants (cos, sin, t_w_o, scale) are computed by the caller with the reference's own torch
// expressions; the kernel only repeats the multiply / add chain in the same order (-ffp-contract=off), so a sample lands on
// the same side of a box face as in the reference.  One thread per sample, the track loop in registers: the reference
// materialises [N, S, N_obj, 3] tensors for this.
#include "nlr_kernels.h"

__global__ void __launch_bounds__(256) nlr_box_winner_kernel(const float *__restrict__ tdist, const float *__restrict__ origins,
                                                            const float *__restrict__ dirs, const float *__restrict__ box,
                                                            uint32_t N, uint32_t S, uint32_t n_obj, int32_t *__restrict__ winner) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * S) return;
    const uint32_t ray = (uint32_t)(i / S), k = (uint32_t)(i - (size_t)ray * S);
    const float t0 = tdist[(size_t)ray * (S + 1) + k], t1 = tdist[(size_t)ray * (S + 1) + k + 1];
    const float tm = 0.5f * (t0 + t1);
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = tm * dirs[(size_t)ray * 3 + c] + origins[(size_t)ray * 3 + c];
    int32_t w = -1;
    const float *b = box + (size_t)ray * n_obj * 8;
    for (uint32_t o = 0; o < n_obj; ++o, b += 8) {
        const float cs = b[0], sn = b[1];
        const float rx = cs * p[0] - sn * p[1];
        const float ry = sn * rx + cs * p[1];  // (sic) the rotated x
        const float x = b[5] * (rx + b[2]), y = b[6] * (ry + b[3]), z = b[7] * (p[2] + b[4]);
        if (fabsf(x) < 1.0f && fabsf(y) < 1.0f && fabsf(z) < 1.0f) w = (int32_t)o;
    }
    winner[i] = w;
}

extern "C" int nlr_box_winner(const float *tdist, const float *origins, const float *directions, const float *box_params, uint32_t N,
                              uint32_t S, uint32_t n_obj, int32_t *winner, void *stream) {
    if (N == 0 || S == 0) return NLR_OK;
    NLR_CHECK_ARG(tdist && origins && directions && winner && (box_params || n_obj == 0), "box_winner: NULL tensor");
    const size_t M = (size_t)N * S;
    hipLaunchKernelGGL(nlr_box_winner_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tdist, origins, directions,
                       box_params, N, S, n_obj, winner);
    NLR_LAUNCH_CHECK("nlr_box_winner_kernel");
    return NLR_OK;
}
