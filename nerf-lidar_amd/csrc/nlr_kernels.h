// Kernel parameter blocks and internal launchers shared between the kernel files and nlr_api.hip.
#pragma once
#include "nlr_common.h"

#define NLR_MAX_MULTI 16

struct CastParams {
    const float *origins, *directions, *base_x, *base_y, *radii;  // [N,3]/[N,1]
    const float *tdist;      // [N, S+1]
    const float *rand_deg;   // [N, S, n] uniform draws or null
    uint32_t N, S, n;        // rays, samples per ray, multisamples
    // Which samples share a wave of the fused cast + encode kernels (nlr_encode.hip:nlr_slot_sample; no effect on the results):
    // 0 = 8 consecutive samples of ONE ray, 1 = 8 ADJACENT rays at one sample index, 2 = decided on the device from `votes`
    // (nlr_launch_ray_vote: 1 when more than vote_min of the (adjacent rays, sample) pairs are coherent).
    uint32_t ray_groups, vote_min;
    const uint32_t *votes;
    float std_scale;
    float cosd[NLR_MAX_MULTI], sind[NLR_MAX_MULTI];  // cos/sin(2*pi*m*j/n), render.py:148
    float degj[NLR_MAX_MULTI];                       // the angles themselves (rand path)
};

struct CompositeParams {
    const float *density, *tdist, *dirs, *rgb, *sem, *inten, *far, *origins;
    uint32_t N, S, K;
    int opaque, extras;
    float bg, scale_factor;
    float *weights;  // [N,S] or null
    float *o_rgb, *o_depth, *o_sem, *o_int, *o_acc, *o_dmean, *o_dmed, *o_p5, *o_p95, *o_points, *level_depth;
    int32_t *o_labels;
    const float *seg;        // (instead of rgb / sem / inten) segment records of the compositing-mode MLP kernel, int_row = K
    uint32_t seg_int;        // the records carry the intensity in slot K
    float *o_packed;         // [N, 7] records (see NlrOut.packed)
    uint32_t pk_h, pk_w;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct TileH {
    bf16x8 f[2];  // 32 features x 32 samples as two k-steps of 16 (permuted k order, see pack_bf16)
};

struct MlpParams {
    const float *feat;      // [M, F] f32, or [F/4][M][4] when feat_piece_major
    int feat_piece_major;
    uint32_t M, S, F;       // samples, samples per ray, feature count
    // Weight tape: every GEMM's A fragments (1 KiB each: 64 lanes x 16 B) in consumption order
    // D0, D2, [H1, H2], V0, V1, V2..V(D-1), RGB (fragments output-tile-major); each GEMM padded to whole 32 KiB chunks.
    const uint4 *tape;
    uint32_t tape_chunks;
    // all biases, concatenated and zero-padded to whole 32-row tiles, copied to LDS once per workgroup:
    // b_d0 (64) | b_d2 (WB) | b_h1 (32*HT) | b_h2 (32) | view0 (W) | view1 (W) | view2.. (W each) | rgb (32)
    const float *bias_all;
    uint32_t bias_count;
    const float *enc;       // [N, 32] per-ray direction encoding (27 values + zero padding), one extra input tile
    uint32_t depth;         // net_depth_viewdirs
    uint32_t K, int_row;    // class_num (0 = no semantic head), row of the intensity output (or 0xffffffff)
    float density_bias, rgb_premul, rgb_bias, rgb_padding;
    float *density, *rgb, *sem, *inten;  // outputs: [M], channel-major [3,M], class-major [K,M], [M]
    // compositing mode (see nlr_mlp_kernel.h, COMP): instead of rgb / sem / inten one 32-float record per 32-sample segment
    const float *tdist;     // [N, S+1]
    const float *dnorm;     // [N] |directions|
    float *seg;             // [M / 32, 32]
    uint32_t opaque;
};

struct DirEncParams {
    const float *viewdirs;  // [N,3]
    const float *dirs;      // [N,3] or null
    uint32_t N, deg, E;
    float *out;             // [N, 32]
    float *dnorm;           // [N] |dirs| or null
};

int nlr_launch_resample(const float *prev_sdist, const float *prev_weights, uint32_t n_prev, float dilation, float anneal,
                        float pad, uint32_t S, const float *u_dev, const float *jitter, float max_jitter, const float *near,
                        const float *far, float lam, uint32_t N, float *sdist, float *tdist, hipStream_t st);
int nlr_fill_cast_params(CastParams *cp, const NlrRays *rays, const float *tdist, const float *rand_deg, uint32_t N,
                         uint32_t S, uint32_t n, uint32_t mloops, float std_scale);
// piece_major: features as [F/4][M][4] (only honoured by the 8-lane kernel with C == 4; see nlr_encode8_kernel)
int nlr_launch_encode(const CastParams &cp, const GridParams &gp, int re_weights, float *feat, int piece_major, hipStream_t st);
int nlr_launch_ray_vote(const float *tdist, uint32_t N, uint32_t S, uint32_t *votes, hipStream_t st);
int nlr_launch_prop(const CastParams &cp, const GridParams &gp, const float *w1, const float *b1, const float *w2, float b2,
                    float density_bias, int re_weights, float *density, float *feat_out, hipStream_t st);
int nlr_launch_direnc(const DirEncParams &P, hipStream_t st);
int nlr_launch_mlp(const MlpParams &P, uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, uint32_t cus, hipStream_t st);
// can nlr_launch_mlp run this level in compositing mode (MlpParams.seg)?
bool nlr_mlp_can_composite(uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, uint32_t F, uint32_t S, uint32_t K, bool use_int, uint64_t M);
int nlr_launch_composite(const CompositeParams &P, hipStream_t st);
int nlr_composite_segments(const float *density, const float *tdist, const float *directions, const float *seg, uint32_t class_num,
                           int has_intensity, const float *far, const float *origins, uint32_t N, uint32_t S, int opaque_background,
                           float bg, int compute_extras, float scale_factor, float *weights, const NlrOut *out, float *level_depth,
                           hipStream_t st);
