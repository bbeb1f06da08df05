/*
 * nerflidar_hip.h -- C ABI of libnerflidar_hip.so (MI355X / gfx950 volume-render hot path).
 *
 * Drop-in boundary for fudan-zvg/NeRF-LiDAR's render path.  Citations are relative to
 * NeRF_LiDAR/zipnerf/ in the reference (ZI = internal/).  Everything is `extern "C"`, plain
 * pointers and sizes, no torch types.  Conventions:
 *   - every function returns 0 (NLR_OK) or a negative NLR_ERR_* code; nlr_last_error() returns a
 *     thread-local message (the reference raises via TORCH_CHECK, gridencoder.cu:15-18);
 *   - all tensors are caller-owned and contiguous; "dev" = device (HBM) pointer, "host" = host
 *     pointer; kernels never allocate (gridencoder/grid.py:47-50 does the same);
 *   - `stream` is a hipStream_t passed as void* (NULL = legacy default stream).  The reference
 *     always launches on stream 0 (gridencoder.cu:377); callers should pass PyTorch's current
 *     stream;
 *   - re-entrant, no global mutable state besides the thread-local error string.
 */
#ifndef NERFLIDAR_HIP_H
#define NERFLIDAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLR_MAX_LEVELS 4      /* sampling levels (Model.num_levels, ZI/models.py:37) */
#define NLR_MAX_GRID_LEVELS 16
#define NLR_MAX_VIEW_DEPTH 16

enum {
    NLR_OK = 0,
    NLR_ERR_INVALID = -1,     /* bad argument (shape, null pointer, alignment) */
    NLR_ERR_UNSUPPORTED = -2, /* configuration outside the fused path (see message) */
    NLR_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed */
    NLR_ERR_WORKSPACE = -4    /* workspace too small: see nlr_workspace_bytes */
};

/* MLP arithmetic (NlrModelDesc.mlp_precision) */
enum {
    NLR_PREC_F32 = 0,   /* every layer on exact-f32 MFMA (v_mfma_f32_16x16x4_f32): reference-grade */
    NLR_PREC_MIXED = 1, /* density/semantic/intensity layers f32 MFMA, view-MLP bf16 MFMA */
    NLR_PREC_FAST = 2   /* density/semantic/intensity layers split-bf16 (hi+lo, 3 MFMAs), view bf16: the default of the Python host side, bench.py and smoke() */
};

const char *nlr_last_error(void);
int nlr_version(void);

/* ------------------------------------------------------------------------------------------
 * (1) Hash-grid operator.  Replaces the pybind entry points of gridencoder/src/bindings.cpp:5-7
 *     (declared gridencoder/src/gridencoder.h:12-15).
 *
 * nlr_grid_encode_forward  <->  grid_encode_forward(inputs, embeddings, offsets, outputs,
 *                                   B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp)
 *   inputs      dev f32 [B, D]          points in [0,1]; anything outside -> zeros (cu:110-135)
 *   embeddings  dev [sO, C]             table, f32 (table_dtype 0) or f16 (1; grid.py:43-44)
 *   offsets     HOST i32 [L+1]          level offsets (grid.py:122-137).  The reference passes a
 *                                       device tensor; the launcher needs the values on the host
 *                                       (per-level scale/size go into kernel arguments), so the
 *                                       Python shim keeps a CPU copy -- they never change.
 *   outputs     dev f32: out_layout 0 = [L, B, C] (what the reference kernel writes, cu:388),
 *                        out_layout 1 = [B, L*C] (what grid.py:57 permutes it into)
 *   dy_dx       dev f32 [B, L*D*C] or NULL (cu:201-244)
 *   D must be 3; C in {1,2,4,8}; L <= 16; S = log2(per_level_scale); H = base resolution.
 * ------------------------------------------------------------------------------------------ */
int nlr_grid_encode_forward(const float *inputs, const void *embeddings, int table_dtype,
                            const int32_t *offsets_host, float *outputs, uint32_t B, uint32_t D,
                            uint32_t C, uint32_t L, float S, uint32_t H, float *dy_dx,
                            uint32_t gridtype, int align_corners, uint32_t interp, int out_layout,
                            void *stream);

/* nlr_grid_encode_backward <-> grid_encode_backward (gridencoder.h:13, cu:473-503).
 *   grad [L,B,C] (grad_layout 0) or [B,L*C] (1); grad_embeddings dev f32 [sO,C] must be zeroed by
 *   the caller (grid.py:77); float atomics -> summation order not reproducible, like the reference.
 *   dy_dx / grad_inputs may be NULL together. */
int nlr_grid_encode_backward(const float *grad, const float *inputs, const int32_t *offsets_host,
                             float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                             uint32_t gridtype, int align_corners, uint32_t interp, int grad_layout,
                             void *stream);

/* The same scatter with a caller-provided workspace (round 4): when `workspace_bytes` >= nlr_grid_backward_workspace_bytes(...) and the batch
 * is large (B * C >= 2^18), the levels that do not fit an LDS copy are scattered through BINS - corner updates written bucket by bucket
 * (128 KiB of table per bucket) into the workspace, then accumulated per bucket in LDS and added to the table with coalesced atomics -
 * instead of one scattered global atomic per merged corner update (csrc/nlr_grid.hip, "Binned scatter").  Same result up to the order
 * of float additions; workspace NULL or too small = nlr_grid_encode_backward.  C in {1, 2} (proposal / object grids; 0 bytes otherwise). */
size_t nlr_grid_backward_workspace_bytes(uint32_t B, uint32_t C, uint32_t L, float S, uint32_t H, const int32_t *offsets_host,
                                         uint32_t gridtype, int align_corners);
int nlr_grid_encode_backward_ws(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                                uint32_t gridtype, int align_corners, uint32_t interp, int grad_layout, void *workspace,
                                size_t workspace_bytes, void *stream);

/* nlr_grad_total_variation <-> grad_total_variation (gridencoder.h:15, cu:506-645; bound by grid.py:176-198).
 *   inputs dev f32 [B, D] already mapped to [0,1] (points outside are skipped); embeddings dev f32 [sO, C];
 *   grad dev f32 [sO, C] is accumulated into (embeddings.grad, between loss.backward() and optimizer.step()):
 *   grad[cell] += weight / (2 D) * sum_nb (e[cell] - e[nb]) * rsqrt(sum_nb (e[cell] - e[nb])^2 + 1e-9) over the 2 D axis
 *   neighbours of the cell corner floor(x * scale + 0.5) that lie inside the level.  f32 tables only (grid.py runs it with
 *   autocast disabled); float atomics, order not reproducible, like the reference. */
int nlr_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets_host, float weight,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                             int align_corners, void *stream);

/* Host helper shared with the CPU checker so both use bit-identical level constants
 * (scale = exp2f(l*S)*H - 1, resolution = ceil(scale)+1; cu:138-139). */
void nlr_level_scale(uint32_t L, float S, uint32_t H, float *scale, uint32_t *resolution);

/* torch.linspace(start, end, n) in float32, as ZI/stepfun.py:203-216 uses it for the inverse-CDF
 * sample positions u.  rand = 0: linspace(1/2n, 1 - 1/2n - eps, n); rand = 1: linspace(0, 1-u_max, n)
 * and *max_jitter = (1-u_max)/(n-1) - eps.  Host only. */
void nlr_sample_u(uint32_t n, int rand, float *u_host, float *max_jitter);

/* ------------------------------------------------------------------------------------------
 * (2) Model object: packed MLP weights (the only thing the library owns).
 * ------------------------------------------------------------------------------------------ */
typedef struct NlrLinear {           /* torch.nn.Linear: y = x W^T + b */
    const float *weight;             /* HOST f32 [out_features, in_features] (state_dict tensor) */
    const float *bias;               /* HOST f32 [out_features] */
    uint32_t out_features, in_features;
} NlrLinear;

typedef struct NlrGridDesc {         /* gridencoder.GridEncoder (grid.py:96-149) */
    const void *table;               /* DEV  [sO, C]; not copied: training may update it in place */
    int32_t table_dtype;             /* 0 = f32, 1 = f16 */
    uint32_t num_levels, level_dim;  /* L, C */
    uint32_t base_resolution;        /* H */
    float log2_per_level_scale;      /* S */
    const int32_t *offsets;          /* HOST i32 [L+1] (copied) */
    uint32_t gridtype;               /* 0 hash, 1 tiled */
    uint32_t align_corners;
    uint32_t interp;                 /* 0 linear, 1 smoothstep */
} NlrGridDesc;

typedef struct NlrMlpDesc {          /* ZI/models.py:MLP (796-961) as configured at inference */
    NlrGridDesc grid;
    NlrLinear density0, density2;    /* density_layer.0 (L*C->64), density_layer.2 (64-> 1 | bottleneck) */
    uint32_t disable_rgb;            /* PropMLP: density only (models.py:1119-1122) */
    uint32_t bottleneck_width;       /* 256 */
    uint32_t net_depth_viewdirs, net_width_viewdirs, skip_layer_dir, deg_view;
    NlrLinear view[NLR_MAX_VIEW_DEPTH]; /* lin_second_stage_i (models.py:939-950) */
    NlrLinear rgb_layer;
    uint32_t use_semantic, no_sem_layer, class_num;
    NlrLinear sem0, sem2;            /* sem_layer.0/.2 (models.py:955-957) */
    uint32_t use_intensity;
    NlrLinear int0, int2;            /* intensity_layer.0/.2 (models.py:959-961) */
    float density_bias, rgb_premultiplier, rgb_bias, rgb_padding;
    uint32_t re_weights;             /* erf down-weighting of fine levels (models.py:975-977) */
} NlrMlpDesc;

typedef struct NlrModelDesc {        /* ZI/models.py:Model (31-58) */
    uint32_t num_levels;                         /* sampling levels; last one is the NerfMLP */
    uint32_t num_samples[NLR_MAX_LEVELS];        /* num_prop_samples..., num_nerf_samples */
    const NlrMlpDesc *mlps[NLR_MAX_LEVELS];      /* prop_mlp_0.., nerf_mlp */
    float dilation_multiplier, dilation_bias;    /* 0.5, 0.0025 */
    float anneal_slope, resample_padding;        /* 10, 0 */
    float power_lambda, std_scale;               /* -1.5, 0.35 */
    float bg_intensity;                          /* bg_intensity_range midpoint (1.0) */
    uint32_t opaque_background;
    uint32_t mlp_precision;                      /* NLR_PREC_* */
} NlrModelDesc;

typedef struct NlrModel NlrModel;
/* Packs the weights and uploads them with copies enqueued on `stream`; returns once they have completed (the host staging
 * buffers are temporaries).  The model is bound to the device that is current at creation (its CU count sizes the
 * persistent grid of the MLP kernel). */
int nlr_model_create(const NlrModelDesc *desc, NlrModel **out, void *stream);
void nlr_model_destroy(NlrModel *m);
/* Re-point a level's hash table (e.g. after an optimizer step re-allocated the parameter). */
int nlr_model_set_table(NlrModel *m, uint32_t level, const void *table_dev, int table_dtype);

/* ------------------------------------------------------------------------------------------
 * (3) Fused render op.  Replaces Model.forward (ZI/models.py:239-576) for instance_obj=False,
 *     num_glo_features=0:  per level  max_dilate_weights -> sample_intervals -> s_to_t ->
 *     cast_rays -> MLP -> compute_alpha_weights -> volumetric_rendering.
 * ------------------------------------------------------------------------------------------ */
typedef struct NlrRays {             /* the batch dict of ZI/lidar_utils.py:8-33 / camera_utils.py:567-617 */
    const float *origins, *directions, *viewdirs;   /* dev f32 [N,3] */
    const float *radii, *near, *far;                /* dev f32 [N,1] */
    const float *base_x, *base_y;                   /* dev f32 [N,3] (LiDAR: both = directions) */
} NlrRays;

typedef struct NlrRenderCfg {
    float train_frac;                /* 1.0 at render time (models.py:343-346 anneal) */
    uint32_t compute_extras;         /* acc, distance_mean, percentiles (render.py:255-282) */
    uint32_t sample_n, sample_m;     /* 7, 3 multisamples / loops (configs.py:143-148) */
    /* rand=True support: per-level uniform [0,1) draws replacing torch.rand; NULL = deterministic.
       jitter [N] (stepfun.py:216, single_jitter), deg [N, S_l, sample_n] (render.py:150). */
    const float *rand_jitter[NLR_MAX_LEVELS];
    const float *rand_deg[NLR_MAX_LEVELS];
    float scale_factor;              /* >0: also emit points/labels (render_lidar.py:142-161) */
    /* Performance hint, no effect on the results.  0 (sweeps in beam-major / azimuth-fastest order, images and tiles in row-major order):
     * consecutive rays of the batch are neighbours in space; per level the library then counts on the device whether their samples are
     * too (nlr_ray_vote_kernel) and lets a wave of the fused cast + encode kernels take either 8 ADJACENT rays at one sample index (a
     * trained field: they gather the same table lines) or 8 consecutive samples of ONE ray.  1: the batch is shuffled (training-style
     * random rays): always the latter, nothing is counted. */
    uint32_t shuffled_rays;
} NlrRenderCfg;

typedef struct NlrLevelOut {         /* one ray_history entry (models.py:553-557); any may be NULL */
    float *sdist, *tdist;            /* [N, S+1] */
    float *weights, *density;        /* [N, S] */
    float *rgb;                      /* channel-major [3, N, S]  (final level) */
    float *semantic;                 /* class-major [class_num, N, S] probabilities (final level) */
    float *intensity;                /* [N, S] (final level) */
    float *depth;                    /* [N] per-level rendering['depth'] */
    /* the rest of the level's `rendering` dict (ZI/models.py:514-531) for the levels before the last (whose rendering is
     * NlrOut itself): proposal MLPs return rgb = 0, so r_rgb is the background seen through the level's weights */
    float *r_rgb;                    /* [N,3] */
    float *r_acc, *r_distance_mean, *r_distance_median, *r_distance_percentile_5, *r_distance_percentile_95; /* [N] */
} NlrLevelOut;

typedef struct NlrOut {              /* renderings[-1] (render.py:219-284); any may be NULL */
    float *rgb;                      /* [N,3] */
    float *depth;                    /* [N]   */
    float *semantic;                 /* [N,class_num] */
    float *intensity;                /* [N]   */
    float *acc, *distance_mean, *distance_median, *distance_percentile_5, *distance_percentile_95; /* [N] */
    int32_t *labels;                 /* [N] argmax_c semantic (render_lidar.py:158-159) */
    float *points;                   /* [N,3] (o + depth*d)/scale_factor (render_lidar.py:142-156) */
    /* One 7-float record per ray: depth, intensity, acc, rgb[3], label (as float; exact below 2^24) -- the tile a rank
     * contributes to the all-gathered range image, written by the compositing kernel itself.  Record of ray i:
     * packed_h == 0: row i.  packed_h = H > 0: the batch is a beam-major [H, packed_w] azimuth sector (ray = beam * packed_w
     * + column) and the record goes to row column * H + beam, i.e. the tile is [packed_w, H, 7] (azimuth-major), so that
     * the rank-major concatenation an all-gather produces IS the [W, H, 7] sweep image, with no reassembly pass. */
    float *packed;
    uint32_t packed_h, packed_w;
    NlrLevelOut history[NLR_MAX_LEVELS];
} NlrOut;

size_t nlr_workspace_bytes(const NlrModel *m, uint32_t N);
int nlr_render_rays(const NlrModel *m, const NlrRays *rays, uint32_t N, const NlrRenderCfg *cfg,
                    const NlrOut *out, void *workspace, size_t workspace_bytes, void *stream);

/* Comma-separated names of the kernels the library launches, in NLR_K_* order (host only), so that
 * bench.py can match rocprofv3 --kernel-trace rows. */
const char *nlr_kernel_names(void);

/* SHA-256 (hex) of the kernel sources this BINARY was compiled from (every file under csrc/, this header, the Makefile; the hash of
 * nerflidar_hip/buildinfo.py), fixed at compile time.  bench.py stamps its JSON line with it and quotes PMC traffic only from a
 * profile that recorded the same value; a library that is stale against the sources next to it is detectable
 * (buildinfo.kernel_source_sha() != nlr_build_sha()). */
const char *nlr_build_sha(void);

/* Diagnostic switches (round 4, ADVICE r3: they used to be environment variables read inside the library).  Process-wide, explicit,
 * and readable back so that a benchmark line can record them; all default to 0 and nothing else in the library reads the environment.
 *   NLR_DBG_FORCE_GENERIC (key 0): 1 sends the fused encode / proposal launches through the generic level body (nlr_encode8g_kernel /
 *     nlr_prop8g_kernel) - the bit-identity test of the fast body.
 *   NLR_DBG_MLP_WORKGROUPS (key 1): n > 0 caps the persistent MLP grid of models created AFTERWARDS at n workgroups (power / clock
 *     experiments, profiles/r03_mlp_cu_sweep.txt).
 *   NLR_DBG_BINNED_C4 (key 2): 1 lets the binned scatter (nlr_grid_encode_backward_ws) take level_dim = 4 grids as well (A/B only).
 *   NLR_DBG_NO_XPAIR_SCATTER (key 3): 1 sends the atomic scatter of level_dim <= 4 grids through the one-corner-per-instruction kernel of
 *     rounds 2-3 (nlr_grid_bwd_kernel) instead of nlr_grid_bwd_xpair_kernel (A/B only; same sums up to the order of the atomics).
 *   NLR_DBG_SCATTER_LEVELS (key 4): a non-zero bit mask restricts nlr_grid_bwd_xpair_kernel to the levels whose bit is set (per-level
 *     timing, scripts/train_scene_profile.py; the gradient of the other levels is then NOT written).
 *   NLR_DBG_NO_SCATTER_CACHE (key 5): 1 keeps the coarse levels beyond the LDS copy (up to 128^3 cells) out of the tagged LDS cache of
 *     nlr_grid_bwd_lds_kernel: they are scattered (atomics / bins) as in round 3 (A/B only; nlr_grid_backward_workspace_bytes follows).
 *   NLR_DBG_RAY_GROUPS (key 6): model renders normally decide per level on the device which samples share a wave of the encode kernels
 *     (NlrRenderCfg.shuffled_rays); 1 forces 8 adjacent rays x one sample index, 2 forces 8 consecutive samples of one ray (A/B only). */
#define NLR_DBG_FORCE_GENERIC 0
#define NLR_DBG_MLP_WORKGROUPS 1
#define NLR_DBG_BINNED_C4 2
#define NLR_DBG_NO_XPAIR_SCATTER 3
#define NLR_DBG_SCATTER_LEVELS 4
#define NLR_DBG_NO_SCATTER_CACHE 5
#define NLR_DBG_RAY_GROUPS 6
int nlr_debug_set(uint32_t key, int value);
int nlr_debug_get(uint32_t key);
/* 1 when the fused kernels' fast level body covers this grid (see csrc/nlr_level_fast.h:nlr_level_fast_ok), 0 when the generic body
 * serves it; host arithmetic only (offsets_host [L+1]). */
int nlr_grid_fast_path(const int32_t *offsets_host, uint32_t L, uint32_t C, float S, uint32_t H, int table_dtype, uint32_t gridtype,
                       int align_corners, uint32_t interp);

/* Per-kernel timing with HIP events recorded on the SAME stream the kernels are launched on.
 * nlr_profile_begin arms the model: every launch made by nlr_render_rays / nlr_mlp_level is then
 * bracketed by an event pair.  nlr_profile_end synchronises the stream, writes total milliseconds and
 * launch counts per kernel kind (arrays of NLR_K_COUNT) and disarms.  Costs two hipEventRecord per
 * launch while armed; nothing when not armed. */
enum { NLR_K_RESAMPLE = 0, NLR_K_PROP = 1, NLR_K_ENCODE = 2, NLR_K_DIRBIAS = 3, NLR_K_MLP = 4, NLR_K_COMPOSITE = 5, NLR_K_COUNT = 6 };
int nlr_profile_begin(NlrModel *m);
/* The same for a subset of the kernel kinds (bit k of kind_mask = NLR_K_k): a pair of event records costs ~4 us of stream time, ten
 * kernels per sweep make that 0.1 ms of a 7 ms step (profiles/r03_emulated_sector_steps.txt), so bench.py brackets only the two kernels
 * its roofline objects need inside the timed region and takes the others' durations in a separate, untimed pass. */
int nlr_profile_begin_kinds(NlrModel *m, uint32_t kind_mask);
int nlr_profile_end(NlrModel *m, void *stream, float *total_ms, uint32_t *launches);

/* ------------------------------------------------------------------------------------------
 * (4) Stage entry points: the same kernels nlr_render_rays chains, exposed one by one so each
 *     row of the scope table can be parity-checked in isolation.
 * ------------------------------------------------------------------------------------------ */
/* a-3 + a-4 + a-2: resample one level.  prev_* NULL/n_prev = 0 means the initial [0,1] interval
 * (models.py:296-300).  dilation <= 0 skips max_dilate_weights (level 0, models.py:332). */
int nlr_resample_level(const float *prev_sdist, const float *prev_weights, uint32_t n_prev,
                       float dilation, float anneal, float resample_padding, uint32_t num_samples,
                       const float *rand_jitter, const float *near, const float *far,
                       float power_lambda, uint32_t N, float *sdist, float *tdist, void *stream);

/* a-5..a-12: cast + contract + encode + MLP for level `level` of the model.  tdist [N,S+1].
 * density [N,S]; rgb channel-major [3,N,S]; semantic class-major [K,N,S]; intensity [N,S]. */
int nlr_mlp_level(const NlrModel *m, uint32_t level, const NlrRays *rays, const float *tdist,
                  uint32_t N, uint32_t sample_n, uint32_t sample_m, const float *rand_deg,
                  float *features /* [N*S, L*C] or NULL */, float *density, float *rgb,
                  float *semantic, float *intensity, void *workspace, size_t workspace_bytes,
                  void *stream);

/* a-13 + a-14 (+ a-16 post-step when labels/points are given).  rgb [3,N,S] and semantic [K,N,S] are
 * channel-/class-major (what nlr_mlp_level writes: coalesced for both kernels). */
int nlr_composite_level(const float *density, const float *tdist, const float *directions,
                        const float *rgb, const float *semantic, const float *intensity,
                        const float *far, const float *origins, uint32_t N, uint32_t S,
                        uint32_t class_num, int opaque_background, float bg, int compute_extras,
                        float scale_factor, float *weights, const NlrOut *out, float *level_depth,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * (5) Renderer output -> ray-drop UNet input.  Replaces LaserScan.do_range_projection
 *     (NeRF_Lidar_code/src/lidar_utils.py:215-282): spherical projection into an H x W range image where the NEAREST
 *     point of a pixel wins (the reference sorts far->near and scatters).  points are float64 [N,3] in the LiDAR
 *     frame (what nerf2world.py:22-38 produces); semantic [N] f32 / rgb [N,3] f32 may be NULL.  Outputs (any may be
 *     NULL): proj_range [H,W] (-1 = empty), proj_xyz [H,W,3], proj_semantic [H,W], proj_rgb [H,W,3], proj_idx [H,W]
 *     (-1 = empty), proj_mask [H,W] = (proj_idx > 0) as in the reference.
 * ------------------------------------------------------------------------------------------ */
size_t nlr_range_workspace_bytes(uint32_t H, uint32_t W);
int nlr_range_project(const double *points, const float *semantic, const float *rgb, uint32_t N, uint32_t H, uint32_t W,
                      float fov_up_deg, float fov_down_deg, void *workspace, size_t workspace_bytes, float *proj_range,
                      float *proj_xyz, float *proj_semantic, float *proj_rgb, int32_t *proj_idx, float *proj_mask,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * (6) Training-side operators (with nlr_grid_encode_backward above).
 *     nlr_composite_backward: gradient of compute_alpha_weights + volumetric_rendering (ZI/render.py:170-252) as
 *       `loss.backward()` gives it in ZI/train.py:459 - upstream gradients of the per-ray rgb [N,3], depth [N],
 *       semantic [N,K], intensity [N], acc [N] and of the weights [N,S] themselves (interlevel / distortion losses);
 *       any may be NULL (= zero).  Outputs: d_density [N,S] (required), d_rgb [3,N,S], d_semantic [K,N,S],
 *       d_intensity [N,S] (NULL to skip).  semantic / intensity are composited with detached weights
 *       (render.py:240-252), so their gradients do not reach the density.  Layouts as nlr_composite_level.
 *     nlr_hash_decay_forward / _backward: ZI/models.py:203-223 for ONE encoder:
 *       loss = mean over (level, channel) of the level's mean of embeddings^2.  forward writes the per-level sums of
 *       squares (double [L], device); the caller finishes loss = sum_l level_sumsq[l] / (rows_l * L * C).
 *       backward ACCUMULATES upstream * dloss/dembeddings into grad_embeddings.  offsets is the HOST table [L+1].
 * ------------------------------------------------------------------------------------------ */
int nlr_composite_backward(const float *density, const float *tdist, const float *directions, const float *rgb,
                           const float *semantic, const float *intensity, uint32_t N, uint32_t S, uint32_t class_num,
                           int opaque_background, float bg, const float *g_rgb, const float *g_depth,
                           const float *g_semantic, const float *g_intensity, const float *g_acc, const float *g_weights,
                           float *d_density, float *d_rgb, float *d_semantic, float *d_intensity, void *stream);
/* rows a-5 + a-6 as an operator: cast_rays + contract + /bound (ZI/render.py:129-168, ZI/coord.py:51-63, ZI/models.py:965-973).
 * means [N,S,n,3] (in [-1,1], what GridEncoder(bound=1) takes), stds [N,S,n].  No gradient: tdist is detached in training
 * (Model.stop_level_grad) and rays are data. */
int nlr_cast_contract(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                      float std_scale, const float *rand_deg, float *means, float *stds, void *stream);
int nlr_hash_decay_forward(const float *embeddings, const int32_t *offsets_host, uint32_t L, uint32_t C, double *level_sumsq,
                           void *stream);
int nlr_hash_decay_backward(const float *embeddings, const int32_t *offsets_host, uint32_t L, uint32_t C, float upstream,
                            float *grad_embeddings, void *stream);

/* ------------------------------------------------------------------------------------------
 * (6b) Fused NerfMLP for training: forward that also saves every layer's output, and the backward of what
 *     autograd does for the reference's Linear stack in `loss.backward()` (Z/train.py:272-281,459; ZI/models.py:1116-1251:
 *     density_layer, sem_layer, intensity_layer, lin_second_stage_* with the skip concat, rgb_layer; ReLU, softplus, softmax,
 *     sigmoid), both as one MFMA chain per 32 samples (bf16 operands, f32 accumulation: mixed-precision training).
 *     A plan fixes the shapes and the order of the flat parameter buffer: for each Linear in the order density_layer.0,
 *     density_layer.2, [sem_layer.0, sem_layer.2], [intensity_layer.0, intensity_layer.2], lin_second_stage_0..D-1, rgb_layer its
 *     weight (row-major [out, in]) then its bias; nlr_train_param_layout returns the offsets (2 per Linear).  nlr_train_pack
 *     re-packs the weight tapes on the device from that buffer (call it after every optimizer step).
 *     forward: features [M, F] f32 row-major, enc [M / S, 32] (pos_enc of the ray's viewdirs, zero padded); outputs in the
 *       layouts of nlr_mlp_level; acts [M, nlr_train_act_width()] bf16 = [hid 64 | bottleneck | head hidden | x_0 .. x_{D-1}].
 *     backward: upstream gradients in the layouts of the outputs (NULL = zero); gacts [M, act_width + 64] bf16 = the gradient of
 *       every pre-activation in the columns of acts, then [d head outputs 32 | d rgb_layer outputs 32]; d_features [M, F] f32.
 *       The weight gradients are GEMMs over these tensors, dW_l = gacts_l^T . acts_{l-1} (plain library GEMMs, host side).
 * ------------------------------------------------------------------------------------------ */
typedef struct NlrTrainPlan NlrTrainPlan;
int nlr_train_plan_create(uint32_t F, uint32_t W, uint32_t WB, uint32_t D, uint32_t deg_view, uint32_t class_num,
                          int use_semantic, int use_intensity, float density_bias, float rgb_premultiplier, float rgb_bias,
                          float rgb_padding, NlrTrainPlan **out, uint32_t *n_params);
void nlr_train_plan_destroy(NlrTrainPlan *p);
uint32_t nlr_train_act_width(const NlrTrainPlan *p);
int nlr_train_param_layout(const NlrTrainPlan *p, uint32_t *offsets, uint32_t capacity);
int nlr_train_pack(NlrTrainPlan *p, const float *params_dev, void *stream);
int nlr_mlp_train_forward(const NlrTrainPlan *p, const float *features, const float *enc, uint32_t M, uint32_t S, float *density,
                          float *rgb, float *semantic, float *intensity, void *acts, void *stream);
int nlr_mlp_train_backward(const NlrTrainPlan *p, uint32_t M, uint32_t S, const float *density, const float *rgb,
                           const float *semantic, const void *acts, const float *g_density, const float *g_rgb,
                           const float *g_semantic, const float *g_intensity, void *gacts, float *d_features, void *stream);

/* ------------------------------------------------------------------------------------------
 * (7) Dynamic-object branch: owner of every sample.  winner [N,S] int32 = index of the LAST
 *     track whose box contains the sample's interval midpoint (ZI/models.py:415,475 let later tracks overwrite
 *     earlier ones; ZI/obj_utils.py:203-216 inside test), -1 outside every box.  box_params [N, n_obj, 8] =
 *     (cos theta_z, sin theta_z, t_w_o xyz, scale xyz) per ray and track, with t_w_o = rotate_yaw_z(-center, theta_z)
 *     and scale = 1 / (wlh / 2 + 1e-9) computed by the caller with the reference's expressions
 *     (obj_utils.py:5-28,76-113,158-170).
 * ------------------------------------------------------------------------------------------ */
int nlr_box_winner(const float *tdist, const float *origins, const float *directions, const float *box_params, uint32_t N,
                   uint32_t S, uint32_t n_obj, int32_t *winner, void *stream);

/* (7b) The whole branch on the device, no host synchronisation anywhere (Config.instance_obj = True in latent mode, the shipped
 *      gin: one ObjMLP per class + one latent code per track, ZI/models.py:125-142,401-477).
 *
 * nlr_track_box_params <-> get_pose (ZI/obj_utils.py:431-475) + the per-(ray, track) constants of world2object
 *      (obj_utils.py:116-176): blend of the two recorded poses closest in time to the ray's timestamp, then
 *      (cos theta, sin theta, t_w_o, scale) as nlr_box_winner takes them.  tracks dev f32 [n_obj, T, 9] =
 *      (center3, theta_z, wlh3, timestamp, track id), T >= 2; timestamps dev f32 [N]; box_params dev f32 [N, n_obj, 8].
 *
 * NlrObjects: the packed object networks (owned, like NlrModel).  Each class: an L x C hash grid on box coordinates (bound 1,
 *      no erf re-weighting), density_layer (L*C + shape half of the latent -> 64 -> bottleneck <= 64), view MLP of width <= 32
 *      on [bottleneck | pos_enc(box-frame view direction, deg_view) | texture half of the latent] with the skip concatenation
 *      after layer `skip_layer_dir`, rgb layer; semantic = one-hot of `class_type` (fixed_semantic; 255 = all zeros).
 *      NlrMlpDesc carries grid, density0/2, view[], rgb_layer and the scalars; its sem/intensity members are ignored.
 *
 * nlr_objects_apply: winner (see nlr_box_winner) -> per-class lists of the owned samples (device-side compaction) -> the
 *      class's network on them -> results written over density [N,S], rgb [3,N,S] and semantic [K,N,S] in place (rgb and
 *      semantic may be NULL: proposal levels replace the density only, models.py:466-477).  workspace: dev, at least
 *      nlr_objects_workspace_bytes(o, N, S) bytes.  winner_out [N,S] int32 optional.
 *
 * nlr_render_rays_dynamic: nlr_render_rays with the object merge between the MLP and the compositing of every level.
 *      `winner` (optional) receives one [N, S_l] int32 owner map per level (ray_history's obj_mask = winner >= 0).
 *      workspace: nlr_workspace_bytes(m, N) + nlr_objects_workspace_bytes(o, N, max S). */
int nlr_track_box_params(const float *tracks, const float *timestamps, uint32_t N, uint32_t n_obj, uint32_t T, float *box_params,
                         void *stream);

typedef struct NlrObjClassDesc {
    NlrMlpDesc mlp;
    uint32_t latent_size, split_latent;  /* Config.latent_size (0 = none), MLP.split_latent (models.py:881-885,924-925) */
    int32_t class_type;                  /* label of the fixed one-hot semantic, 255 = none (models.py:1124-1130) */
} NlrObjClassDesc;

typedef struct NlrObjectsDesc {
    uint32_t n_classes;
    const NlrObjClassDesc *classes;
    uint32_t n_tracks;
    const int32_t *track_class;          /* HOST [n_tracks]: index into `classes` */
    const float *latents;                /* HOST f32 [n_tracks, latent_size] (latent_vector_dict), NULL when latent_size = 0 */
} NlrObjectsDesc;

typedef struct NlrObjects NlrObjects;
int nlr_objects_create(const NlrObjectsDesc *desc, NlrObjects **out, void *stream);
void nlr_objects_destroy(NlrObjects *o);
size_t nlr_objects_workspace_bytes(const NlrObjects *o, uint32_t N, uint32_t S);
int nlr_objects_apply(const NlrObjects *o, const NlrRays *rays, const float *tdist, const float *box_params, uint32_t N, uint32_t S,
                      uint32_t n_obj, float *density, float *rgb, float *semantic, uint32_t K, int32_t *winner_out, void *workspace,
                      size_t workspace_bytes, void *stream);
int nlr_render_rays_dynamic(const NlrModel *m, const NlrObjects *o, const NlrRays *rays, const float *box_params, uint32_t n_obj,
                            uint32_t N, const NlrRenderCfg *cfg, const NlrOut *out, int32_t *const *winner, void *workspace,
                            size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * (8) PropMLP density network for training (ZI/models.py:887-889,996-997 with disable_rgb = True):
 *     raw [M] = W2 relu(W1 f + b1) + b2, f = [M, F] grid features (F = L*C <= 16), nn.Linear layouts W1 [64, F], W2 [1, 64];
 *     all pointers dev f32 (b2 is the 1-element bias tensor).  The backward recomputes the hidden units (only the features are
 *     kept between the passes), writes d_feat [M, F] (may be NULL) and OVERWRITES d_w1 [64,F], d_b1 [64], d_w2 [64], d_b2 [1].
 * ------------------------------------------------------------------------------------------ */
int nlr_prop_mlp_forward(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, uint32_t M, uint32_t F,
                         float *raw, void *stream);
int nlr_prop_mlp_backward(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, const float *g_raw,
                          uint32_t M, uint32_t F, float *d_feat, float *d_w1, float *d_b1, float *d_w2, float *d_b2, void *stream);

/* ------------------------------------------------------------------------------------------
 * (9) The front half of MLP.predict_density as ONE differentiable operator (ZI/models.py:965-979): cast_rays (render.py:129-168),
 *     contraction (coord.py:51-100), GridEncoder, erf re-weighting and the mean over the multisamples.
 *     forward:  features [N*S, L*C] f32 from the intervals of tdist [N, S+1] (what nlr_mlp_level feeds its MLP).
 *     backward: grad_table [sO, C] f32 += d features / d table applied to d_features [N*S, L*C] (accumulates: zero it first).
 *     The forward materialises nothing of size [N, S, sample_n, L, C]; the backward one such tensor (the per-point feature
 *     gradients it hands to nlr_grid_encode_backward) in caller-provided scratch.  The sample positions carry no gradient
 *     (Model.stop_level_grad); rand_deg as in NlrRenderCfg; f32 grad table.
 * ------------------------------------------------------------------------------------------ */
int nlr_encode_features_forward(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights, float *features,
                                void *stream);
int nlr_encode_features_backward(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                 float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights,
                                 const float *d_features, float *points_tmp /* [N*S*sample_n, 3] scratch */,
                                 float *grad_tmp /* [N*S*sample_n, L*C] scratch */, float *grad_table, void *stream);
/* ... with the binned scatter's workspace (nlr_grid_backward_workspace_bytes for B = N*S*sample_n points), see nlr_grid_encode_backward_ws */
int nlr_encode_features_backward_ws(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                    float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights,
                                    const float *d_features, float *points_tmp, float *grad_tmp, float *grad_table, void *workspace,
                                    size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFLIDAR_HIP_H */
