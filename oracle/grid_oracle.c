/*
 * oracle/grid_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker, never shipped, never measured
 * as the product).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 *
 * Plain-C restatement of the reference's only native operator, the multi-resolution hash grid
 * encoder.  The reference implementation is CUDA-only and cannot be compiled or run here (no
 * nvcc, no NVIDIA device), so this file is a restatement of the published algorithm, reviewed
 * line by line against
 *   NeRF_LiDAR/zipnerf/gridencoder/src/gridencoder.cu:50-63    fast_hash
 *   NeRF_LiDAR/zipnerf/gridencoder/src/gridencoder.cu:66-84    get_grid_index
 *   NeRF_LiDAR/zipnerf/gridencoder/src/gridencoder.cu:87-245   kernel_grid (forward + dy_dx)
 *   NeRF_LiDAR/zipnerf/gridencoder/src/gridencoder.cu:248-340  kernel_grid_backward
 *   NeRF_LiDAR/zipnerf/gridencoder/src/gridencoder.cu:343-369  kernel_input_backward
 *   NeRF_LiDAR/zipnerf/gridencoder/src/gridencoder.cu:506-601  kernel_grad_tv
 * Parity status for this file alone: "parity unpinned" against the CUDA binary (it cannot
 * run); it is pinned indirectly through the reference's Python call sites (grid.py:158-174,
 * models.py:974-979), which consume its output in every whole-forward golden fixture.
 *
 * Numerics notes that matter for bit-exactness against the HIP kernel:
 *  - nvcc contracts `x*scale + 0.5f` and `results += w*g` into FMAs; both are written as
 *    fmaf() here and in the HIP kernel so the two agree bit for bit.
 *  - the per-level scale `exp2f(level*S)*H - 1` (gridencoder.cu:138) is computed once on the
 *    host (nlr_oracle_level_scale) and the HIP launcher uses the same host function, so a
 *    1-ulp libm difference in exp2f can never split the two.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define NLR_MAX_D 3

static inline uint32_t fast_hash3(const uint32_t *p, uint32_t D) {
    /* gridencoder.cu:54 -- primes[0] = 1 keeps x-neighbours adjacent in memory */
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                       2097192037u, 1434869437u, 2165219737u};
    uint32_t r = 0;
    for (uint32_t i = 0; i < D; ++i) r ^= p[i] * primes[i];
    return r;
}

static inline uint32_t grid_index(uint32_t gridtype, int align_corners, uint32_t D, uint32_t C,
                                  uint32_t hashmap_size, uint32_t resolution, const uint32_t *p) {
    /* gridencoder.cu:66-84: dense stride walk stops as soon as stride exceeds the level size;
       hashed levels then hash the FULL integer coordinate and discard the partial index. */
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += p[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash3(p, D);
    return (index % hashmap_size) * C;
}

/* gridencoder.cu:138-139.  S = log2(per_level_scale), H = base resolution. */
void nlr_oracle_level_scale(uint32_t L, float S, uint32_t H, float *scale, uint32_t *resolution) {
    for (uint32_t l = 0; l < L; ++l) {
        float sc = exp2f((float)l * S) * (float)H - 1.0f;
        scale[l] = sc;
        resolution[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

/*
 * inputs  [B, D] f32 in [0,1] (points outside -> zeros, gridencoder.cu:110-135)
 * table   [sO, C] f32
 * offsets [L+1] i32
 * outputs [L, B, C] f32 (level-major, as the reference kernel writes it)
 * dy_dx   [B, L*D*C] f32 or NULL
 */
void nlr_oracle_grid_forward(const float *inputs, const float *table, const int32_t *offsets,
                             float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             float S, uint32_t H, float *dy_dx, uint32_t gridtype,
                             int align_corners, uint32_t interp) {
    float scale_l[32];
    uint32_t res_l[32];
    nlr_oracle_level_scale(L, S, H, scale_l, res_l);
#pragma omp parallel for schedule(static)
    for (int64_t bl = 0; bl < (int64_t)B * L; ++bl) {
        const uint32_t level = (uint32_t)(bl / B);
        const uint32_t b = (uint32_t)(bl % B);
        const float *x = inputs + (size_t)b * D;
        const float *grid = table + (size_t)(uint32_t)offsets[level] * C;
        float *out = outputs + ((size_t)level * B + b) * C;
        float *dd = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : 0;

        int oob = 0;
        for (uint32_t d = 0; d < D; d++)
            if (x[d] < 0 || x[d] > 1) oob = 1;
        if (oob) {
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
            if (dd)
                for (uint32_t i = 0; i < D * C; i++) dd[i] = 0;
            continue;
        }
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = scale_l[level];
        const uint32_t resolution = res_l[level];

        float pos[NLR_MAX_D], pos_deriv[NLR_MAX_D];
        uint32_t pos_grid[NLR_MAX_D];
        for (uint32_t d = 0; d < D; d++) {
            pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
            pos_grid[d] = (uint32_t)floorf(pos[d]);
            pos[d] -= (float)pos_grid[d];
            if (interp == 1) {
                pos_deriv[d] = 6 * pos[d] * (1.0f - pos[d]);
                pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            } else {
                pos_deriv[d] = 1.0f;
            }
        }
        float results[8] = {0};
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1;
            uint32_t pl[NLR_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) {
                    w *= 1 - pos[d];
                    pl[d] = pos_grid[d];
                } else {
                    w *= pos[d];
                    pl[d] = pos_grid[d] + 1;
                }
            }
            uint32_t index = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pl);
            for (uint32_t ch = 0; ch < C; ch++) results[ch] = fmaf(w, grid[index + ch], results[ch]);
        }
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];

        if (dd) {
            for (uint32_t gd = 0; gd < D; gd++) {
                float rg[8] = {0};
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float w = scale;
                    uint32_t pl[NLR_MAX_D];
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1u << nd)) == 0) {
                            w *= 1 - pos[d];
                            pl[d] = pos_grid[d];
                        } else {
                            w *= pos[d];
                            pl[d] = pos_grid[d] + 1;
                        }
                    }
                    pl[gd] = pos_grid[gd];
                    uint32_t il = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pl);
                    pl[gd] = pos_grid[gd] + 1;
                    uint32_t ir = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pl);
                    for (uint32_t ch = 0; ch < C; ch++)
                        rg[ch] += w * (grid[ir + ch] - grid[il + ch]) * pos_deriv[gd];
                }
                for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = rg[ch];
            }
        }
    }
}

/*
 * Backward (gridencoder.cu:248-369).  grad [L,B,C]; grad_table [sO,C] must be zero-initialised
 * by the caller (grid.py:77); accumulation here is sequential in b (deterministic), whereas the
 * reference uses atomicAdd in arbitrary order, so only a tolerance comparison is meaningful.
 */
void nlr_oracle_grid_backward(const float *grad, const float *inputs, const int32_t *offsets,
                              float *grad_table, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                              float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                              uint32_t gridtype, int align_corners, uint32_t interp) {
    float scale_l[32];
    uint32_t res_l[32];
    nlr_oracle_level_scale(L, S, H, scale_l, res_l);
#pragma omp parallel for schedule(static)
    for (int32_t level = 0; level < (int32_t)L; ++level) {
        float *gg = grad_table + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = scale_l[level];
        const uint32_t resolution = res_l[level];
        for (uint32_t b = 0; b < B; ++b) {
            const float *x = inputs + (size_t)b * D;
            const float *g = grad + ((size_t)level * B + b) * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            float pos[NLR_MAX_D];
            uint32_t pos_grid[NLR_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pl[NLR_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1 - pos[d];
                        pl[d] = pos_grid[d];
                    } else {
                        w *= pos[d];
                        pl[d] = pos_grid[d] + 1;
                    }
                }
                uint32_t index = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += w * g[ch];
            }
        }
    }
    if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; ++t) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t % D);
            const float *dd = dy_dx + (size_t)b * L * D * C;
            float r = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    r += grad[((size_t)l * B + b) * C + ch] * dd[l * D * C + d * C + ch];
            grad_inputs[t] = r;
        }
    }
}

/*
 * Total-variation gradient (gridencoder.cu:506-601, bound to grid.py:176-198).  For every point and level: the cell corner
 * pos_grid = floor(x * scale + 0.5); over the 2*D axis neighbours that exist (right when pos_grid[d] < resolution, left when
 * pos_grid[d] > 0): results += g, idelta += g*g with g = table[corner] - table[neighbour]; then
 * grad[corner] += (weight / (2*D)) * results * rsqrt(idelta + 1e-9).  inputs [B, D] f32 already in [0,1] (points outside are
 * skipped); grad [sO, C] is accumulated into (the caller's embeddings.grad).  Sequential in b here, atomicAdd in arbitrary order
 * in the reference: tolerance comparison only.  `g*g + idelta` is one FMA under nvcc's default contraction; rsqrtf is the
 * reference's 2-ulp intrinsic, 1/sqrtf here.
 */
void nlr_oracle_grid_tv(const float *inputs, const float *table, float *grad, const int32_t *offsets, float weight, uint32_t B,
                        uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners) {
    float scale_l[32];
    uint32_t res_l[32];
    nlr_oracle_level_scale(L, S, H, scale_l, res_l);
    const float w = weight / (float)(2 * D);
    for (uint32_t level = 0; level < L; ++level) {
        const float *tb = table + (size_t)(uint32_t)offsets[level] * C;
        float *gg = grad + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = scale_l[level];
        const uint32_t resolution = res_l[level];
        for (uint32_t b = 0; b < B; ++b) {
            const float *x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            uint32_t pos_grid[NLR_MAX_D];
            for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
            float results[8] = {0}, idelta[8] = {0};
            const uint32_t index = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pos_grid);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur = pos_grid[d];
                if (cur < resolution) {
                    pos_grid[d] = cur + 1;
                    const uint32_t ir = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float g = tb[index + ch] - tb[ir + ch];
                        results[ch] += g;
                        idelta[ch] = fmaf(g, g, idelta[ch]);
                    }
                }
                if (cur > 0) {
                    pos_grid[d] = cur - 1;
                    const uint32_t il = grid_index(gridtype, align_corners, D, C, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float g = tb[index + ch] - tb[il + ch];
                        results[ch] += g;
                        idelta[ch] = fmaf(g, g, idelta[ch]);
                    }
                }
                pos_grid[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += (w * results[ch]) * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
}
