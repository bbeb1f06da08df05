// Multi-resolution hash-grid operator for gfx950: forward (+dy_dx) and backward.
//
// Replaces the reference's only native code, gridencoder/src/gridencoder.cu (kernel_grid :87-245,
// kernel_grid_backward :248-340, kernel_input_backward :343-369) behind the C ABI declared in
// include/nerflidar_hip.h.  Written for CDNA4, not translated:
//   * per-level constants (offset, size, scale, resolution) are computed once on the host and
//     travel as kernel arguments (SGPRs) instead of being re-derived per thread from a device
//     `offsets` tensor with exp2f/ceil;
//   * one thread owns one (point, level) and issues all 8 corner gathers as independent
//     16-byte (C=4) loads before the first use, so a 64-lane wave keeps 512 gathers in flight;
//   * blockIdx.y = level keeps a level's table resident in the XCD L2 / Infinity Cache while
//     the grid's x-dimension streams over points (level-major dispatch order);
//   * launched on the caller's stream (the reference uses the legacy default stream).
// Arithmetic order (fmaf for pos and for the corner accumulation) matches oracle/grid_oracle.c
// so the forward is bit-exact against the CPU checker.
#include "nlr_common.h"
#include "nlr_grid_level.h"

#include <hip/hip_fp16.h>

void nlr_level_scale(uint32_t L, float S, uint32_t H, float *scale, uint32_t *resolution) {
    for (uint32_t l = 0; l < L; ++l) {
        float sc = exp2f((float)l * S) * (float)H - 1.0f;
        scale[l] = sc;
        resolution[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

int nlr_fill_grid_params(GridParams *gp, const void *table, int table_dtype, const int32_t *offsets_host,
                         uint32_t L, uint32_t C, float S, uint32_t H, uint32_t gridtype, int align_corners,
                         uint32_t interp) {
    NLR_CHECK_ARG(L >= 1 && L <= NLR_MAX_GRID_LEVELS, "grid: num_levels %u outside [1,%d]", L, NLR_MAX_GRID_LEVELS);
    NLR_CHECK_ARG(C == 1 || C == 2 || C == 4 || C == 8, "GridEncoding: C must be 1, 2, 4, or 8 (got %u)", C);
    NLR_CHECK_ARG(table_dtype == 0 || table_dtype == 1, "grid: table_dtype must be 0 (f32) or 1 (f16)");
    NLR_CHECK_ARG(offsets_host != nullptr, "grid: offsets (host) is NULL");
    memset(gp, 0, sizeof(*gp));
    gp->table = table;
    gp->table_dtype = table_dtype;
    gp->L = L;
    gp->C = C;
    gp->gridtype = gridtype;
    gp->align_corners = align_corners ? 1u : 0u;
    gp->interp = interp;
    nlr_level_scale(L, S, H, gp->scale, gp->res);
    for (uint32_t l = 0; l < L; ++l) {
        NLR_CHECK_ARG(offsets_host[l + 1] > offsets_host[l], "grid: offsets must be increasing (level %u)", l);
        gp->offset[l] = (uint32_t)offsets_host[l];
        gp->hsize[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        // grid_sizes buffer of grid.py:128-129,142: ceil(H * pls^l) (+1 unless align_corners)
        double r = ceil((double)H * exp2((double)l * (double)S));
        gp->gsize[l] = (float)((int)r + (align_corners ? 0 : 1));
        gp->inv_gsize[l] = 1.0f / gp->gsize[l];
        uint64_t step = align_corners ? gp->res[l] : gp->res[l] + 1, stride = 1;
        int dense = 1;
        for (int d = 0; d < 3; ++d) {
            if (stride > gp->hsize[l]) break;
            stride *= step;
        }
        if (gridtype == 0 && stride > gp->hsize[l]) dense = 0;
        gp->dense[l] = dense;
        gp->step[l] = (uint32_t)step;
        const uint32_t hs = gp->hsize[l];
        const bool all3 = (step <= hs) && (step * step <= hs);  // the stride walk adds all three coordinates
        if (all3 && step * step * step <= hs) gp->mode[l] = 0;                 // dense, index < hsize
        else if (gridtype == 0 && (hs & (hs - 1)) == 0) gp->mode[l] = 1;       // hashed, power-of-two table
        else gp->mode[l] = 2;
    }
    return NLR_OK;
}

template <typename T>
__device__ __forceinline__ float ld(const T *p);
template <>
__device__ __forceinline__ float ld<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float ld<__half>(const __half *p) { return __half2float(*p); }

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <typename T, int C, bool DYDX>
__global__ void __launch_bounds__(256) nlr_grid_fwd_kernel(const float *__restrict__ x, GridParams gp, float *__restrict__ out,
                                                           float *__restrict__ dy_dx, uint32_t B, int out_layout) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const uint32_t L = gp.L;
    const T *grid = (const T *)gp.table + (size_t)gp.offset[level] * C;
    float *o = out_layout == 0 ? out + ((size_t)level * B + b) * C : out + (size_t)b * L * C + level * C;
    float *dd = DYDX ? dy_dx + (size_t)b * 3 * L * C + (size_t)level * 3 * C : nullptr;

    const float x0 = x[(size_t)b * 3 + 0], x1 = x[(size_t)b * 3 + 1], x2 = x[(size_t)b * 3 + 2];
    // `inputs[d] < 0 || inputs[d] > 1` (cu:114): NaN compares false on both sides, i.e. in range.
    const bool oob = (x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1);
    if (oob) {
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = 0.0f;
        if (DYDX)
            for (int i = 0; i < 3 * C; ++i) dd[i] = 0.0f;
        return;
    }
    const uint32_t hsize = gp.hsize[level], res = gp.res[level];
    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    float pos[3] = {fmaf(x0, scale, half), fmaf(x1, scale, half), fmaf(x2, scale, half)};
    uint32_t pg[3];
    float pd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
        if (gp.interp == 1) {
            pd[d] = 6 * pos[d] * (1.0f - pos[d]);
            pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
        } else {
            pd[d] = 1.0f;
        }
    }
    uint32_t idx[8];
    float w[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        float ww = 1.0f;
        uint32_t pl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((c8 >> d) & 1) {
                ww *= pos[d];
                pl[d] = pg[d] + 1;
            } else {
                ww *= 1 - pos[d];
                pl[d] = pg[d];
            }
        }
        w[c8] = ww;
        idx[c8] = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pl[0], pl[1], pl[2]) * C;
    }
    float g[8][C];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
        for (int c = 0; c < C; ++c) g[c8][c] = ld<T>(grid + idx[c8] + c);
    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = 0.0f;
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
        for (int c = 0; c < C; ++c) r[c] = fmaf(w[c8], g[c8][c], r[c]);
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = r[c];

    if (DYDX) {
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float rg[C];
#pragma unroll
            for (int c = 0; c < C; ++c) rg[c] = 0.0f;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                float ww = scale;
                uint32_t pl[3];
#pragma unroll
                for (int nd = 0; nd < 2; ++nd) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((i4 >> nd) & 1) {
                        ww *= pos[d];
                        pl[d] = pg[d] + 1;
                    } else {
                        ww *= 1 - pos[d];
                        pl[d] = pg[d];
                    }
                }
                pl[gd] = pg[gd];
                const uint32_t il = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pl[0], pl[1], pl[2]) * C;
                pl[gd] = pg[gd] + 1;
                const uint32_t ir = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pl[0], pl[1], pl[2]) * C;
#pragma unroll
                for (int c = 0; c < C; ++c) rg[c] += ww * (ld<T>(grid + ir + c) - ld<T>(grid + il + c)) * pd[gd];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) dd[gd * C + c] = rg[c];
        }
    }
}

template <typename T, int C>
static int launch_fwd(const float *x, const GridParams &gp, float *out, float *dy_dx, uint32_t B, int out_layout,
                      hipStream_t st) {
    dim3 grid((B + 255) / 256, gp.L), block(256);
    if (dy_dx)
        hipLaunchKernelGGL((nlr_grid_fwd_kernel<T, C, true>), grid, block, 0, st, x, gp, out, dy_dx, B, out_layout);
    else
        hipLaunchKernelGGL((nlr_grid_fwd_kernel<T, C, false>), grid, block, 0, st, x, gp, out, dy_dx, B, out_layout);
    NLR_LAUNCH_CHECK("nlr_grid_fwd_kernel");
    return NLR_OK;
}

extern "C" int nlr_grid_encode_forward(const float *inputs, const void *embeddings, int table_dtype,
                                       const int32_t *offsets_host, float *outputs, uint32_t B, uint32_t D,
                                       uint32_t C, uint32_t L, float S, uint32_t H, float *dy_dx, uint32_t gridtype,
                                       int align_corners, uint32_t interp, int out_layout, void *stream) {
    NLR_CHECK_ARG(D == 3, "GridEncoding: this build supports input_dim D = 3 only (got %u)", D);
    NLR_CHECK_ARG(inputs && embeddings && outputs, "grid_encode_forward: NULL tensor");
    NLR_CHECK_ARG(out_layout == 0 || out_layout == 1, "grid_encode_forward: out_layout must be 0 or 1");
    NLR_CHECK_ARG(gridtype <= 1 && interp <= 1, "grid_encode_forward: gridtype/interp out of range");
    if (B == 0) return NLR_OK;
    GridParams gp;
    int rc = nlr_fill_grid_params(&gp, embeddings, table_dtype, offsets_host, L, C, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
#define NLR_DISPATCH_C(T)                                                          \
    switch (C) {                                                                   \
        case 1: return launch_fwd<T, 1>(inputs, gp, outputs, dy_dx, B, out_layout, st); \
        case 2: return launch_fwd<T, 2>(inputs, gp, outputs, dy_dx, B, out_layout, st); \
        case 4: return launch_fwd<T, 4>(inputs, gp, outputs, dy_dx, B, out_layout, st); \
        default: return launch_fwd<T, 8>(inputs, gp, outputs, dy_dx, B, out_layout, st); \
    }
    if (table_dtype == 0) { NLR_DISPATCH_C(float) } else { NLR_DISPATCH_C(__half) }
#undef NLR_DISPATCH_C
}

// Levels that nlr_grid_bwd_lds_kernel (below) accumulates in LDS when the batch is large; every kernel and the bin plan decide with this
// one predicate.  1: dense level whose whole table fits an LDS copy; 2: a coarse level beyond that (cells of 1/128 of the cube or larger:
// a batch of LiDAR rays visits a few hundred to a few thousand of its entries), accumulated in a tagged LDS cache; 0: scattered.
#define NLR_LDS_TABLE_FLOATS 36864  // 144 KiB
#define NLR_LDS_CACHE_MAX_STEP 129u
__host__ __device__ __forceinline__ int nlr_level_lds_kind(const GridParams &gp, uint32_t level, uint32_t C, int no_cache) {
    if (gp.mode[level] == 0 && gp.hsize[level] * C <= NLR_LDS_TABLE_FLOATS) return 1;
    if (!no_cache && C <= 4 && gp.mode[level] <= 1 && gp.step[level] <= NLR_LDS_CACHE_MAX_STEP) return 2;
    return 0;
}
// The 8 corner indices of a cell with the level's mode resolved once (wave-uniform branch; nlr_common.h:nlr_corner_idx): the generic
// nlr_grid_index pays a 32-bit modulo per corner, which made the LDS and bin kernels below issue-bound.
__device__ __forceinline__ void nlr_corner_idx_any(const GridParams &gp, uint32_t level, const uint32_t (&pg)[3], uint32_t (&idx)[8]) {
    const uint32_t mode = gp.mode[level];
    if (mode == 0) nlr_corner_idx<0>(gp, level, pg, idx);
    else if (mode == 1) nlr_corner_idx<1>(gp, level, pg, idx);
    else nlr_corner_idx<2>(gp, level, pg, idx);
}
// One lane per (point, channel): the C channel atomics of a corner go out in ONE instruction as C adjacent lanes on C
// consecutive floats, so a 64-lane atomic touches 64/C table entries instead of 64 - the L2 atomic path is paid per
// distinct line (the per-point form spent C instructions of 64 scattered lines each on the same bytes).
template <int C>
__global__ void __launch_bounds__(256) nlr_grid_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ x,
                                                           GridParams gp, float *__restrict__ grad_table, uint32_t B,
                                                           int grad_layout, uint64_t level_list) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // channel-major inside the wave: a wave owns P = 64 / C consecutive points, lane = ch * P + point.  Neighbouring lanes are
    // then neighbouring points of ONE channel (runs of equal address for the run aggregation below), and the C channels of a
    // cell still leave in one instruction on C consecutive floats.
    constexpr uint32_t P = 64 / C;
    const uint32_t b0 = (t >> 6) * P + ((uint32_t)lane % P), ch = (uint32_t)lane / P;
    const uint32_t level = (uint32_t)(level_list >> (4 * blockIdx.y)) & 15u;  // the levels nlr_grid_bwd_lds_kernel does not take
    const bool inb = b0 < B;
    // (no early exit per lane: the run scan reads its neighbours through DPP, inactive lanes would read as zero keys)
    const uint32_t b = inb ? b0 : B - 1;
    const float x0 = x[(size_t)b * 3 + 0], x1 = x[(size_t)b * 3 + 1], x2 = x[(size_t)b * 3 + 2];
    const bool valid = inb && !((x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1));
    if (__ballot(valid) == 0ull) return;  // wave-uniform
    const float gc = grad_layout == 0 ? grad[((size_t)level * B + b) * C + ch] : grad[(size_t)b * gp.L * C + level * C + ch];
    float *gt = grad_table + (size_t)gp.offset[level] * C;
    const uint32_t hsize = gp.hsize[level], res = gp.res[level];
    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    // out-of-range lanes of an aggregating wave walk along with a harmless in-range position and valid = false
    float pos[3] = {fmaf(valid ? x0 : 0.5f, scale, half), fmaf(valid ? x1 : 0.5f, scale, half), fmaf(valid ? x2 : 0.5f, scale, half)};
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
        if (gp.interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
    }
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        float ww = 1.0f;
        uint32_t pl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((c8 >> d) & 1) {
                ww *= pos[d];
                pl[d] = pg[d] + 1;
            } else {
                ww *= 1 - pos[d];
                pl[d] = pg[d];
            }
        }
        const uint32_t addr = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pl[0], pl[1], pl[2]) * C + ch;
        nlr_run_atomic<1>(gt, addr, ww * gc, valid, lane);
    }
}

// The same scatter with BOTH x-corners of a cell edge in one instruction (round 4).  What an atomic costs on this chip
// (scripts/micro/atomic_scope.hip, profiles/r04_atomic_microbench.txt): 21 G (instruction x distinct 64-byte line) pairs per second
// chip-wide, whatever the table size (64 KiB or 32 MiB), the number of busy CUs (64 or 256), the scope, or the number of lanes that add
// to DIFFERENT floats of the line (16 are as cheap as 1); lanes on the SAME float serialise at about the same price each.  So the
// quantity to shrink is lines per instruction.  The corners (x, y, z) and (x + 1, y, z) are adjacent entries on a dense level and,
// on a hashed level with even x, the entries `i` and `i ^ 1` (the x coordinate enters the hash with the prime 1, gridencoder.cu:66-84):
// one line for both, three times out of four / every second point.  A wave here owns P = 32 / C points; lane = ch * 2P + xc * P + point,
// so that the two x-corners of a point leave in one instruction and a 16-lane row still holds consecutive points of one channel
// and one x-corner for the run aggregation.  Four atomic instructions per point and level instead of eight; same sums.
template <int C>
__global__ void __launch_bounds__(256) nlr_grid_bwd_xpair_kernel(const float *__restrict__ grad, const float *__restrict__ x,
                                                                 GridParams gp, float *__restrict__ grad_table, uint32_t B,
                                                                 int grad_layout, uint64_t level_list, uint32_t level_mask) {
    static_assert(C == 1 || C == 2 || C == 4, "x-pair scatter: C in {1, 2, 4}");
    const uint32_t level = (uint32_t)(level_list >> (4 * blockIdx.y)) & 15u;
    if (level_mask && !((level_mask >> level) & 1u)) return;  // NLR_DBG_SCATTER_LEVELS: per-level timing
    constexpr uint32_t P = 32 / C;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t ch = (uint32_t)lane / (2 * P), xc = ((uint32_t)lane / P) & 1u, b0 = w * P + (uint32_t)lane % P;
    const bool inb = b0 < B;
    const uint32_t b = inb ? b0 : B - 1;
    const float x0 = x[(size_t)b * 3 + 0], x1 = x[(size_t)b * 3 + 1], x2 = x[(size_t)b * 3 + 2];
    const bool valid = inb && !((x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1));
    if (__ballot(valid) == 0ull) return;  // wave-uniform
    const float gc = grad_layout == 0 ? grad[((size_t)level * B + b) * C + ch] : grad[(size_t)b * gp.L * C + level * C + ch];
    float *gt = grad_table + (size_t)gp.offset[level] * C;
    const uint32_t hsize = gp.hsize[level], res = gp.res[level];
    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    float pos[3] = {fmaf(valid ? x0 : 0.5f, scale, half), fmaf(valid ? x1 : 0.5f, scale, half), fmaf(valid ? x2 : 0.5f, scale, half)};
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
        if (gp.interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
    }
    const float wx = xc ? pos[0] : 1 - pos[0];   // the weight products keep the order of the 8-corner loop: (wx * wy) * wz
    const uint32_t px = pg[0] + xc;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        const float wy = (c4 & 1) ? pos[1] : 1 - pos[1], wz = (c4 & 2) ? pos[2] : 1 - pos[2];
        const uint32_t addr = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, px, pg[1] + (c4 & 1), pg[2] + (c4 >> 1)) * C + ch;
        nlr_run_atomic<1>(gt, addr, ((1.0f * wx) * wy) * wz * gc, valid, lane);
    }
}

// Coarse levels in LDS.  Every sample of a batch lands in the same few thousand cells of the coarse levels, and global atomics on one
// line queue behind each other (~40 cycles each at the L2: the 33^3 level of the C = 4 grid alone kept the round-3 scatter busy for
// 11.6 ms of a 14.7 M-point backward, profiles/r04_scatter_levels.txt).  A workgroup strides over the batch for ONE level and adds into LDS
// (ds_add_f32), then adds its LDS image to the table once:
//   kind 1 (direct): the level's whole table is the image (17^3 x C <= 4, 33^3 x 1);
//   kind 2 (cache):  NS slots of C floats with a tag each; an entry claims the slot its hash names (ds_cmpst on the tag) and keeps it to
//     the end of the workgroup; an entry that finds its slot taken by another goes to the table with a global atomic as before.  Correct
//     for any occupancy; it pays when the batch visits fewer distinct entries than there are slots (levels up to 128^3 cells).
// A wave is laid out channel-major as in the scatter kernels (lane = ch * P + point, P = 64 / C) and merges runs of equal addresses inside
// a 16-lane row first (nlr_run_merge): neighbouring lanes are neighbouring points of a ray, 16 lanes adding to one LDS word serialise.
template <int C>
__global__ void __launch_bounds__(1024) nlr_grid_bwd_lds_kernel(const float *__restrict__ grad, const float *__restrict__ x, GridParams gp,
                                                               float *__restrict__ grad_table, uint32_t B, int grad_layout, uint32_t direct_mask,
                                                               uint32_t cache_mask) {
    __shared__ float acc[NLR_LDS_TABLE_FLOATS];
    const uint32_t level = blockIdx.y;
    const bool direct = (direct_mask >> level) & 1u;
    if (!direct && !((cache_mask >> level) & 1u)) return;  // (wave-uniform: the whole workgroup)
    constexpr uint32_t NS = NLR_LDS_TABLE_FLOATS / (C + 1);  // cache: NS x C values, then NS tags
    uint32_t *tags = (uint32_t *)(acc + NS * C);
    const uint32_t cells = direct ? gp.hsize[level] * C : NS * C;
    for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x) acc[i] = 0.0f;
    if (!direct)
        for (uint32_t i = threadIdx.x; i < NS; i += blockDim.x) tags[i] = 0xffffffffu;
    __syncthreads();
    float *gt = grad_table + (size_t)gp.offset[level] * C;

    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    constexpr uint32_t P = 64 / C;
    const int lane = threadIdx.x & 63;
    const uint32_t ch = (uint32_t)lane / P, pl0 = (uint32_t)lane % P;
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6), groups = (B + P - 1) / P;
    // (uniform control flow per wave for the DPP scan: the trip count depends on the wave, invalid points ride along masked; one workgroup
    // of 16 waves per CU: the next group's coordinates and gradient are fetched while this one is scattered)
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));  // scalar loop
    float nx0 = 0, nx1 = 0, nx2 = 0, ngc = 0;
    auto fetch = [&](uint32_t g) {
        const uint32_t bb = g * P + pl0 < B ? g * P + pl0 : B - 1;
        nx0 = x[(size_t)bb * 3 + 0], nx1 = x[(size_t)bb * 3 + 1], nx2 = x[(size_t)bb * 3 + 2];
        ngc = grad_layout == 0 ? grad[((size_t)level * B + bb) * C + ch] : grad[(size_t)bb * gp.L * C + level * C + ch];
    };
    if (g0 < groups) fetch(g0);
    for (uint32_t g = g0; g < groups; g += nwaves) {
        const bool inb = g * P + pl0 < B;
        const float x0 = nx0, x1 = nx1, x2 = nx2, gc = ngc;
        if (g + nwaves < groups) fetch(g + nwaves);
        const bool valid = inb && !((x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1));
        float pos[3] = {fmaf(valid ? x0 : 0.5f, scale, half), fmaf(valid ? x1 : 0.5f, scale, half), fmaf(valid ? x2 : 0.5f, scale, half)};
        uint32_t pg[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            pg[d] = (uint32_t)floorf(pos[d]);
            pos[d] -= (float)pg[d];
            if (gp.interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
        }
        uint32_t cidx[8];
        nlr_corner_idx_any(gp, level, pg, cidx);
        // three sweeps over the corners so that the 8 tag exchanges of a wave are in flight together (each is an LDS round trip)
        const NlrRuns runs = nlr_cell_runs(pg, valid, lane);
        float vv[8];
        bool tail[8];
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            float ww = 1.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) ww *= ((c8 >> d) & 1) ? pos[d] : 1 - pos[d];
            vv[c8] = nlr_runs_sum(runs, ww * gc);
            tail[c8] = runs.tail;
        }
        if (direct) {
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8)
                if (tail[c8]) atomicAdd(&acc[cidx[c8] * C + ch], vv[c8]);
        } else {
            uint32_t slot[8], was[8];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                slot[c8] = __umulhi(cidx[c8] * 2654435761u, NS);
                was[c8] = tail[c8] ? atomicCAS(&tags[slot[c8]], 0xffffffffu, cidx[c8]) : 0u;
            }
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8)
                if (tail[c8]) {
                    if (was[c8] == 0xffffffffu || was[c8] == cidx[c8]) atomicAdd(&acc[slot[c8] * C + ch], vv[c8]);
                    else atomicAdd(gt + (size_t)cidx[c8] * C + ch, vv[c8]);
                }
        }
    }
    __syncthreads();
    if (direct) {
        for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x) {
            const float v = acc[i];
            if (v != 0.0f) atomicAdd(gt + i, v);
        }
    } else {
        for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x) {
            const uint32_t tag = tags[i / C];
            const float v = acc[i];
            if (tag != 0xffffffffu && v != 0.0f) atomicAdd(gt + (size_t)tag * C + i % C, v);
        }
    }
}

// =====================================================================================================================================
// Binned scatter (round 4).  The atomic kernels above end at the memory side's rate for scattered float atomics
// (~27 G/s measured: 5.5e8 atomics of a 29.4 M-point proposal-grid backward in 20 ms).  Every level that does not fit the LDS copy of
// nlr_grid_bwd_lds_kernel is cut into BUCKETS of 32 768 floats (128 KiB of its table), and the scatter becomes two streaming passes:
//   pass 1 (nlr_grid_bwd_bin_kernel): a workgroup takes a chunk of NLR_BIN_CHUNK consecutive points of one level, computes their corner
//     updates (runs of equal entries inside a 16-lane row merged as in nlr_run_atomic), counts them per bucket in LDS, scans, and
//     writes them bucket by bucket into the chunk's own region of the workspace: items = (entry index inside the bucket, C values).
//     Per (level, bucket, chunk) it leaves a start and a count.  No global atomics, coalesced-by-bucket writes.
//   pass 2 (nlr_grid_bwd_acc_kernel): NLR_BIN_SPLIT workgroups per bucket walk the chunks' segments of THEIR bucket, add them into an
//     LDS image of the bucket (ds_add_f32) and add the image to the table once, with coalesced atomics on consecutive addresses.
// Global atomics per level drop from (points x 8 corners / run length) scattered ones to (bucket floats x NLR_BIN_SPLIT) coalesced ones;
// the price is the item stream through HBM (8 + 4 C bytes written and read per merged corner update).
// =====================================================================================================================================
#define NLR_BIN_FLOATS 32768u   // floats of table per bucket (128 KiB of LDS in pass 2)
#define NLR_BIN_CHUNK 8192u     // points per pass-1 workgroup (segments of ~1 000 items per bucket: pass 2 reads them in long runs)
#define NLR_BIN_SPLIT 16u       // pass-2 workgroups per bucket at most (a dense level concentrates its items in the few buckets the scene occupies)
#define NLR_BIN_MAXB 256u       // buckets per level at most (2^21 entries x 4 channels)

struct BinArgs {
    uint32_t nlev;             // levels that go through the bins
    uint32_t level[NLR_MAX_GRID_LEVELS];
    uint32_t nb[NLR_MAX_GRID_LEVELS];   // buckets of that level
    uint32_t nchunks, shift;   // entry index >> shift = bucket
    uint32_t split;            // pass-2 workgroups per bucket: every one flushes a whole bucket image, so fewer for the large C = 4 tables
    uint32_t *counts, *starts; // [nlev][NLR_BIN_MAXB][nchunks]
    uint32_t *item_idx;        // [nlev][nchunks][NLR_BIN_CHUNK * 8]
    float *item_val;           // [nlev][nchunks][NLR_BIN_CHUNK * 8][C]
};

template <int C>
__global__ void __launch_bounds__(256) nlr_grid_bwd_bin_kernel(const float *__restrict__ grad, const float *__restrict__ x, GridParams gp,
                                                               uint32_t B, int grad_layout, BinArgs a) {
    __shared__ uint32_t cnt[NLR_BIN_MAXB], cur[NLR_BIN_MAXB];
    const uint32_t chunk = blockIdx.x, li = blockIdx.y, level = a.level[li], nb = a.nb[li];
    const int lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < NLR_BIN_MAXB; i += 256) cnt[i] = 0;
    __syncthreads();

    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    const size_t region = ((size_t)li * a.nchunks + chunk) * (NLR_BIN_CHUNK * 8);
    const uint32_t emask = (1u << a.shift) - 1u;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t it = 0; it < NLR_BIN_CHUNK; it += 256) {
            const uint32_t b0 = chunk * NLR_BIN_CHUNK + it + threadIdx.x;
            const bool inb = b0 < B;
            const uint32_t b = inb ? b0 : B - 1;
            const float x0 = x[(size_t)b * 3 + 0], x1 = x[(size_t)b * 3 + 1], x2 = x[(size_t)b * 3 + 2];
            const bool valid = inb && !((x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1));
            float g[C];
#pragma unroll
            for (int c = 0; c < C; ++c)
                g[c] = grad_layout == 0 ? grad[((size_t)level * B + b) * C + c] : grad[(size_t)b * gp.L * C + level * C + c];
            float pos[3] = {fmaf(valid ? x0 : 0.5f, scale, half), fmaf(valid ? x1 : 0.5f, scale, half), fmaf(valid ? x2 : 0.5f, scale, half)};
            uint32_t pg[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
                if (gp.interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            }
            uint32_t cidx[8];
            nlr_corner_idx_any(gp, level, pg, cidx);
            const NlrRuns runs = nlr_cell_runs(pg, valid, lane);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                float ww = 1.0f;
#pragma unroll
                for (int d = 0; d < 3; ++d) ww *= ((c8 >> d) & 1) ? pos[d] : 1 - pos[d];
                const uint32_t idx = cidx[c8];
                float v[C];
#pragma unroll
                for (int c = 0; c < C; ++c) v[c] = nlr_runs_sum(runs, ww * g[c]);
                if (runs.tail) {
                    const uint32_t bucket = idx >> a.shift;
                    if (pass == 0) {
                        atomicAdd(&cnt[bucket], 1u);
                    } else {
                        const size_t p = region + atomicAdd(&cur[bucket], 1u);
                        a.item_idx[p] = idx & emask;
#pragma unroll
                        for (int c = 0; c < C; ++c) a.item_val[p * C + c] = v[c];
                    }
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x == 0) {
                uint32_t run = 0;
                for (uint32_t i = 0; i < nb; ++i) {
                    cur[i] = run;
                    run += cnt[i];
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < nb; i += 256) {
                const size_t o = ((size_t)li * NLR_BIN_MAXB + i) * a.nchunks + chunk;
                a.counts[o] = cnt[i];
                a.starts[o] = cur[i];
            }
            __syncthreads();
        }
    }
}

template <int C>
__global__ void __launch_bounds__(1024) nlr_grid_bwd_acc_kernel(GridParams gp, float *__restrict__ grad_table, BinArgs a) {
    extern __shared__ float acc[];  // NLR_BIN_FLOATS
    const uint32_t bucket = blockIdx.x, li = blockIdx.y, split = blockIdx.z;
    if (bucket >= a.nb[li]) return;
    const uint32_t level = a.level[li];
    for (uint32_t i = threadIdx.x; i < NLR_BIN_FLOATS; i += 1024) acc[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // 16 waves per workgroup: the 128 KiB image leaves room for ONE workgroup per CU, and a segment read is a dependent chain (start /
    // count, then items), so the loads in flight per CU are what the waves of this workgroup bring (4 waves: 8-10 ms per launch)
    const uint32_t wave = split * 16 + (threadIdx.x >> 6), nwaves = a.split * 16;
    const uint32_t *cn = a.counts + ((size_t)li * NLR_BIN_MAXB + bucket) * a.nchunks;
    const uint32_t *st = a.starts + ((size_t)li * NLR_BIN_MAXB + bucket) * a.nchunks;
    // a wave takes blocks of 8 consecutive chunks (one load brings their (start, count), then the segments one by one).  Small blocks: a
    // dense level puts its items into the few buckets the scene occupies, and all NLR_BIN_SPLIT x 16 waves of such a bucket must get work
    // (blocks of 64 chunks left 57 waves with 350 k items each and the rest idle: 7.3 ms per launch whatever else changed)
    for (uint32_t c0 = wave * 8; c0 < a.nchunks; c0 += nwaves * 8) {
        const uint32_t mine = c0 + (lane & 7) < a.nchunks ? c0 + (lane & 7) : a.nchunks - 1;
        const uint32_t my_n = c0 + (lane & 7) < a.nchunks ? cn[mine] : 0u, my_s = st[mine];
        const uint32_t lim = a.nchunks - c0 < 8 ? a.nchunks - c0 : 8;
        for (uint32_t j = 0; j < lim; ++j) {
            const uint32_t n = __builtin_amdgcn_readlane(my_n, j), s0 = __builtin_amdgcn_readlane(my_s, j);
            const size_t base = ((size_t)li * a.nchunks + (c0 + j)) * (NLR_BIN_CHUNK * 8) + s0;
            // four loads per lane in flight before the first LDS add (one at a time, the 16 waves of a CU leave the memory pipe idle)
            for (uint32_t i = lane; i < n; i += 256) {
                uint32_t e[4];
                float v[4][C];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = i + u * 64 < n ? i + u * 64 : i;
                    e[u] = a.item_idx[base + k];
#pragma unroll
                    for (int c = 0; c < C; ++c) v[u][c] = a.item_val[(base + k) * C + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u * 64 < n) {
#pragma unroll
                        for (int c = 0; c < C; ++c) atomicAdd(&acc[e[u] * C + c], v[u][c]);
                    }
            }
        }
    }
    __syncthreads();
    const uint32_t floats = gp.hsize[level] * C, f0 = bucket * NLR_BIN_FLOATS;
    float *gt = grad_table + (size_t)gp.offset[level] * C + f0;
    for (uint32_t i = threadIdx.x; i < NLR_BIN_FLOATS && f0 + i < floats; i += 1024) {
        const float v = acc[i];
        if (v != 0.0f) atomicAdd(gt + i, v);
    }
}

// Which levels go through the bins, and what workspace that takes (host).
static size_t nlr_bin_plan(const GridParams &gp, uint32_t B, uint32_t C, BinArgs *a) {
    memset(a, 0, sizeof(*a));
    uint32_t sh = 0;
    while ((NLR_BIN_FLOATS / C) >> (sh + 1)) ++sh;  // log2(entries per bucket)
    a->shift = sh;
    a->nchunks = (B + NLR_BIN_CHUNK - 1) / NLR_BIN_CHUNK;
    // C = 4 (the NerfMLP grid: 256 buckets per level, 20-byte items): the first version lost against the atomics (34.1 against 28.2 ms at
    // 14.7 M points, profiles/r04_grid_scatter_ab.txt); behind nlr_debug_set(2, 1) until an A/B on the final pass 2 says otherwise
    if (C > 2 && !nlr_debug_get(2)) return 0;
    a->split = C > 2 ? 4u : NLR_BIN_SPLIT;
    for (uint32_t l = 0; l < gp.L; ++l) {
        if (nlr_level_lds_kind(gp, l, C, nlr_debug_get(NLR_DBG_NO_SCATTER_CACHE))) continue;
        const uint32_t nb = (gp.hsize[l] + (1u << sh) - 1) >> sh;
        if (nb > NLR_BIN_MAXB) return 0;  // (tables beyond 2^21 x 4 floats per level: atomics)
        a->level[a->nlev] = l;
        a->nb[a->nlev++] = nb;
    }
    if (!a->nlev) return 0;
    const size_t tab = (size_t)a->nlev * NLR_BIN_MAXB * a->nchunks * sizeof(uint32_t);
    const size_t items = (size_t)a->nlev * a->nchunks * NLR_BIN_CHUNK * 8;
    return 2 * tab + items * sizeof(uint32_t) + items * C * sizeof(float);
}

__global__ void __launch_bounds__(256) nlr_grid_input_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ dy_dx,
                                                                 float *__restrict__ grad_inputs, uint32_t B, uint32_t L,
                                                                 uint32_t C, int grad_layout) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    const float *dd = dy_dx + (size_t)b * L * 3 * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t c = 0; c < C; ++c) {
            const float g = grad_layout == 0 ? grad[((size_t)l * B + b) * C + c] : grad[(size_t)b * L * C + l * C + c];
            r += g * dd[l * 3 * C + d * C + c];
        }
    grad_inputs[t] = r;
}

extern "C" size_t nlr_grid_backward_workspace_bytes(uint32_t B, uint32_t C, uint32_t L, float S, uint32_t H, const int32_t *offsets_host,
                                                    uint32_t gridtype, int align_corners) {
    if (!offsets_host || B == 0 || !(C == 1 || C == 2 || C == 4)) return 0;
    GridParams gp;
    if (nlr_fill_grid_params(&gp, offsets_host /* any non-null pointer: only the level constants are read */, 0, offsets_host, L, C, S, H, gridtype,
                             align_corners, 0))
        return 0;
    BinArgs a;
    return nlr_bin_plan(gp, B, C, &a);
}

static int nlr_grid_backward_impl(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                  uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                                  uint32_t gridtype, int align_corners, uint32_t interp, int grad_layout, void *workspace, size_t workspace_bytes,
                                  void *stream);

extern "C" int nlr_grid_encode_backward(const float *grad, const float *inputs, const int32_t *offsets_host,
                                        float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                        uint32_t H, const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                                        int align_corners, uint32_t interp, int grad_layout, void *stream) {
    return nlr_grid_backward_impl(grad, inputs, offsets_host, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp,
                                  grad_layout, nullptr, 0, stream);
}

extern "C" int nlr_grid_encode_backward_ws(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                                           float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, int grad_layout,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    return nlr_grid_backward_impl(grad, inputs, offsets_host, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp,
                                  grad_layout, workspace, workspace_bytes, stream);
}

static int nlr_grid_backward_impl(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                  uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                                  uint32_t gridtype, int align_corners, uint32_t interp, int grad_layout, void *workspace, size_t workspace_bytes,
                                  void *stream) {
    NLR_CHECK_ARG(D == 3, "GridEncoding: this build supports input_dim D = 3 only (got %u)", D);
    NLR_CHECK_ARG(grad && inputs && grad_embeddings, "grid_encode_backward: NULL tensor");
    NLR_CHECK_ARG((dy_dx == nullptr) == (grad_inputs == nullptr), "grid_encode_backward: dy_dx and grad_inputs go together");
    if (B == 0) return NLR_OK;
    // one lane per (point, channel) with a 32-bit lane index in both scatter kernels
    NLR_CHECK_ARG((uint64_t)B * C < (1ull << 32), "grid_encode_backward: B * C = %llu (point, channel) pairs do not fit the 32-bit lane index",
                  (unsigned long long)B * C);
    GridParams gp;
    int rc = nlr_fill_grid_params(&gp, grad_embeddings, 0, offsets_host, L, C, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
    // coarse levels go through nlr_grid_bwd_lds_kernel when the batch is large enough to pay for the flush
    uint32_t direct_mask = 0, cache_mask = 0;
    if ((size_t)B * C >= (1u << 18) && C <= 4) {
        const int no_cache = nlr_debug_get(NLR_DBG_NO_SCATTER_CACHE);
        for (uint32_t l = 0; l < L; ++l) {
            const int kind = nlr_level_lds_kind(gp, l, C, no_cache);
            if (kind == 1) direct_mask |= 1u << l;
            if (kind == 2) cache_mask |= 1u << l;
        }
    }
    const uint32_t skip_mask = direct_mask | cache_mask;
    uint64_t level_list = 0;  // 4 bits per scattered level (NLR_MAX_GRID_LEVELS <= 16)
    uint32_t nscat = 0;
    for (uint32_t l = 0; l < L; ++l)
        if (!((skip_mask >> l) & 1u)) level_list |= (uint64_t)l << (4 * nscat++);
    const bool lds_levels = (size_t)B * C >= (1u << 18) && C <= 4;  // (the bin plan leaves the coarse levels to the LDS kernel)
    if (skip_mask) {
        const uint32_t nb = (uint32_t)std::min<size_t>(256, ((size_t)B * C + 16383) / 16384);  // >= 16 K lanes of work per workgroup, one per CU
        dim3 g2(nb, L), b2(1024);
        switch (C) {
            case 1: hipLaunchKernelGGL(nlr_grid_bwd_lds_kernel<1>, g2, b2, 0, st, grad, inputs, gp, grad_embeddings, B, grad_layout, direct_mask, cache_mask); break;
            case 2: hipLaunchKernelGGL(nlr_grid_bwd_lds_kernel<2>, g2, b2, 0, st, grad, inputs, gp, grad_embeddings, B, grad_layout, direct_mask, cache_mask); break;
            default: hipLaunchKernelGGL(nlr_grid_bwd_lds_kernel<4>, g2, b2, 0, st, grad, inputs, gp, grad_embeddings, B, grad_layout, direct_mask, cache_mask); break;
        }
        NLR_LAUNCH_CHECK("nlr_grid_bwd_lds_kernel");
    }
    // the other levels: through the bins when the caller brought the workspace for it (and the LDS kernel took the small levels), else atomics
    BinArgs ba;
    const size_t need = (lds_levels && workspace && (C == 1 || C == 2 || C == 4)) ? nlr_bin_plan(gp, B, C, &ba) : 0;
    if (need && workspace_bytes >= need) {
        const size_t tab = (size_t)ba.nlev * NLR_BIN_MAXB * ba.nchunks, items = (size_t)ba.nlev * ba.nchunks * NLR_BIN_CHUNK * 8;
        ba.counts = (uint32_t *)workspace;
        ba.starts = ba.counts + tab;
        ba.item_idx = ba.starts + tab;
        ba.item_val = (float *)(ba.item_idx + items);
        uint32_t nbmax = 0;
        for (uint32_t i = 0; i < ba.nlev; ++i) nbmax = ba.nb[i] > nbmax ? ba.nb[i] : nbmax;
        dim3 g1(ba.nchunks, ba.nlev), g2b(nbmax, ba.nlev, ba.split), b1024(1024);
        const size_t lds = NLR_BIN_FLOATS * sizeof(float);
#define NLR_BIN_LAUNCH(CC)                                                                                                            \
    do {                                                                                                                              \
        hipLaunchKernelGGL(nlr_grid_bwd_bin_kernel<CC>, g1, block, 0, st, grad, inputs, gp, B, grad_layout, ba);                      \
        static bool attr_##CC = false;                                                                                                \
        if (!attr_##CC) {                                                                                                             \
            (void)hipFuncSetAttribute((const void *)nlr_grid_bwd_acc_kernel<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_##CC = true;                                                                                                         \
        }                                                                                                                             \
        hipLaunchKernelGGL(nlr_grid_bwd_acc_kernel<CC>, g2b, b1024, lds, st, gp, grad_embeddings, ba);                                \
    } while (0)
        if (C == 1) NLR_BIN_LAUNCH(1);
        else if (C == 2) NLR_BIN_LAUNCH(2);
        else NLR_BIN_LAUNCH(4);
#undef NLR_BIN_LAUNCH
        NLR_LAUNCH_CHECK("nlr_grid_bwd_bin_kernel / nlr_grid_bwd_acc_kernel");
    } else if (nscat) {
        dim3 grid((unsigned)(((size_t)B * C + 255) / 256), nscat);  // one lane per (point, channel)
        // both x-corners per atomic instruction (twice the waves, half the instructions each) unless the A/B switch asks for round 3's kernel
        const bool xpair = C <= 4 && !nlr_debug_get(NLR_DBG_NO_XPAIR_SCATTER);
        if (xpair) NLR_CHECK_ARG((uint64_t)B * 2 * C < (1ull << 32), "grid_encode_backward: B = %u points do not fit the 32-bit lane index of the x-pair scatter", B);
        dim3 gridx((unsigned)(((size_t)B * 2 * C + 255) / 256), nscat);
        const uint32_t lvmask = (uint32_t)nlr_debug_get(NLR_DBG_SCATTER_LEVELS);
#define NLR_SCATTER_LAUNCH(CC)                                                                                                                    \
    do {                                                                                                                                          \
        if (xpair) hipLaunchKernelGGL(nlr_grid_bwd_xpair_kernel<CC>, gridx, block, 0, st, grad, inputs, gp, grad_embeddings, B, grad_layout, level_list, lvmask); \
        else hipLaunchKernelGGL(nlr_grid_bwd_kernel<CC>, grid, block, 0, st, grad, inputs, gp, grad_embeddings, B, grad_layout, level_list);        \
    } while (0)
        switch (C) {
            case 1: NLR_SCATTER_LAUNCH(1); break;
            case 2: NLR_SCATTER_LAUNCH(2); break;
            case 4: NLR_SCATTER_LAUNCH(4); break;
            default: hipLaunchKernelGGL(nlr_grid_bwd_kernel<8>, grid, block, 0, st, grad, inputs, gp, grad_embeddings, B, grad_layout, level_list); break;
        }
#undef NLR_SCATTER_LAUNCH
        NLR_LAUNCH_CHECK("nlr_grid_bwd_kernel");
    }
    if (dy_dx) {
        hipLaunchKernelGGL(nlr_grid_input_bwd_kernel, dim3((B * 3 + 255) / 256), block, 0, st, grad, dy_dx, grad_inputs, B, L, C,
                           grad_layout);
        NLR_LAUNCH_CHECK("nlr_grid_input_bwd_kernel");
    }
    return NLR_OK;
}

// ---------------------------------------------------------------------------------------------
// total-variation gradient (gridencoder.cu:506-601; grid.py:176-198 calls it between loss.backward() and optimizer.step())
// ---------------------------------------------------------------------------------------------
// One thread per (point, level), as the reference: the work per thread is 7 gathers of C values and C atomics; the atomics go to
// the cell corners the batch happens to draw (distinct addresses in the hashed levels, a few thousand hot cells in the dense ones).
template <int C>
__global__ void __launch_bounds__(256) nlr_grid_tv_kernel(const float *__restrict__ x, GridParams gp, float *__restrict__ grad, float weight,
                                                         uint32_t B) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const float *tb = (const float *)gp.table + (size_t)gp.offset[level] * C;
    float *gg = grad + (size_t)gp.offset[level] * C;
    const float x0 = x[(size_t)b * 3 + 0], x1 = x[(size_t)b * 3 + 1], x2 = x[(size_t)b * 3 + 2];
    if ((x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1)) return;
    const uint32_t hsize = gp.hsize[level], res = gp.res[level];
    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    uint32_t pg[3] = {(uint32_t)floorf(fmaf(x0, scale, half)), (uint32_t)floorf(fmaf(x1, scale, half)), (uint32_t)floorf(fmaf(x2, scale, half))};
    float results[C], idelta[C], center[C];
    const uint32_t index = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pg[0], pg[1], pg[2]) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        results[ch] = 0.0f;
        idelta[ch] = 0.0f;
        center[ch] = tb[index + ch];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const uint32_t cur = pg[d];
        if (cur < res) {  // right neighbour
            pg[d] = cur + 1;
            const uint32_t ir = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pg[0], pg[1], pg[2]) * C;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float g = center[ch] - tb[ir + ch];
                results[ch] += g;
                idelta[ch] = fmaf(g, g, idelta[ch]);
            }
        }
        if (cur > 0) {  // left neighbour
            pg[d] = cur - 1;
            const uint32_t il = nlr_grid_index(gp.gridtype, gp.align_corners, hsize, res, pg[0], pg[1], pg[2]) * C;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float g = center[ch] - tb[il + ch];
                results[ch] += g;
                idelta[ch] = fmaf(g, g, idelta[ch]);
            }
        }
        pg[d] = cur;
    }
    const float w = weight / 6.0f;  // weight / (2 * D)
#pragma unroll
    for (int ch = 0; ch < C; ++ch) atomicAdd(gg + index + ch, (w * results[ch]) * __frsqrt_rn(idelta[ch] + 1e-9f));
}

extern "C" int nlr_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets_host, float weight,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                        int align_corners, void *stream) {
    NLR_CHECK_ARG(D == 3, "GridEncoding: this build supports input_dim D = 3 only (got %u)", D);
    NLR_CHECK_ARG(inputs && embeddings && grad, "grad_total_variation: NULL tensor");
    if (B == 0) return NLR_OK;
    GridParams gp;
    int rc = nlr_fill_grid_params(&gp, embeddings, 0, offsets_host, L, C, S, H, gridtype, align_corners, 0);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((B + 255) / 256, L), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL(nlr_grid_tv_kernel<1>, grid, block, 0, st, inputs, gp, grad, weight, B); break;
        case 2: hipLaunchKernelGGL(nlr_grid_tv_kernel<2>, grid, block, 0, st, inputs, gp, grad, weight, B); break;
        case 4: hipLaunchKernelGGL(nlr_grid_tv_kernel<4>, grid, block, 0, st, inputs, gp, grad, weight, B); break;
        default: hipLaunchKernelGGL(nlr_grid_tv_kernel<8>, grid, block, 0, st, inputs, gp, grad, weight, B); break;
    }
    NLR_LAUNCH_CHECK("nlr_grid_tv_kernel");
    return NLR_OK;
}
