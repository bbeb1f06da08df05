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
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // (N * S < 2^32, checked by the host: no 64-bit division here)
    if (i >= N * S) return;
    const uint32_t ray = i / S, k = i - ray * S;
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
    NLR_CHECK_ARG(M < (1ull << 32), "box_winner: N * S = %zu does not fit the 32-bit sample index", M);
    hipLaunchKernelGGL(nlr_box_winner_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tdist, origins, directions,
                       box_params, N, S, n_obj, winner);
    NLR_LAUNCH_CHECK("nlr_box_winner_kernel");
    return NLR_OK;
}

// =============================================================================================================
// The whole branch on the device (header section 7b): pose blend -> owner + per-class compaction -> object networks
// =============================================================================================================
#include "nlr_grid_level.h"
#include "nlr_objects.h"

#include <memory>
#include <vector>

// ---- get_pose + world2object constants (obj_utils.py:431-475, :5-28,:158-170), one thread per (ray, track) ----------
__global__ void __launch_bounds__(256) nlr_track_box_kernel(const float *__restrict__ tracks, const float *__restrict__ ts, uint32_t N,
                                                           uint32_t n_obj, uint32_t T, float *__restrict__ box) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * n_obj) return;
    const uint32_t ray = (uint32_t)(i / n_obj), o = (uint32_t)(i - (size_t)ray * n_obj);
    const float time = ts[ray];
    const float *tr = tracks + (size_t)o * T * 9;
    // the two records closest in time (torch.sort(time_diff)[..., :2]); on equal distances the earlier record first
    uint32_t i1 = 0, i2 = 0;
    float d1 = INFINITY, d2 = INFINITY;
    for (uint32_t k = 0; k < T; ++k) {
        const float d = fabsf(time - tr[k * 9 + 7]);
        if (d < d1) {
            d2 = d1;
            i2 = i1;
            d1 = d;
            i1 = k;
        } else if (d < d2) {
            d2 = d;
            i2 = k;
        }
    }
    const float t1 = tr[i1 * 9 + 7], t2 = tr[i2 * 9 + 7];
    float w1 = fabsf(time - t2) / (fabsf(t1 - t2) + 1e-9f);
    w1 = fminf(fmaxf(w1, 0.0f), 1.0f);
    float pose[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) pose[c] = w1 * tr[i1 * 9 + c] + (1.0f - w1) * tr[i2 * 9 + c];
    const float cs = cosf(pose[3]), sn = sinf(pose[3]);
    // t_w_o = rotate_yaw_z(-center, theta) with the reference's quirk (y' from the rotated x', obj_utils.py:106-107)
    const float nx = -pose[0], ny = -pose[1];
    const float px = cs * nx - sn * ny;
    const float py = sn * px + cs * ny;
    float *b = box + i * 8;
    b[0] = cs;
    b[1] = sn;
    b[2] = px;
    b[3] = py;
    b[4] = -pose[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) b[5 + c] = 1.0f / (pose[4 + c] / 2.0f + 1e-9f);
}

extern "C" int nlr_track_box_params(const float *tracks, const float *timestamps, uint32_t N, uint32_t n_obj, uint32_t T,
                                    float *box_params, void *stream) {
    if (N == 0 || n_obj == 0) return NLR_OK;
    NLR_CHECK_ARG(tracks && timestamps && box_params, "track_box_params: NULL tensor");
    NLR_CHECK_ARG(T >= 2, "track_box_params: get_pose blends two recorded poses, T = %u", T);
    const size_t tot = (size_t)N * n_obj;
    hipLaunchKernelGGL(nlr_track_box_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tracks, timestamps, N,
                       n_obj, T, box_params);
    NLR_LAUNCH_CHECK("nlr_track_box_kernel");
    return NLR_OK;
}

// ---- owner of every sample + per-class lists of the owned samples ---------------------------------------------------------
// Three small kernels and no atomics (device-scope atomics on one counter serialise at the memory side across the 8 XCDs:
// 10 k of them cost more than the rest of the branch): (1) owner map + the number of owned samples of every class in every
// wave, (2) exclusive scan of those counts per class, (3) scatter.  The lists come out in sample order.
//   lists[c * cap + i] = flat sample index (ray * S + k), counts[c] = length.
__global__ void __launch_bounds__(256) nlr_box_owner_kernel(const float *__restrict__ tdist, const float *__restrict__ origins,
                                                           const float *__restrict__ dirs, const float *__restrict__ box, uint32_t N,
                                                           uint32_t S, uint32_t n_obj, const int32_t *__restrict__ track_class,
                                                           uint32_t n_classes, int32_t *__restrict__ winner,
                                                           uint32_t *__restrict__ wave_counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // (N * S < 2^32, checked by the host: no 64-bit division here)
    const bool in = i < N * S;
    int32_t w = -1;
    if (in) {
        const uint32_t ray = i / S, k = i - ray * S;
        const float t0 = tdist[(size_t)ray * (S + 1) + k], t1 = tdist[(size_t)ray * (S + 1) + k + 1];
        const float tm = 0.5f * (t0 + t1);
        float p[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = tm * dirs[(size_t)ray * 3 + c] + origins[(size_t)ray * 3 + c];
        const f32x4 *b = reinterpret_cast<const f32x4 *>(box + (size_t)ray * n_obj * 8);  // (32-byte records of a float array)
        // four boxes per trip: their eight 16-byte loads are issued together, and the inside test is branch-free
        for (uint32_t o0 = 0; o0 < n_obj; o0 += 4) {
            f32x4 b0[4], b1[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t o = o0 + u < n_obj ? o0 + u : n_obj - 1;
                b0[u] = b[2 * o];
                b1[u] = b[2 * o + 1];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const float cs = b0[u][0], sn = b0[u][1];
                const float rx = cs * p[0] - sn * p[1];
                const float ry = sn * rx + cs * p[1];  // (sic) the rotated x
                const float x = b1[u][1] * (rx + b0[u][2]), y = b1[u][2] * (ry + b0[u][3]), z = b1[u][3] * (p[2] + b1[u][0]);
                const int inside = (int)(fabsf(x) < 1.0f) & (int)(fabsf(y) < 1.0f) & (int)(fabsf(z) < 1.0f) & (int)(o0 + u < n_obj);
                w = inside ? (int32_t)(o0 + u) : w;
            }
        }
        winner[i] = w;
    }
    const int32_t cls = w >= 0 ? track_class[w] : -1;
    const uint32_t lane = threadIdx.x & 63, gw = i >> 6;
    for (uint32_t c = 0; c < n_classes; ++c) {
        const uint64_t mask = __ballot(cls == (int32_t)c);
        if (lane == 0) wave_counts[(size_t)c * gridDim.x * 4 + gw] = (uint32_t)__popcll(mask);
    }
}

// exclusive scan of one class's per-wave counts (blockIdx.x = class), in place; counts[c] = total
__global__ void __launch_bounds__(1024) nlr_obj_scan_kernel(uint32_t *__restrict__ wave_counts, uint32_t nw, uint32_t *__restrict__ counts) {
    __shared__ uint32_t part[1024];
    uint32_t *wc = wave_counts + (size_t)blockIdx.x * nw;  // nw is a multiple of 4 and the slab 16-byte aligned: 16-byte accesses
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t per = (((nw + 1023) / 1024) + 3) & ~3u, lo = threadIdx.x * per, hi = lo + per < nw ? lo + per : nw;
    uint32_t sum = 0;
    for (uint32_t j = lo; j < hi; j += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(wc + j);
        sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t j = lo; j < hi; j += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(wc + j);
        u32x4 o;
        o[0] = run;
        o[1] = o[0] + v[0];
        o[2] = o[1] + v[1];
        o[3] = o[2] + v[2];
        run = o[3] + v[3];
        *reinterpret_cast<u32x4 *>(wc + j) = o;
    }
    if (threadIdx.x == 1023) counts[blockIdx.x] = part[1023];
}

__global__ void __launch_bounds__(256) nlr_obj_scatter_kernel(const int32_t *__restrict__ winner, const int32_t *__restrict__ track_class,
                                                             uint32_t M, uint32_t n_classes, const uint32_t *__restrict__ wave_offsets,
                                                             uint32_t *__restrict__ lists, uint32_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t w = i < M ? winner[i] : -1;
    const int32_t cls = w >= 0 ? track_class[w] : -1;
    const uint32_t lane = threadIdx.x & 63, gw = i >> 6;
    if (__ballot(cls >= 0) == 0) return;
    for (uint32_t c = 0; c < n_classes; ++c) {
        const uint64_t mask = __ballot(cls == (int32_t)c);
        if (cls == (int32_t)c)
            lists[(size_t)c * cap + wave_offsets[(size_t)c * gridDim.x * 4 + gw] + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    }
}

// ---- one class's network on its list -------------------------------------------------------------------------------------
// One lane per owned sample, 64 samples per wave, 3 waves per workgroup, one workgroup per CU.  The class's weights (78 KB for
// the shipped ObjMLP) are copied into LDS once per workgroup, transposed ([input][output], outputs padded to 64 / 32 / 4): a
// weight row is read with wave-uniform ds_read_b128s (broadcast) and feeds packed-f32 FMAs whose 64 / 32 accumulators - the
// layer's outputs - stay in VGPRs.  A layer's inputs are walked by the running input index: grid features and direction
// encoding from registers (compile-time loops), the latent code from the track's row in memory, the previous layer's outputs
// from the wave's [feature][lane] slab in LDS (written only after the layer that reads the old contents has finished: the
// outputs wait in registers, so one 64-feature and one 32-feature slab per wave are enough).
struct ObjNet {
    GridParams gp;
    const float *wpack;       // dev: all weight matrices + biases as the kernel's LDS image
    uint32_t wfloats;         // its size (multiple of 4)
    uint32_t o_d0, o_d2, o_v[NLR_OBJ_MAX_DEPTH], o_rgb;        // weight offsets (floats) inside the image
    uint32_t o_bd0, o_bd2, o_bv[NLR_OBJ_MAX_DEPTH], o_brgb;    // bias offsets
    const float *latents;     // [n_tracks, latent_size]
    uint32_t n_grid, lat_size, lat_shape, lat_tex, lat_tex_off, BW, W, D, skip, deg, DE;
    float density_bias, premul, rgb_bias, rgb_pad;
    int32_t class_type;
};

struct ObjApply {
    const uint32_t *lists, *counts;  // [n_classes][cap], [n_classes]
    uint32_t cap;
    const int32_t *winner;
    const float *tdist, *origins, *dirs, *viewdirs, *box;
    uint32_t N, S, n_obj, K;
    float *density, *rgb, *sem;
};

#define NLR_OBJ_WAVES 3
#define NLR_OBJ_WMAX 20480  // floats of LDS for the weight image (80 KiB)

template <int OUT>
__device__ __forceinline__ void obj_bias(float (&acc)[OUT], const float *b) {
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = b[o];
}
template <int OUT>
__device__ __forceinline__ void obj_fma_row(float (&acc)[OUT], const float *wrow, float x) {
#pragma unroll
    for (int o = 0; o < OUT; o += 4) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[o + e] = fmaf(w[e], x, acc[o + e]);
    }
}
// inputs from the wave's LDS slab [k][lane]
template <int OUT>
__device__ __forceinline__ void obj_seg_lds(float (&acc)[OUT], const float *w, uint32_t row0, uint32_t n, const float *slab, uint32_t lane) {
    const float *wr = w + (size_t)row0 * OUT;
#pragma unroll 2
    for (uint32_t k = 0; k < n; ++k) obj_fma_row<OUT>(acc, wr + (size_t)k * OUT, slab[k * 64 + lane]);
}
// inputs from a per-lane row in memory (the track's latent code)
template <int OUT>
__device__ __forceinline__ void obj_seg_mem(float (&acc)[OUT], const float *w, uint32_t row0, uint32_t n, const float *x) {
    const float *wr = w + (size_t)row0 * OUT;
#pragma unroll 2
    for (uint32_t k = 0; k < n; ++k) obj_fma_row<OUT>(acc, wr + (size_t)k * OUT, x[k]);
}
// inputs from registers
template <int OUT, int NR>
__device__ __forceinline__ void obj_seg_reg(float (&acc)[OUT], const float *w, uint32_t row0, uint32_t n, const float (&x)[NR]) {
    const float *wr = w + (size_t)row0 * OUT;
#pragma unroll
    for (int k = 0; k < NR; ++k)
        if ((uint32_t)k < n) obj_fma_row<OUT>(acc, wr + (size_t)k * OUT, x[k]);
}

template <typename T, int C, int LMAX>
__global__ void __launch_bounds__(64 * NLR_OBJ_WAVES) nlr_objmlp_kernel(const ObjNet *__restrict__ nets, ObjApply a) {
    // blockIdx.y = class: the workgroups of a class with few (or no) owned samples leave at once and their CUs go to the others
    const ObjNet &net = nets[blockIdx.y];
    const uint32_t *list = a.lists + (size_t)blockIdx.y * a.cap;
    __shared__ __attribute__((aligned(16))) float wl[NLR_OBJ_WMAX];
    __shared__ float slab_a[NLR_OBJ_WAVES][64 * 64];  // trunk hidden, then the bottleneck
    __shared__ float slab_v[NLR_OBJ_WAVES][32 * 64];  // view layers
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t count = a.counts[blockIdx.y];
    if (blockIdx.x * (64 * NLR_OBJ_WAVES) >= count) return;  // (the whole workgroup: nothing to do)
    for (uint32_t i = threadIdx.x * 4; i < net.wfloats; i += 64 * NLR_OBJ_WAVES * 4)
        *reinterpret_cast<f32x4 *>(wl + i) = *reinterpret_cast<const f32x4 *>(net.wpack + i);
    __syncthreads();
    float *sa = slab_a[wave], *sv = slab_v[wave];
    const bool want_rgb = a.rgb != nullptr;
    for (uint32_t base = (blockIdx.x * NLR_OBJ_WAVES + wave) * 64; base < count; base += gridDim.x * (64 * NLR_OBJ_WAVES)) {
        const bool valid = base + lane < count;
        const uint32_t m = list[valid ? base + lane : count - 1];
        const uint32_t ray = m / a.S, k = m - ray * a.S;
        // ---- box coordinates of the interval midpoint, obj_utils.py:158-176,203-216
        const float t0 = a.tdist[(size_t)ray * (a.S + 1) + k], t1 = a.tdist[(size_t)ray * (a.S + 1) + k + 1];
        const float tm = 0.5f * (t0 + t1);
        float pw[3], vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pw[c] = tm * a.dirs[(size_t)ray * 3 + c] + a.origins[(size_t)ray * 3 + c];
            vd[c] = want_rgb ? a.viewdirs[(size_t)ray * 3 + c] : 0.0f;
        }
        const uint32_t tr = (uint32_t)a.winner[m];
        const float *b = a.box + ((size_t)ray * a.n_obj + tr) * 8;
        const float cs = b[0], sn = b[1];
        const float rx = cs * pw[0] - sn * pw[1];
        const float ry = sn * rx + cs * pw[1];
        Gauss g;  // GridEncoder(bound = 1): (x + 1) / 2 (grid.py:162)
        g.x0 = ((b[5] * (rx + b[2])) + 1.0f) / 2.0f;
        g.x1 = ((b[6] * (ry + b[3])) + 1.0f) / 2.0f;
        g.x2 = ((b[7] * (pw[2] + b[4])) + 1.0f) / 2.0f;
        g.zs = 0.0f;
        float de[3], ds0[3 * NLR_OBJ_MAX_DEG], ds1[3 * NLR_OBJ_MAX_DEG];  // [x | sin(2^j x) | sin(2^j x + pi/2)], j-major
        if (want_rgb) {  // view direction in the box frame, normalised (obj_utils.py:171-176), then pos_enc (coord.py:199-210)
            const float vx = cs * vd[0] - sn * vd[1];
            const float vy = sn * vx + cs * vd[1];
            float d[3] = {b[5] * vx, b[6] * vy, b[7] * vd[2]};
            const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d[c] = d[c] / nrm;
                de[c] = d[c];
            }
#pragma unroll
            for (int j = 0; j < NLR_OBJ_MAX_DEG; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float xb = d[c] * (float)(1 << j);
                    ds0[3 * j + c] = sinf(xb);
                    ds1[3 * j + c] = sinf(xb + 0.5f * 3.14159265358979323846f);
                }
        }
        // ---- grid features (gridencoder.cu:87-199 on the box coordinates)
        float gf[LMAX * C];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            float acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.0f;
            if ((uint32_t)l < net.gp.L) nlr_level_accum<T, C>(net.gp, l, g, 1.0f, acc);
#pragma unroll
            for (int c = 0; c < C; ++c) gf[l * C + c] = acc[c];
        }
        const float *lat = net.latents + (size_t)tr * net.lat_size;
        // ---- density_layer: [grid | shape latent] -> 64 -> ReLU -> bottleneck (models.py:887-889,996-1003)
        {
            float h[64];
            obj_bias<64>(h, wl + net.o_bd0);
            obj_seg_reg<64, LMAX * C>(h, wl + net.o_d0, 0, net.n_grid, gf);
            obj_seg_mem<64>(h, wl + net.o_d0, net.n_grid, net.lat_shape, lat);
#pragma unroll
            for (int o = 0; o < 64; ++o) sa[o * 64 + lane] = fmaxf(h[o], 0.0f);
        }
        float x[64];
        if (want_rgb) {
            obj_bias<64>(x, wl + net.o_bd2);
            obj_seg_lds<64>(x, wl + net.o_d2, 0, 64, sa, lane);
        } else {  // proposal levels replace the density only: output 0 of density_layer.2
            float r = wl[net.o_bd2];
            for (uint32_t kk = 0; kk < 64; ++kk) r = fmaf(wl[net.o_d2 + kk * 64], sa[kk * 64 + lane], r);
            x[0] = r;
        }
        {
            const float xd = x[0] + net.density_bias;
            if (valid) a.density[m] = xd > 20.0f ? xd : log1pf(expf(xd));  // F.softplus, models.py:1116
        }
        if (valid && a.sem) {  // fixed_semantic: one-hot of the class (models.py:1124-1130)
            for (uint32_t c = 0; c < a.K; ++c) a.sem[(size_t)c * a.N * a.S + m] = ((int32_t)c == net.class_type) ? 1.0f : 0.0f;
        }
        if (!want_rgb) continue;
#pragma unroll
        for (int o = 0; o < 64; ++o) sa[o * 64 + lane] = x[o];  // (every read of the hidden units above has completed: same wave)
        // ---- view MLP: inputs = [bottleneck | dir enc | texture latent] (models.py:1190-1234)
        const float *lat_tex = lat + net.lat_tex_off;
        for (uint32_t i = 0; i < net.D; ++i) {
            float h[32];
            const float *wv = wl + net.o_v[i];
            obj_bias<32>(h, wl + net.o_bv[i]);
            uint32_t row = 0;
            if (i > 0) {
                obj_seg_lds<32>(h, wv, 0, net.W, sv, lane);
                row = net.W;
            }
            if (i == 0 || i - 1 == net.skip) {
                obj_seg_lds<32>(h, wv, row, net.BW, sa, lane);
                obj_seg_reg<32, 3>(h, wv, row + net.BW, 3, de);
                obj_seg_reg<32, 3 * NLR_OBJ_MAX_DEG>(h, wv, row + net.BW + 3, 3 * net.deg, ds0);
                obj_seg_reg<32, 3 * NLR_OBJ_MAX_DEG>(h, wv, row + net.BW + 3 + 3 * net.deg, 3 * net.deg, ds1);
                obj_seg_mem<32>(h, wv, row + net.BW + net.DE, net.lat_tex, lat_tex);
            }
#pragma unroll
            for (int o = 0; o < 32; ++o) sv[o * 64 + lane] = fmaxf(h[o], 0.0f);
        }
        float c4[4];
        const float *wr = wl + net.o_rgb;
        obj_bias<4>(c4, wl + net.o_brgb);
        obj_seg_lds<4>(c4, wr, 0, net.W, sv, lane);
        if (net.D - 1 == net.skip) {
            obj_seg_lds<4>(c4, wr, net.W, net.BW, sa, lane);
            obj_seg_reg<4, 3>(c4, wr, net.W + net.BW, 3, de);
            obj_seg_reg<4, 3 * NLR_OBJ_MAX_DEG>(c4, wr, net.W + net.BW + 3, 3 * net.deg, ds0);
            obj_seg_reg<4, 3 * NLR_OBJ_MAX_DEG>(c4, wr, net.W + net.BW + 3 + 3 * net.deg, 3 * net.deg, ds1);
            obj_seg_mem<4>(c4, wr, net.W + net.BW + net.DE, net.lat_tex, lat_tex);
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {  // models.py:1251-1255
                const float sg = 1.0f / (1.0f + expf(-(net.premul * c4[c] + net.rgb_bias)));
                a.rgb[(size_t)c * a.N * a.S + m] = sg * (1.0f + 2.0f * net.rgb_pad) - net.rgb_pad;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct ObjClass {
    ObjNet net;
};
struct NlrObjects {
    std::vector<ObjClass> cls;
    ObjNet *nets_dev = nullptr;  // the same, as the kernel reads them
    std::vector<void *> allocs;
    int32_t *track_class = nullptr;  // dev
    uint32_t n_tracks = 0;
    int device = 0;
    uint32_t cus = 0;
    ~NlrObjects() {
        for (void *p : allocs) (void)hipFree(p);
    }
};

static int obj_upload(NlrObjects *o, const void *host, size_t bytes, void **out, hipStream_t st) {
    void *p = nullptr;
    NLR_HIP(hipMalloc(&p, bytes ? bytes : 16));
    o->allocs.push_back(p);
    if (bytes) {
        NLR_HIP(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, st));
        NLR_HIP(hipStreamSynchronize(st));  // the sources are temporaries
    }
    *out = p;
    return NLR_OK;
}
// nn.Linear weight [out, in] -> [in][OUTP] (zero padded) appended to the class's LDS image; returns its offset
static int obj_pack_linear(std::vector<float> &img, const NlrLinear &l, uint32_t out_f, uint32_t in_f, uint32_t outp, const char *name,
                           uint32_t *w_off) {
    NLR_CHECK_ARG(l.weight && l.bias, "objects: %s weight/bias is NULL", name);
    NLR_CHECK_ARG(l.out_features == out_f && l.in_features == in_f, "objects: %s expected [%u,%u], got [%u,%u]", name, out_f, in_f,
                  l.out_features, l.in_features);
    *w_off = (uint32_t)img.size();
    img.resize(img.size() + (size_t)in_f * outp, 0.0f);
    float *w = img.data() + *w_off;
    for (uint32_t r = 0; r < out_f; ++r)
        for (uint32_t c = 0; c < in_f; ++c) w[(size_t)c * outp + r] = l.weight[(size_t)r * in_f + c];
    return NLR_OK;
}
static void obj_pack_bias(std::vector<float> &img, const NlrLinear &l, uint32_t outp, uint32_t *b_off) {
    *b_off = (uint32_t)img.size();
    img.resize(img.size() + outp, 0.0f);
    for (uint32_t r = 0; r < l.out_features; ++r) img[*b_off + r] = l.bias[r];
}

extern "C" int nlr_objects_create(const NlrObjectsDesc *d, NlrObjects **out, void *stream) {
    NLR_CHECK_ARG(d && out, "objects_create: NULL argument");
    NLR_CHECK_ARG(d->n_classes >= 1 && d->n_classes <= NLR_OBJ_MAX_CLASSES && d->classes, "objects_create: n_classes = %u outside [1,%d]",
                  d->n_classes, NLR_OBJ_MAX_CLASSES);
    NLR_CHECK_ARG(d->n_tracks >= 1 && d->track_class, "objects_create: no tracks");
    hipStream_t st = (hipStream_t)stream;
    std::unique_ptr<NlrObjects> own(new NlrObjects());
    NlrObjects *o = own.get();
    int rc = NLR_OK;
    auto fail = [&](int code) { return code; };
    NLR_HIP(hipGetDevice(&o->device));
    int cus = 0;
    NLR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, o->device));
    o->cus = (uint32_t)cus;
    o->n_tracks = d->n_tracks;
    uint32_t lat_size = d->classes[0].latent_size;
    for (uint32_t t = 0; t < d->n_tracks; ++t)
        if (d->track_class[t] < 0 || (uint32_t)d->track_class[t] >= d->n_classes) {
            nlr_set_error("objects_create: track_class[%u] = %d outside [0,%u)", t, d->track_class[t], d->n_classes);
            return fail(NLR_ERR_INVALID);
        }
    if ((rc = obj_upload(o, d->track_class, (size_t)d->n_tracks * 4, (void **)&o->track_class, st))) return fail(rc);
    const float *lat_dev = nullptr;
    if (lat_size) {
        if (!d->latents) {
            nlr_set_error("objects_create: latent_size = %u but latents is NULL", lat_size);
            return fail(NLR_ERR_INVALID);
        }
        if ((rc = obj_upload(o, d->latents, (size_t)d->n_tracks * lat_size * 4, (void **)&lat_dev, st))) return fail(rc);
    }
    o->cls.resize(d->n_classes);
    for (uint32_t c = 0; c < d->n_classes; ++c) {
        const NlrObjClassDesc &cd = d->classes[c];
        const NlrMlpDesc &md = cd.mlp;
        ObjNet &n = o->cls[c].net;
        memset(&n, 0, sizeof(n));
#define OBJ_REQ(cond, ...)                   \
    if (!(cond)) {                           \
        nlr_set_error(__VA_ARGS__);          \
        return fail(NLR_ERR_UNSUPPORTED);    \
    }
        OBJ_REQ(cd.latent_size == lat_size, "objects_create: classes disagree on latent_size (%u vs %u)", cd.latent_size, lat_size);
        OBJ_REQ(!md.disable_rgb, "objects_create: class %u: an ObjMLP has an rgb branch", c);
        OBJ_REQ(md.bottleneck_width >= 1 && md.bottleneck_width <= 64, "objects_create: bottleneck_width %u outside [1,64]", md.bottleneck_width);
        OBJ_REQ(md.net_width_viewdirs >= 1 && md.net_width_viewdirs <= 32, "objects_create: net_width_viewdirs %u outside [1,32]", md.net_width_viewdirs);
        OBJ_REQ(md.net_depth_viewdirs >= 1 && md.net_depth_viewdirs <= NLR_OBJ_MAX_DEPTH, "objects_create: net_depth_viewdirs %u outside [1,%d]",
                md.net_depth_viewdirs, NLR_OBJ_MAX_DEPTH);
        OBJ_REQ(md.deg_view <= NLR_OBJ_MAX_DEG, "objects_create: deg_view %u > %d", md.deg_view, NLR_OBJ_MAX_DEG);
        OBJ_REQ(md.grid.table_dtype == 0 || md.grid.table_dtype == 1, "objects_create: table dtype");
        OBJ_REQ((md.grid.level_dim == 2 && md.grid.num_levels <= 8) || (md.grid.level_dim == 4 && md.grid.num_levels <= 4) ||
                    (md.grid.level_dim == 1 && md.grid.num_levels <= 16),
                "objects_create: grid L=%u C=%u not supported by the object kernel", md.grid.num_levels, md.grid.level_dim);
        OBJ_REQ(!md.re_weights, "objects_create: ObjMLP runs without erf re-weighting (models.py:137)");
        if ((rc = nlr_fill_grid_params(&n.gp, md.grid.table, md.grid.table_dtype, md.grid.offsets, md.grid.num_levels, md.grid.level_dim,
                                       md.grid.log2_per_level_scale, md.grid.base_resolution, md.grid.gridtype, (int)md.grid.align_corners,
                                       md.grid.interp)))
            return fail(rc);
        n.n_grid = md.grid.num_levels * md.grid.level_dim;
        n.lat_size = lat_size;
        n.lat_shape = lat_size ? (cd.split_latent ? lat_size / 2 : lat_size) : 0;
        n.lat_tex = (lat_size && cd.split_latent) ? lat_size - lat_size / 2 : 0;
        n.lat_tex_off = lat_size / 2;
        n.BW = md.bottleneck_width;
        n.W = md.net_width_viewdirs;
        n.D = md.net_depth_viewdirs;
        n.skip = md.skip_layer_dir;
        n.deg = md.deg_view;
        n.DE = 3 + 6 * md.deg_view;
        n.density_bias = md.density_bias;
        n.premul = md.rgb_premultiplier;
        n.rgb_bias = md.rgb_bias;
        n.rgb_pad = md.rgb_padding;
        n.class_type = cd.class_type;
        n.latents = lat_dev;
        const uint32_t f0 = n.n_grid + n.lat_shape, in_rgb = n.BW + n.DE + n.lat_tex;
        std::vector<float> img;
        if ((rc = obj_pack_linear(img, md.density0, 64, f0, 64, "density_layer.0", &n.o_d0))) return fail(rc);
        if ((rc = obj_pack_linear(img, md.density2, n.BW, 64, 64, "density_layer.2", &n.o_d2))) return fail(rc);
        uint32_t last = in_rgb;
        for (uint32_t i = 0; i < n.D; ++i) {
            if ((rc = obj_pack_linear(img, md.view[i], n.W, last, 32, "lin_second_stage", &n.o_v[i]))) return fail(rc);
            last = n.W + (i == n.skip ? in_rgb : 0);
        }
        if ((rc = obj_pack_linear(img, md.rgb_layer, 3, last, 4, "rgb_layer", &n.o_rgb))) return fail(rc);
        obj_pack_bias(img, md.density0, 64, &n.o_bd0);
        obj_pack_bias(img, md.density2, 64, &n.o_bd2);
        for (uint32_t i = 0; i < n.D; ++i) obj_pack_bias(img, md.view[i], 32, &n.o_bv[i]);
        obj_pack_bias(img, md.rgb_layer, 4, &n.o_brgb);
        OBJ_REQ(img.size() <= NLR_OBJ_WMAX, "objects_create: class %u needs %zu floats of LDS for its weights, the kernel holds %d", c, img.size(),
                NLR_OBJ_WMAX);
        n.wfloats = (uint32_t)img.size();
        if ((rc = obj_upload(o, img.data(), img.size() * 4, (void **)&n.wpack, st))) return fail(rc);
        OBJ_REQ(md.grid.level_dim == d->classes[0].mlp.grid.level_dim && md.grid.table_dtype == d->classes[0].mlp.grid.table_dtype,
                "objects_create: the classes' grids must share level_dim and table dtype (one kernel instance serves them all)");
#undef OBJ_REQ
    }
    {
        std::vector<ObjNet> nets;
        for (auto &c : o->cls) nets.push_back(c.net);
        if ((rc = obj_upload(o, nets.data(), nets.size() * sizeof(ObjNet), (void **)&o->nets_dev, st))) return fail(rc);
    }
    *out = own.release();
    return NLR_OK;
}

extern "C" void nlr_objects_destroy(NlrObjects *o) { delete o; }

static inline size_t obj_al(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t nlr_objects_workspace_bytes(const NlrObjects *o, uint32_t N, uint32_t S) {
    if (!o) return 0;
    const size_t M = (size_t)N * S;
    const size_t nw = ((M + 255) / 256) * 4;
    return 256 + 256 /* counts */ + obj_al(M * 4) /* winner */ + obj_al(o->cls.size() * M * 4) /* lists */ +
           obj_al(o->cls.size() * nw * 4) /* per-wave counts / offsets */;
}

int nlr_objects_apply_impl(const NlrObjects *o, const NlrRays *rays, const float *tdist, const float *box_params, uint32_t N, uint32_t S,
                           uint32_t n_obj, float *density, float *rgb, float *semantic, uint32_t K, int32_t *winner_out, void *workspace,
                           size_t workspace_bytes, hipStream_t st) {
    NLR_CHECK_ARG(o && rays && tdist && density, "objects_apply: NULL argument");
    if (N == 0 || S == 0 || n_obj == 0) return NLR_OK;
    NLR_CHECK_ARG(box_params, "objects_apply: box_params is NULL");
    NLR_CHECK_ARG(n_obj == o->n_tracks, "objects_apply: %u boxes per ray, the object set has %u tracks", n_obj, o->n_tracks);
    NLR_CHECK_ARG(rays->origins && rays->directions && (rays->viewdirs || !rgb), "objects_apply: ray batch has NULL origins/directions/viewdirs");
    const size_t need = nlr_objects_workspace_bytes(o, N, S);
    if (!workspace || workspace_bytes < need)
        NLR_FAIL(NLR_ERR_WORKSPACE, "objects_apply: workspace %zu B < nlr_objects_workspace_bytes() = %zu B", workspace_bytes, need);
    const size_t M = (size_t)N * S;
    NLR_CHECK_ARG(M < (1ull << 32), "objects_apply: N * S = %zu does not fit the 32-bit sample index", M);
    char *p = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    uint32_t *counts = (uint32_t *)p;
    p += 256;
    int32_t *winner = (int32_t *)p;
    p += obj_al(M * 4);
    uint32_t *lists = (uint32_t *)p;
    const uint32_t nc = (uint32_t)o->cls.size();
    p += obj_al((size_t)nc * M * 4);
    uint32_t *wave_counts = (uint32_t *)p;
    const uint32_t nblk = (uint32_t)((M + 255) / 256), nw = nblk * 4;
    hipLaunchKernelGGL(nlr_box_owner_kernel, dim3(nblk), dim3(256), 0, st, tdist, rays->origins, rays->directions, box_params, N, S, n_obj,
                       o->track_class, nc, winner, wave_counts);
    NLR_LAUNCH_CHECK("nlr_box_owner_kernel");
    hipLaunchKernelGGL(nlr_obj_scan_kernel, dim3(nc), dim3(1024), 0, st, wave_counts, nw, counts);
    NLR_LAUNCH_CHECK("nlr_obj_scan_kernel");
    hipLaunchKernelGGL(nlr_obj_scatter_kernel, dim3(nblk), dim3(256), 0, st, winner, o->track_class, (uint32_t)M, nc, wave_counts, lists,
                       (uint32_t)M);
    NLR_LAUNCH_CHECK("nlr_obj_scatter_kernel");
    if (winner_out) NLR_HIP(hipMemcpyAsync(winner_out, winner, M * 4, hipMemcpyDeviceToDevice, st));
    // the networks: one launch, grid.y = class; persistent workgroups that read their class's count on the device
    const uint32_t per = 64 * NLR_OBJ_WAVES, nb = (uint32_t)((M + per - 1) / per);
    const uint32_t grid = o->cus < nb ? o->cus : nb;
    ObjApply a;
    a.lists = lists;
    a.counts = counts;
    a.cap = (uint32_t)M;
    a.winner = winner;
    a.tdist = tdist;
    a.origins = rays->origins;
    a.dirs = rays->directions;
    a.viewdirs = rays->viewdirs;
    a.box = box_params;
    a.N = N;
    a.S = S;
    a.n_obj = n_obj;
    a.K = semantic ? K : 0;
    a.density = density;
    a.rgb = rgb;
    a.sem = semantic;
    const GridParams &gp0 = o->cls[0].net.gp;
#define OBJ_LAUNCH(T, C, LM) hipLaunchKernelGGL((nlr_objmlp_kernel<T, C, LM>), dim3(grid, nc), dim3(64 * NLR_OBJ_WAVES), 0, st, o->nets_dev, a)
    const bool f32t = gp0.table_dtype == 0;
    if (gp0.C == 2) { if (f32t) OBJ_LAUNCH(float, 2, 8); else OBJ_LAUNCH(__half, 2, 8); }
    else if (gp0.C == 4) { if (f32t) OBJ_LAUNCH(float, 4, 4); else OBJ_LAUNCH(__half, 4, 4); }
    else { if (f32t) OBJ_LAUNCH(float, 1, 16); else OBJ_LAUNCH(__half, 1, 16); }
#undef OBJ_LAUNCH
    NLR_LAUNCH_CHECK("nlr_objmlp_kernel");
    return NLR_OK;
}

extern "C" int nlr_objects_apply(const NlrObjects *o, const NlrRays *rays, const float *tdist, const float *box_params, uint32_t N, uint32_t S,
                                 uint32_t n_obj, float *density, float *rgb, float *semantic, uint32_t K, int32_t *winner_out, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    return nlr_objects_apply_impl(o, rays, tdist, box_params, N, S, n_obj, density, rgb, semantic, K, winner_out, workspace, workspace_bytes,
                                  (hipStream_t)stream);
}
