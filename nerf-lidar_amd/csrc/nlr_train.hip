// Training-side operators of the path (scope row f-3): backward of alpha compositing, and the hash-decay regulariser.
//
//   ZI/render.py:170-189,192-252  compute_alpha_weights + volumetric_rendering, differentiated by hand:
//       w_k = alpha_k T_k,  alpha_k = 1 - exp(-dd_k),  T_k = exp(-sum_{j<k} dd_j),  dd_k = density_k * (t_{k+1} - t_k) * |d|
//       dw_k/ddd_k = T_k - w_k,   dw_k/ddd_j = -w_k (j < k)
//       => dL/ddd_j = G_j (T_j - w_j) - sum_{k>j} G_k w_k        with G = dL/dw
//       G_k = g_w[k] + sum_c g_rgb[c] (rgb[c,k] - bg [1 - acc > 0]) + g_depth (tmid_k / accc - [acc > eps] sdep / accc^2) + g_acc
//     semantic and intensity are composited with DETACHED weights (render.py:240-252, sem_detach=True): their upstream
//     gradients reach only the per-sample semantic / intensity values.
//   ZI/models.py:203-223  hash_decay_loss = sum over encoders of mean_{level,channel}( mean_{rows of level} emb^2 )
// One wavefront per ray (as the forward kernel), reverse scan for the suffix sums.
#include "nlr_kernels.h"

#define NLR_CB_MAXPER 8  // S <= 512, as the forward kernel

struct CompositeBwdParams {
    const float *density, *tdist, *dirs, *rgb, *sem, *inten;
    uint32_t N, S, K;
    int opaque;
    float bg;
    const float *g_rgb, *g_depth, *g_sem, *g_int, *g_acc, *g_w;
    float *d_density, *d_rgb, *d_sem, *d_int;
};

__global__ void __launch_bounds__(256) nlr_composite_bwd_kernel(CompositeBwdParams P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4 + wave;
    if (ray >= P.N) return;  // no block-level synchronisation below
    const uint32_t S = P.S, per = (S + 63) / 64, k0 = lane * per;
    const size_t Mt = (size_t)P.N * S;
    const float dx = P.dirs[(size_t)ray * 3], dy = P.dirs[(size_t)ray * 3 + 1], dz = P.dirs[(size_t)ray * 3 + 2];
    const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float *td = P.tdist + (size_t)ray * (S + 1);
    const float *dn = P.density + (size_t)ray * S;

    float dd[NLR_CB_MAXPER], tm[NLR_CB_MAXPER], dl[NLR_CB_MAXPER], w[NLR_CB_MAXPER], T[NLR_CB_MAXPER];
    float run = 0.0f;
#pragma unroll
    for (int i = 0; i < NLR_CB_MAXPER; ++i) {
        const uint32_t k = k0 + i;
        dd[i] = tm[i] = dl[i] = w[i] = T[i] = 0.0f;
        if (i < (int)per && k < S) {
            const float ta0 = td[k], ta1 = td[k + 1];
            dl[i] = (ta1 - ta0) * dnorm;
            float v = dn[k] * dl[i];
            if (P.opaque && k == S - 1) v = INFINITY;
            dd[i] = v;
            tm[i] = 0.5f * (ta0 + ta1);
            run += v;
        }
    }
    const float incl = nlr_wave_incl_scan_add(run, lane);
    float cum = __shfl_up(incl, 1, 64);
    if (lane == 0) cum = 0.0f;
    float acc = 0.0f, sdep = 0.0f;
#pragma unroll
    for (int i = 0; i < NLR_CB_MAXPER; ++i) {
        const uint32_t k = k0 + i;
        if (i < (int)per && k < S) {
            const float alpha = 1.0f - expf(-dd[i]);
            T[i] = expf(-cum);
            w[i] = alpha * T[i];
            cum += dd[i];
            acc += w[i];
            sdep += w[i] * tm[i];
        }
    }
    acc = nlr_wave_sum(acc);
    sdep = nlr_wave_sum(sdep);
    const float accc = fmaxf(acc, NLR_EPS);
    const float gr0 = P.g_rgb ? P.g_rgb[(size_t)ray * 3] : 0.0f, gr1 = P.g_rgb ? P.g_rgb[(size_t)ray * 3 + 1] : 0.0f,
                gr2 = P.g_rgb ? P.g_rgb[(size_t)ray * 3 + 2] : 0.0f;
    const float gd = P.g_depth ? P.g_depth[ray] : 0.0f, ga = P.g_acc ? P.g_acc[ray] : 0.0f, gi = P.g_int ? P.g_int[ray] : 0.0f;
    const float bgterm = (1.0f - acc > 0.0f) ? P.bg * ((gr0 + gr1) + gr2) : 0.0f;  // d(max(1-acc,0) bg)/dw_k = -bg
    const float dterm = (acc > NLR_EPS) ? sdep / (accc * accc) : 0.0f;

    // G_k w_k per sample, per-lane totals, suffix sums over lanes
    float G[NLR_CB_MAXPER], tot = 0.0f;
#pragma unroll
    for (int i = 0; i < NLR_CB_MAXPER; ++i) {
        const uint32_t k = k0 + i;
        G[i] = 0.0f;
        if (i < (int)per && k < S) {
            const size_t mi = (size_t)ray * S + k;
            float g = P.g_w ? P.g_w[mi] : 0.0f;
            if (P.rgb) g += (gr0 * P.rgb[mi] + gr1 * P.rgb[Mt + mi]) + gr2 * P.rgb[2 * Mt + mi];
            g -= bgterm;
            g += gd * (tm[i] / accc - dterm) + ga;
            G[i] = g;
            tot += g * w[i];
            if (P.d_rgb) {
                P.d_rgb[mi] = w[i] * gr0;
                P.d_rgb[Mt + mi] = w[i] * gr1;
                P.d_rgb[2 * Mt + mi] = w[i] * gr2;
            }
            if (P.d_int) P.d_int[mi] = w[i] * gi;
            if (P.d_sem)
                for (uint32_t c = 0; c < P.K; ++c) P.d_sem[(size_t)c * Mt + mi] = w[i] * (P.g_sem ? P.g_sem[(size_t)ray * P.K + c] : 0.0f);
        }
    }
    const float tincl = nlr_wave_incl_scan_add(tot, lane);
    const float total = __shfl(tincl, 63, 64);
    float suffix = total - tincl;  // sum of G w over the lanes to the right
#pragma unroll
    for (int i = NLR_CB_MAXPER - 1; i >= 0; --i) {
        const uint32_t k = k0 + i;
        if (i < (int)per && k < S) {
            // T - w = T exp(-dd): 0 for the opaque last interval (its dd does not come from the density)
            const float own = (P.opaque && k == S - 1) ? 0.0f : G[i] * (T[i] - w[i]);
            const float ddd = own - suffix;
            P.d_density[(size_t)ray * S + k] = (P.opaque && k == S - 1) ? 0.0f : ddd * dl[i];
            suffix += G[i] * w[i];
        }
    }
}

extern "C" int nlr_composite_backward(const float *density, const float *tdist, const float *directions, const float *rgb,
                                      const float *semantic, const float *intensity, uint32_t N, uint32_t S, uint32_t class_num,
                                      int opaque_background, float bg, const float *g_rgb, const float *g_depth,
                                      const float *g_semantic, const float *g_intensity, const float *g_acc, const float *g_weights,
                                      float *d_density, float *d_rgb, float *d_semantic, float *d_intensity, void *stream) {
    if (N == 0) return NLR_OK;
    NLR_CHECK_ARG(density && tdist && directions && d_density, "composite_backward: NULL density/tdist/directions/d_density");
    NLR_CHECK_ARG(S >= 1 && S <= 64 * NLR_CB_MAXPER, "composite_backward: S=%u outside [1,%d]", S, 64 * NLR_CB_MAXPER);
    NLR_CHECK_ARG(!d_semantic || class_num > 0, "composite_backward: d_semantic without class_num");
    CompositeBwdParams P;
    memset(&P, 0, sizeof(P));
    P.density = density; P.tdist = tdist; P.dirs = directions; P.rgb = rgb; P.sem = semantic; P.inten = intensity;
    P.N = N; P.S = S; P.K = class_num; P.opaque = opaque_background; P.bg = bg;
    P.g_rgb = g_rgb; P.g_depth = g_depth; P.g_sem = g_semantic; P.g_int = g_intensity; P.g_acc = g_acc; P.g_w = g_weights;
    P.d_density = d_density; P.d_rgb = d_rgb; P.d_sem = d_semantic; P.d_int = d_intensity;
    hipLaunchKernelGGL(nlr_composite_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, P);
    NLR_LAUNCH_CHECK("nlr_composite_bwd_kernel");
    return NLR_OK;
}

// ---- hash-decay regulariser -------------------------------------------------------------------------------------------
// level_sumsq[l] (double, caller-zeroed) += sum over the level's rows and channels of emb^2
__global__ void __launch_bounds__(256) nlr_hash_sumsq_kernel(const float *__restrict__ emb, uint32_t C, uint32_t row0, uint32_t rows,
                                                           double *__restrict__ out) {
    const size_t n = (size_t)rows * C, base = (size_t)row0 * C;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = emb[base + i];
        s += (double)v * (double)v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    // one atomic per workgroup (<= 256 per level): device-scope atomics on one address serialise at the memory side, 8 192 of
    // them (one per wave of a 2 048-workgroup launch) cost 46 us per level, 24 levels per training step
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}
// grad[i] += scale_l * emb[i] over the level's rows
__global__ void __launch_bounds__(256) nlr_hash_decay_bwd_kernel(const float *__restrict__ emb, uint32_t C, uint32_t row0, uint32_t rows,
                                                               float scale, float *__restrict__ grad) {
    const size_t n = (size_t)rows * C, base = (size_t)row0 * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        grad[base + i] += scale * emb[base + i];
}

extern "C" int nlr_hash_decay_forward(const float *embeddings, const int32_t *offsets_host, uint32_t L, uint32_t C,
                                      double *level_sumsq, void *stream) {
    NLR_CHECK_ARG(embeddings && offsets_host && level_sumsq && L >= 1 && L <= NLR_MAX_GRID_LEVELS && C >= 1, "hash_decay_forward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    NLR_HIP(hipMemsetAsync(level_sumsq, 0, L * sizeof(double), st));
    for (uint32_t l = 0; l < L; ++l) {
        const uint32_t rows = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        if (!rows) continue;
        const size_t n = (size_t)rows * C;
        const unsigned blocks = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
        hipLaunchKernelGGL(nlr_hash_sumsq_kernel, dim3(blocks < 256 ? (blocks ? blocks : 1) : 256), dim3(256), 0, st, embeddings, C,
                           (uint32_t)offsets_host[l], rows, level_sumsq + l);
    }
    NLR_LAUNCH_CHECK("nlr_hash_sumsq_kernel");
    return NLR_OK;
}

extern "C" int nlr_hash_decay_backward(const float *embeddings, const int32_t *offsets_host, uint32_t L, uint32_t C, float upstream,
                                       float *grad_embeddings, void *stream) {
    NLR_CHECK_ARG(embeddings && offsets_host && grad_embeddings && L >= 1 && L <= NLR_MAX_GRID_LEVELS && C >= 1, "hash_decay_backward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (uint32_t l = 0; l < L; ++l) {
        const uint32_t rows = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        if (!rows) continue;
        // d/de [ (1/(L C)) sum_l sum_c (1/n_l) sum_i e_ic^2 ] = 2 e / (n_l L C)
        const float scale = upstream * 2.0f / ((float)rows * (float)L * (float)C);
        const size_t n = (size_t)rows * C;
        const unsigned blocks = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
        hipLaunchKernelGGL(nlr_hash_decay_bwd_kernel, dim3(blocks < 4096 ? (blocks ? blocks : 1) : 4096), dim3(256), 0, st, embeddings, C,
                           (uint32_t)offsets_host[l], rows, scale, grad_embeddings);
    }
    NLR_LAUNCH_CHECK("nlr_hash_decay_bwd_kernel");
    return NLR_OK;
}

// ---- PropMLP density network for training (ZI/models.py:887-889,996-997 with disable_rgb): raw = W2 relu(W1 f + b1) + b2 -------------
// The reference runs it as two nn.Linear on [N*S, 6|8] features; as library GEMMs that is a [262144 x 8] x [8 x 64] product with
// a 64 MB hidden tensor saved for backward (0.53 ms per GEMM on this chip: the shapes are all edge).  Here one lane per sample
// evaluates the 64 hidden units from weights in LDS; the backward recomputes them, so only the features are kept.
#define NLR_PROP_FMAX 16
struct PropTrain {
    const float *feat, *w1, *b1, *w2, *b2;  // [M,F], [64,F], [64], [64], [1]
    uint32_t M, F;
};
__device__ __forceinline__ void nlr_prop_load_weights(const PropTrain &p, float *sw) {  // w1 [64][F] | b1 [64] | w2 [64]
    for (uint32_t i = threadIdx.x; i < 64 * p.F; i += blockDim.x) sw[i] = p.w1[i];
    if (threadIdx.x < 64) {
        sw[64 * p.F + threadIdx.x] = p.b1[threadIdx.x];
        sw[64 * p.F + 64 + threadIdx.x] = p.w2[threadIdx.x];
    }
    __syncthreads();
}
__global__ void __launch_bounds__(256) nlr_prop_mlp_fwd_kernel(PropTrain p, float *__restrict__ raw) {
    __shared__ float sw[64 * NLR_PROP_FMAX + 128];
    nlr_prop_load_weights(p, sw);
    const float b2 = p.b2[0];
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += gridDim.x * blockDim.x) {
        float f[NLR_PROP_FMAX];
#pragma unroll
        for (int i = 0; i < NLR_PROP_FMAX; ++i) f[i] = (uint32_t)i < p.F ? p.feat[(size_t)m * p.F + i] : 0.0f;
        float r = b2;
        for (uint32_t h = 0; h < 64; ++h) {
            float a = sw[64 * p.F + h];
#pragma unroll
            for (int i = 0; i < NLR_PROP_FMAX; ++i)
                if ((uint32_t)i < p.F) a = fmaf(sw[h * p.F + i], f[i], a);
            r = fmaf(sw[64 * p.F + 64 + h], fmaxf(a, 0.0f), r);
        }
        raw[m] = r;
    }
}
// Backward, two kernels.
// (A) d_feat: one lane per sample, the forward loop with dh_h = [a_h > 0] g w2_h pushed back through W1.
// (B) parameter gradients = sums over ALL samples.  One lane per HIDDEN UNIT (lane = h, the 4 waves of a workgroup split each
//     staged chunk of samples): a lane walks its samples with its own row of W1 in registers and accumulates d_w1[h][:], d_b1[h],
//     d_w2[h] privately - no cross-lane reduction inside the loop (the first version reduced 10 values per hidden unit and 64
//     samples across the wave through ds_bpermute: 10.9 ms for 4.2 M samples, LDS-pipe bound).  The samples' features and
//     upstream gradients are staged in LDS and read wave-uniformly (broadcast).  At the end: 4 waves -> LDS -> one global atomic
//     per value and workgroup of a persistent grid.
__global__ void __launch_bounds__(256) nlr_prop_mlp_bwd_feat_kernel(PropTrain p, const float *__restrict__ g_raw, float *__restrict__ d_feat) {
    __shared__ float sw[64 * NLR_PROP_FMAX + 128];
    nlr_prop_load_weights(p, sw);
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += gridDim.x * blockDim.x) {
        float f[NLR_PROP_FMAX], df[NLR_PROP_FMAX];
#pragma unroll
        for (int i = 0; i < NLR_PROP_FMAX; ++i) {
            f[i] = (uint32_t)i < p.F ? p.feat[(size_t)m * p.F + i] : 0.0f;
            df[i] = 0.0f;
        }
        const float g = g_raw[m];
        for (uint32_t h = 0; h < 64; ++h) {
            float a = sw[64 * p.F + h];
#pragma unroll
            for (int i = 0; i < NLR_PROP_FMAX; ++i)
                if ((uint32_t)i < p.F) a = fmaf(sw[h * p.F + i], f[i], a);
            const float dh = a > 0.0f ? g * sw[64 * p.F + 64 + h] : 0.0f;
#pragma unroll
            for (int i = 0; i < NLR_PROP_FMAX; ++i)
                if ((uint32_t)i < p.F) df[i] = fmaf(dh, sw[h * p.F + i], df[i]);
        }
#pragma unroll
        for (int i = 0; i < NLR_PROP_FMAX; ++i)
            if ((uint32_t)i < p.F) d_feat[(size_t)m * p.F + i] = df[i];
    }
}

#define NLR_PROP_CHUNK 512  // samples staged per workgroup and trip
__global__ void __launch_bounds__(256) nlr_prop_mlp_bwd_param_kernel(PropTrain p, const float *__restrict__ g_raw, float *__restrict__ d_w1,
                                                                   float *__restrict__ d_b1, float *__restrict__ d_w2, float *__restrict__ d_b2) {
    __shared__ float sf[NLR_PROP_CHUNK * NLR_PROP_FMAX];  // features of the chunk, [sample][F]
    __shared__ float sg[NLR_PROP_CHUNK];
    __shared__ float acc[64 * NLR_PROP_FMAX + 129];       // d_w1 [64][F] | d_b1 [64] | d_w2 [64] | d_b2
    const uint32_t h = threadIdx.x & 63, part = threadIdx.x >> 6, F = p.F;
    const uint32_t nacc = 64 * F + 129;
    for (uint32_t i = threadIdx.x; i < nacc; i += blockDim.x) acc[i] = 0.0f;
    float w1[NLR_PROP_FMAX], aw1[NLR_PROP_FMAX];
#pragma unroll
    for (int i = 0; i < NLR_PROP_FMAX; ++i) {
        w1[i] = (uint32_t)i < F ? p.w1[h * F + i] : 0.0f;
        aw1[i] = 0.0f;
    }
    const float b1 = p.b1[h], w2 = p.w2[h];
    float ab1 = 0.0f, aw2 = 0.0f, ab2 = 0.0f;
    for (uint32_t c0 = blockIdx.x * NLR_PROP_CHUNK; c0 < p.M; c0 += gridDim.x * NLR_PROP_CHUNK) {
        const uint32_t cn = p.M - c0 < NLR_PROP_CHUNK ? p.M - c0 : NLR_PROP_CHUNK;
        __syncthreads();  // (the previous chunk has been consumed)
        for (uint32_t i = threadIdx.x; i < cn * F; i += blockDim.x) sf[i] = p.feat[(size_t)c0 * F + i];
        for (uint32_t i = threadIdx.x; i < cn; i += blockDim.x) sg[i] = g_raw[c0 + i];
        __syncthreads();
        // this wave's quarter of the chunk; every lane reads the same sample (LDS broadcast) and works on its own hidden unit
        const uint32_t per = (cn + 3) / 4, s0 = part * per, s1 = s0 + per < cn ? s0 + per : cn;
        for (uint32_t s = s0; s < s1; ++s) {
            const float *fs = sf + s * F;
            const float g = sg[s];
            float f[NLR_PROP_FMAX];
            float a = b1;
#pragma unroll
            for (int i = 0; i < NLR_PROP_FMAX; ++i) {
                f[i] = (uint32_t)i < F ? fs[i] : 0.0f;
                a = fmaf(w1[i], f[i], a);
            }
            const float dh = a > 0.0f ? g * w2 : 0.0f;
#pragma unroll
            for (int i = 0; i < NLR_PROP_FMAX; ++i) aw1[i] = fmaf(dh, f[i], aw1[i]);
            ab1 += dh;
            aw2 = fmaf(g, fmaxf(a, 0.0f), aw2);
            ab2 += g;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLR_PROP_FMAX; ++i)
        if ((uint32_t)i < F) atomicAdd(&acc[h * F + i], aw1[i]);
    atomicAdd(&acc[64 * F + h], ab1);
    atomicAdd(&acc[64 * F + 64 + h], aw2);
    if (h == 0) atomicAdd(&acc[64 * F + 128], ab2);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nacc; i += blockDim.x) {
        const float v = acc[i];
        float *dst = i < 64 * F ? d_w1 + i : i < 64 * F + 64 ? d_b1 + (i - 64 * F) : i < 64 * F + 128 ? d_w2 + (i - 64 * F - 64) : d_b2;
        if (v != 0.0f) atomicAdd(dst, v);
    }
}

static int prop_train_args(PropTrain *p, const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, uint32_t M,
                           uint32_t F) {
    NLR_CHECK_ARG(feat && w1 && b1 && w2 && b2, "prop_mlp: NULL tensor");
    NLR_CHECK_ARG(F >= 1 && F <= NLR_PROP_FMAX, "prop_mlp: %u features outside [1,%d]", F, NLR_PROP_FMAX);
    p->feat = feat;
    p->w1 = w1;
    p->b1 = b1;
    p->w2 = w2;
    p->b2 = b2;
    p->M = M;
    p->F = F;
    return NLR_OK;
}

extern "C" int nlr_prop_mlp_forward(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, uint32_t M, uint32_t F,
                                    float *raw, void *stream) {
    if (M == 0) return NLR_OK;
    PropTrain p;
    int rc = prop_train_args(&p, feat, w1, b1, w2, b2, M, F);
    if (rc) return rc;
    NLR_CHECK_ARG(raw, "prop_mlp_forward: NULL output");
    const unsigned blocks = (M + 255) / 256;
    hipLaunchKernelGGL(nlr_prop_mlp_fwd_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, (hipStream_t)stream, p, raw);
    NLR_LAUNCH_CHECK("nlr_prop_mlp_fwd_kernel");
    return NLR_OK;
}

extern "C" int nlr_prop_mlp_backward(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, const float *g_raw,
                                     uint32_t M, uint32_t F, float *d_feat, float *d_w1, float *d_b1, float *d_w2, float *d_b2, void *stream) {
    PropTrain p;
    int rc = prop_train_args(&p, feat, w1, b1, w2, b2, M, F);
    if (rc) return rc;
    NLR_CHECK_ARG(g_raw && d_w1 && d_b1 && d_w2 && d_b2, "prop_mlp_backward: NULL tensor");
    hipStream_t st = (hipStream_t)stream;
    NLR_HIP(hipMemsetAsync(d_w1, 0, (size_t)64 * F * sizeof(float), st));
    NLR_HIP(hipMemsetAsync(d_b1, 0, 64 * sizeof(float), st));
    NLR_HIP(hipMemsetAsync(d_w2, 0, 64 * sizeof(float), st));
    NLR_HIP(hipMemsetAsync(d_b2, 0, sizeof(float), st));
    if (M == 0) return NLR_OK;
    if (d_feat) {
        const unsigned blocks = (M + 255) / 256;
        hipLaunchKernelGGL(nlr_prop_mlp_bwd_feat_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, st, p, g_raw, d_feat);
        NLR_LAUNCH_CHECK("nlr_prop_mlp_bwd_feat_kernel");
    }
    const unsigned chunks = (M + NLR_PROP_CHUNK - 1) / NLR_PROP_CHUNK;
    hipLaunchKernelGGL(nlr_prop_mlp_bwd_param_kernel, dim3(chunks < 256 ? chunks : 256), dim3(256), 0, st, p, g_raw, d_w1, d_b1, d_w2, d_b2);
    NLR_LAUNCH_CHECK("nlr_prop_mlp_bwd_param_kernel");
    return NLR_OK;
}
