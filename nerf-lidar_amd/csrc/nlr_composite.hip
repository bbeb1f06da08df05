// Alpha compositing: density -> weights (wavefront exclusive scan for transmittance) -> RGB /
// depth / semantic / intensity / acc / distance statistics, plus the LiDAR post-step.
//
// Replaces (rows a-13, a-14, a-16 of the scope table):
//   ZI/render.py:170-189   compute_alpha_weights
//   ZI/render.py:192-284   volumetric_rendering (+ ZI/stepfun.py:329-339 weighted_percentile)
//   Z/render_lidar.py:142-161  points = (o + depth*d)/scale_factor, labels = argmax(semantic)
// One 64-lane wavefront per ray, 4 rays per 256-thread workgroup; lane l owns the contiguous
// samples [l*per, (l+1)*per).  The reference materialises alpha, trans, weights, t_mids and the
// [N,S+2] percentile CDFs in HBM; here only the outputs leave the CU.
#include "nlr_kernels.h"


#define NLR_COMP_MAXK 32
#define NLR_COMP_MAXPER 8  // S <= 512

__global__ void __launch_bounds__(256) nlr_composite_kernel(CompositeParams P) {
    extern __shared__ __align__(16) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4 + wave;
    const bool active = ray < P.N;
    const uint32_t S = P.S;
    float *cw = lds + (size_t)wave * (2 * (S + 2));  // S+2
    float *ta = cw + (S + 2);                        // S+2
    const uint32_t per = (S + 63) / 64;
    const uint32_t k0 = lane * per;

    float dnorm = 0.0f;
    if (active) {
        const float dx = P.dirs[(size_t)ray * 3], dy = P.dirs[(size_t)ray * 3 + 1], dz = P.dirs[(size_t)ray * 3 + 2];
        dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
    const float *td = P.tdist + (size_t)(active ? ray : 0) * (S + 1);
    const float *dn = P.density + (size_t)(active ? ray : 0) * S;

    // density * delta, with an infinitely wide last interval when the background is opaque
    float dd[NLR_COMP_MAXPER], tm[NLR_COMP_MAXPER];
    float run = 0.0f;
#pragma unroll
    for (int i = 0; i < NLR_COMP_MAXPER; ++i) {
        const uint32_t k = k0 + i;
        dd[i] = 0.0f;
        tm[i] = 0.0f;
        if (i < (int)per && k < S && active) {
            const float ta0 = td[k], ta1 = td[k + 1];
            float v = dn[k] * ((ta1 - ta0) * dnorm);
            if (P.opaque && k == S - 1) v = INFINITY;
            dd[i] = v;
            tm[i] = 0.5f * (ta0 + ta1);
            run += v;
        }
    }
    // exclusive prefix over lanes; taken from the previous lane's inclusive value so that the
    // +inf of the last interval never meets a subtraction
    const float incl = nlr_wave_incl_scan_add(run, lane);
    float base = __shfl_up(incl, 1, 64);
    if (lane == 0) base = 0.0f;

    float w[NLR_COMP_MAXPER];
    float acc = 0.0f, sdep = 0.0f, slog = 0.0f, sint = 0.0f, srgb[3] = {0.0f, 0.0f, 0.0f};
    float ssem[NLR_COMP_MAXK];
#pragma unroll
    for (int c = 0; c < NLR_COMP_MAXK; ++c) ssem[c] = 0.0f;
    float cum = base, wrun = 0.0f;
#pragma unroll
    for (int i = 0; i < NLR_COMP_MAXPER; ++i) {
        const uint32_t k = k0 + i;
        w[i] = 0.0f;
        if (i < (int)per && k < S && active) {
            const float alpha = 1.0f - expf(-dd[i]);
            const float trans = expf(-cum);
            const float wk = alpha * trans;
            cum += dd[i];
            w[i] = wk;
            wrun += wk;
            acc += wk;
            sdep += wk * tm[i];
            if (P.extras) slog += wk * logf(tm[i]);
            if (P.weights) P.weights[(size_t)ray * S + k] = wk;
            if (P.rgb) {
                const size_t mi = (size_t)ray * S + k, Mt = (size_t)P.N * S;  // channel-major [3, N*S]
                srgb[0] += wk * P.rgb[mi];
                srgb[1] += wk * P.rgb[Mt + mi];
                srgb[2] += wk * P.rgb[2 * Mt + mi];
            }
            if (P.inten) sint += wk * P.inten[(size_t)ray * S + k];
            if (P.sem) {
                const float *ps = P.sem + (size_t)ray * S + k;  // class-major [K, N*S]
                const size_t Mt = (size_t)P.N * S;
#pragma unroll
                for (int c = 0; c < NLR_COMP_MAXK; ++c)
                    if (c < (int)P.K) ssem[c] += wk * ps[(size_t)c * Mt];
            }
        }
    }
    acc = nlr_wave_sum(acc);
    sdep = nlr_wave_sum(sdep);
    const float accc = fmaxf(acc, NLR_EPS);
    const float depth = sdep / accc;
    const float bgw = fmaxf(1.0f - acc, 0.0f);
    int label = 0;
    float segv = 0.0f;  // segment mode: lane c < 32 holds slot c of the ray's record
    if (P.seg) {
        // The MLP kernel (compositing mode) has already summed w' x value inside every 32-sample segment, with w' = the alpha
        // weight relative to the segment start; the ray's value is sum_s T_s * record_s with T_s = exp(-sum of sigma*delta in
        // front of segment s) = exp(-base) of the lane that owns the segment's first sample (32 % per == 0: checked by the host).
        const float tseg = expf(-base);
        const uint32_t nseg = S >> 5;
        const float *rec = P.seg + (size_t)(active ? ray : 0) * nseg * 32 + (lane & 31);
        for (uint32_t s = 0; s < nseg; ++s) segv += __shfl(tseg, (int)((32 * s) / per), 64) * rec[(size_t)s * 32];
        // label = first maximum over slots [0, K)
        float bv = ((uint32_t)(lane & 31) < P.K) ? segv : -INFINITY;
        int bi = lane & 31;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(bv, d, 64);
            const int oi = __shfl_xor(bi, d, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        label = bi;
        sint = __shfl(segv, (int)P.K, 64);
#pragma unroll
        for (int c = 0; c < 3; ++c) srgb[c] = __shfl(segv, 29 + c, 64);
    } else if (P.rgb || P.o_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) srgb[c] = nlr_wave_sum(srgb[c]);
    }
    if (P.inten) sint = nlr_wave_sum(sint);
    if (P.sem) {
        float best = -INFINITY;
#pragma unroll
        for (int c = 0; c < NLR_COMP_MAXK; ++c) {
            if (c < (int)P.K) {
                ssem[c] = nlr_wave_sum(ssem[c]);
                if (ssem[c] > best) {  // np.argmax: first maximum
                    best = ssem[c];
                    label = c;
                }
            }
        }
    }
    if (active && lane == 0) {
        if (P.o_depth) P.o_depth[ray] = depth;
        if (P.level_depth) P.level_depth[ray] = depth;
        if (P.o_rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) P.o_rgb[(size_t)ray * 3 + c] = srgb[c] + bgw * P.bg;
        }
        if (P.o_int && (P.inten || (P.seg && P.seg_int))) P.o_int[ray] = sint;
        if (P.o_sem && P.sem) {
#pragma unroll
            for (int c = 0; c < NLR_COMP_MAXK; ++c)
                if (c < (int)P.K) P.o_sem[(size_t)ray * P.K + c] = ssem[c];
        }
        if (P.o_labels && (P.sem || (P.seg && P.K))) P.o_labels[ray] = label;
        if (P.o_points && P.origins) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                P.o_points[(size_t)ray * 3 + c] = (P.origins[(size_t)ray * 3 + c] + depth * P.dirs[(size_t)ray * 3 + c]) / P.scale_factor;
        }
        if (P.o_acc) P.o_acc[ray] = acc;
        if (P.o_packed) {
            const uint32_t row = P.pk_h ? (ray % P.pk_w) * P.pk_h + ray / P.pk_w : ray;
            float *pk = P.o_packed + (size_t)row * 7;
            pk[0] = depth;
            pk[1] = (P.inten || (P.seg && P.seg_int)) ? sint : 0.0f;
            pk[2] = acc;
#pragma unroll
            for (int c = 0; c < 3; ++c) pk[3 + c] = srgb[c] + bgw * P.bg;
            pk[6] = (float)label;
        }
    }
    if (P.seg && active && P.o_sem && (uint32_t)lane < P.K) P.o_sem[(size_t)ray * P.K + lane] = segv;
    if (!P.extras) return;  // uniform across the block

    // distance_mean (render.py:266-269)
    slog = nlr_wave_sum(slog);
    if (active && lane == 0 && P.o_dmean) {
        float v = expf(slog / accc);
        if (v != v) v = INFINITY;  // nan_to_num(., nan=inf)
        P.o_dmean[ray] = fminf(fmaxf(v, td[0]), td[S]);
    }
    // weighted percentiles over [tdist, far] with weights [w, bg_w] (render.py:274-282)
    {
        const float wincl = nlr_wave_incl_scan_add(wrun, lane);
        float c = wincl - wrun;
        float hi = nlr_wave_incl_scan_max(wincl, lane);
        hi = __shfl_up(hi, 1, 64);
        if (lane == 0) hi = 0.0f;
#pragma unroll
        for (int i = 0; i < NLR_COMP_MAXPER; ++i) {
            const uint32_t k = k0 + i;
            if (i < (int)per && k < S && active) {
                c += w[i];
                hi = fmaxf(hi, c);
                cw[k + 1] = fminf(hi, 1.0f);  // cw_j = min(sum_{i<j} w_i, 1), j = k+1
                ta[k] = td[k];
            }
        }
        if (active && lane == 0) {
            cw[0] = 0.0f;
            cw[S + 1] = 1.0f;
            ta[S] = td[S];
            ta[S + 1] = P.far[ray];
        }
    }
    __syncthreads();
    if (active && lane < 3) {
        const float p = lane == 0 ? 0.05f : (lane == 1 ? 0.5f : 0.95f);  // tensor([5,50,95]) / 100
        const uint32_t m = S + 1;
        uint32_t lo = 0, hi = m + 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cw[mid] <= p) lo = mid + 1; else hi = mid;
        }
        const uint32_t i0 = lo > 0 ? (lo - 1 > m ? m : lo - 1) : 0;
        const uint32_t i1 = lo > m ? m : lo;
        float off = (p - cw[i0]) / (cw[i1] - cw[i0]);
        if (off != off) off = 0.0f;
        off = fminf(fmaxf(off, 0.0f), 1.0f);
        const float v = ta[i0] + off * (ta[i1] - ta[i0]);
        float *dst = lane == 0 ? P.o_p5 : (lane == 1 ? P.o_dmed : P.o_p95);
        if (dst) dst[ray] = v;
    }
}

int nlr_launch_composite(const CompositeParams &P, hipStream_t st) {
    NLR_CHECK_ARG(P.S >= 1 && P.S <= 64 * NLR_COMP_MAXPER, "composite: S=%u outside [1,%d]", P.S, 64 * NLR_COMP_MAXPER);
    NLR_CHECK_ARG(P.K <= NLR_COMP_MAXK, "composite: class_num %u > %d", P.K, NLR_COMP_MAXK);
    NLR_CHECK_ARG(P.density && P.tdist && P.dirs, "composite: NULL density/tdist/directions");
    NLR_CHECK_ARG(!P.seg || (P.S % 32 == 0 && 32 % ((P.S + 63) / 64) == 0 && P.K + P.seg_int <= 29), "composite: segment records do not fit S = %u, K = %u", P.S, P.K);
    NLR_CHECK_ARG(!P.extras || P.far, "composite: compute_extras needs the far plane");
    const size_t lds = (size_t)4 * 2 * (P.S + 2) * sizeof(float);
    hipLaunchKernelGGL(nlr_composite_kernel, dim3((P.N + 3) / 4), dim3(256), lds, st, P);
    NLR_LAUNCH_CHECK("nlr_composite_kernel");
    return NLR_OK;
}

extern "C" int nlr_composite_level(const float *density, const float *tdist, const float *directions, const float *rgb,
                                   const float *semantic, const float *intensity, const float *far, const float *origins,
                                   uint32_t N, uint32_t S, uint32_t class_num, int opaque_background, float bg,
                                   int compute_extras, float scale_factor, float *weights, const NlrOut *out,
                                   float *level_depth, void *stream) {
    if (N == 0) return NLR_OK;
    CompositeParams P;
    memset(&P, 0, sizeof(P));
    P.density = density;
    P.tdist = tdist;
    P.dirs = directions;
    P.rgb = rgb;
    P.sem = semantic;
    P.inten = intensity;
    P.far = far;
    P.origins = origins;
    P.N = N;
    P.S = S;
    P.K = semantic ? class_num : 0;
    P.opaque = opaque_background;
    P.extras = compute_extras;
    P.bg = bg;
    P.scale_factor = scale_factor > 0 ? scale_factor : 1.0f;
    P.weights = weights;
    P.level_depth = level_depth;
    if (out) {
        P.o_rgb = out->rgb;
        P.o_depth = out->depth;
        P.o_sem = out->semantic;
        P.o_int = out->intensity;
        P.o_acc = out->acc;
        P.o_dmean = out->distance_mean;
        P.o_dmed = out->distance_median;
        P.o_p5 = out->distance_percentile_5;
        P.o_p95 = out->distance_percentile_95;
        P.o_labels = out->labels;
        P.o_points = scale_factor > 0 ? out->points : nullptr;
        P.o_packed = out->packed;
        P.pk_h = out->packed_h;
        P.pk_w = out->packed_w;
        if (out->packed && out->packed_h)
            NLR_CHECK_ARG(out->packed_w > 0 && (uint64_t)out->packed_h * out->packed_w == N,
                          "composite: packed tile %u x %u does not match N = %u rays", out->packed_h, out->packed_w, N);
    }
    return nlr_launch_composite(P, (hipStream_t)stream);
}

// Compositing-mode tail of nlr_render_rays: weights / depth / acc / percentiles from density and tdist as above, the per-ray rgb /
// semantic / intensity from the segment records nlr_mlp_kernel<..., COMP = true> wrote (nlr_mlp_kernel.h).
int nlr_composite_segments(const float *density, const float *tdist, const float *directions, const float *seg, uint32_t class_num,
                           int has_intensity, const float *far, const float *origins, uint32_t N, uint32_t S, int opaque_background,
                           float bg, int compute_extras, float scale_factor, float *weights, const NlrOut *out, float *level_depth,
                           hipStream_t st) {
    if (N == 0) return NLR_OK;
    NLR_CHECK_ARG(seg && out, "composite_segments: NULL argument");
    CompositeParams P;
    memset(&P, 0, sizeof(P));
    P.density = density;
    P.tdist = tdist;
    P.dirs = directions;
    P.seg = seg;
    P.seg_int = has_intensity ? 1 : 0;
    P.far = far;
    P.origins = origins;
    P.N = N;
    P.S = S;
    P.K = class_num;
    P.opaque = opaque_background;
    P.extras = compute_extras;
    P.bg = bg;
    P.scale_factor = scale_factor > 0 ? scale_factor : 1.0f;
    P.weights = weights;
    P.level_depth = level_depth;
    P.o_rgb = out->rgb;
    P.o_depth = out->depth;
    P.o_sem = out->semantic;
    P.o_int = out->intensity;
    P.o_acc = out->acc;
    P.o_dmean = out->distance_mean;
    P.o_dmed = out->distance_median;
    P.o_p5 = out->distance_percentile_5;
    P.o_p95 = out->distance_percentile_95;
    P.o_labels = out->labels;
    P.o_points = scale_factor > 0 ? out->points : nullptr;
    P.o_packed = out->packed;
    P.pk_h = out->packed_h;
    P.pk_w = out->packed_w;
    if (out->packed && out->packed_h)
        NLR_CHECK_ARG(out->packed_w > 0 && (uint64_t)out->packed_h * out->packed_w == N,
                      "composite: packed tile %u x %u does not match N = %u rays", out->packed_h, out->packed_w, N);
    return nlr_launch_composite(P, st);
}
