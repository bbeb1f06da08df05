// Proposal resampling for one level: max_dilate_weights -> logits -> softmax -> CDF ->
// inverse-CDF interval sampling -> s_to_t.   One 64-lane wavefront per ray, all per-ray step
// function state staged in LDS.
//
// Replaces (rows a-2, a-3, a-4 of the scope table):
//   ZI/stepfun.py:75-105  max_dilate / max_dilate_weights      (O(S*S') mask tensor in the reference)
//   ZI/models.py:343-355  anneal + logits with -inf for zero-width bins
//   ZI/stepfun.py:108-128,154-161,175-218,251-294  integrate_weights / invert_cdf / sample / sample_intervals
//   ZI/math.py:89-108     sorted_interp (index form: xp, fp are non-decreasing)
//   ZI/coord.py:103-162   power_transformation ray warp (s_to_t)
// The reference materialises [N, 3S'+1, S'] and [N, S'+1, S] boolean masks in HBM; here a ray's
// <= 3*512+1 fenceposts live in LDS, the sort is a 3-way merge by binary-search ranking, the dilation maximum a
// range-max table query, the CDF a wavefront scan and each sample does a binary search.
// Precondition (as in the reference's step functions): prev_sdist is non-decreasing along each ray.
#include "nlr_kernels.h"

struct ResampleParams {
    const float *prev_sdist;    // [N, n_prev+1] or null (level 0)
    const float *prev_weights;  // [N, n_prev]
    uint32_t n_prev;
    float dilation;             // <= 0: no dilation
    float anneal, pad;
    uint32_t S;                 // intervals to draw
    const float *u;             // dev [S] sample positions (nlr_sample_u)
    const float *jitter;        // dev [N] uniform draws or null
    float max_jitter;
    const float *near, *far;    // [N]
    float lam, lam1, c_fwd, inv_exp;  // lambda, |lambda-1|, lam1/lambda, 1/lambda (as float)
    uint32_t N;
    float *sdist, *tdist;       // [N, S+1]
};

// ZI/coord.py:103-108 with x -> 2x (coord.py:145)
__device__ __forceinline__ float nlr_warp_fwd(float x, float lam, float lam1, float c_fwd) {
    return c_fwd * (powf((x * 2.0f) / lam1 + 1.0f, lam) - 1.0f);
}
// ZI/coord.py:111-118 followed by /2 (coord.py:146)
__device__ __forceinline__ float nlr_warp_inv(float y, float lam, float lam1, float inv_exp) {
    return ((powf(((y * lam) / lam1 + 1.0f) + NLR_EPS, inv_exp) - 1.0f) * lam1) / 2.0f;
}

// LDS carve (floats): t[n+1] | p[n] | U[3n+2] | W[3n] | Mx[(LV-1) n]  (dilate)  then  cw[m+1] | cen[S]
__global__ void __launch_bounds__(64) nlr_resample_kernel(ResampleParams P) {
    extern __shared__ __align__(16) float lds[];
    const uint32_t ray = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t n = P.n_prev;
    const bool dilate = (n > 0) && (P.dilation > 0.0f);

    float *tt;  // fenceposts of the step function being sampled, m+1 entries
    float *ww;  // its bin weights, m entries
    uint32_t m;
    float *scratch;

    if (n == 0) {
        tt = lds;
        ww = lds + 2;
        if (lane == 0) {
            tt[0] = 0.0f;
            tt[1] = 1.0f;
            ww[0] = 1.0f;
        }
        m = 1;
        scratch = lds + 4;
    } else {
        float *t = lds;           // n+1
        float *p = t + (n + 1);   // n
        const float *ps = P.prev_sdist + (size_t)ray * (n + 1);
        const float *pw = P.prev_weights + (size_t)ray * n;
        for (uint32_t i = lane; i <= n; i += 64) t[i] = ps[i];
        for (uint32_t i = lane; i < n; i += 64) p[i] = pw[i];
        __syncthreads();
        if (!dilate) {
            tt = t;
            ww = p;
            m = n;
            scratch = p + n;
        } else {
            float *U = p + n;            // 3n+2
            float *W = U + (3 * n + 2);  // 3n
            float *Mx = W + 3 * n;       // (LV-1) * n: sparse table of range maxima over p (level 0 is p itself)
            const float d = P.dilation;
            // weight_to_pdf (stepfun.py:64-67)
            for (uint32_t j = lane; j < n; j += 64) p[j] = p[j] / fmaxf(t[j + 1] - t[j], NLR_EPS);
            // sort(cat([t, t0, t1])) (stepfun.py:77-79) with t0_j = t_j - d, t1_j = t_{j+1} + d.  The three lists are
            // each non-decreasing (t is a step function's fenceposts; float add/sub of a constant is monotone), so the
            // sort is a 3-way merge: an element's position is its index in its own list plus the number of elements of
            // the other two lists that precede it (ties: t before t0 before t1), two binary searches per element
            // instead of a 36-pass bitonic network.  The sorted VALUES are what a full sort gives.
            auto count = [&](uint32_t len, auto pred) {  // #{i < len : pred(i)}, pred true on a prefix
                uint32_t lo = 0, hi = len;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pred(mid)) lo = mid + 1; else hi = mid;
                }
                return lo;
            };
            for (uint32_t i = lane; i <= 3 * n; i += 64) {
                float v;
                uint32_t r;
                if (i <= n) {
                    v = t[i];
                    r = i + count(n, [&](uint32_t q) { return (t[q] - d) < v; }) + count(n, [&](uint32_t q) { return (t[q + 1] + d) < v; });
                } else if (i <= 2 * n) {
                    const uint32_t o = i - (n + 1);
                    v = t[o] - d;
                    r = o + count(n + 1, [&](uint32_t q) { return t[q] <= v; }) + count(n, [&](uint32_t q) { return (t[q + 1] + d) < v; });
                } else {
                    const uint32_t o = i - (2 * n + 1);
                    v = t[o + 1] + d;
                    r = o + count(n + 1, [&](uint32_t q) { return t[q] <= v; }) + count(n, [&](uint32_t q) { return (t[q] - d) <= v; });
                }
                U[r] = fminf(fmaxf(v, 0.0f), 1.0f);  // clip to the domain [0,1] (stepfun.py:80)
            }
            // range-maximum table: Mx[l-1][j] = max p[j .. j + 2^l - 1] (indices clamped to n-1)
            uint32_t LV = 1;
            while ((2u << (LV - 1)) <= n) ++LV;  // levels 0 .. LV-1, 2^(LV-1) <= n
            __syncthreads();
            for (uint32_t l = 1; l < LV; ++l) {
                const float *src = l == 1 ? p : Mx + (size_t)(l - 2) * n;
                float *dst = Mx + (size_t)(l - 1) * n;
                const uint32_t h2 = 1u << (l - 1);
                for (uint32_t j = lane; j < n; j += 64) {
                    const uint32_t j2 = j + h2 < n ? j + h2 : n - 1;
                    dst[j] = fmaxf(src[j], src[j2]);
                }
                __syncthreads();
            }
            // max over covering intervals (stepfun.py:81-87): interval j covers t_k iff (t_j - d) <= t_k < (t_{j+1} + d);
            // both bounds are monotone in j, so the covering set is the index range [j0, j1) found by two binary searches
            // and its maximum comes from the table (the reference builds an [3n+1, n] mask).  Then pdf_to_weight
            // (stepfun.py:70-72).
            float part = 0.0f;
            for (uint32_t k = lane; k < 3 * n; k += 64) {
                const float tk = U[k];
                const uint32_t j1 = count(n, [&](uint32_t q) { return (t[q] - d) <= tk; });
                const uint32_t j0 = count(n, [&](uint32_t q) { return (t[q + 1] + d) <= tk; });
                float best = 0.0f;
                if (j0 < j1) {
                    const uint32_t len = j1 - j0, l = 31u - (uint32_t)__clz((int)len);
                    const float *row = l == 0 ? p : Mx + (size_t)(l - 1) * n;
                    best = fmaxf(best, fmaxf(row[j0], row[j1 - (1u << l)]));
                }
                const float wk = best * (U[k + 1] - tk);
                W[k] = wk;
                part += wk;
            }
            const float tot = fmaxf(nlr_wave_sum(part), NLR_EPS);  // renormalize (stepfun.py:103-104)
            __syncthreads();
            for (uint32_t k = lane; k < 3 * n; k += 64) W[k] = W[k] / tot;
            __syncthreads();
            // caller drops the first and last fencepost / weight (models.py:339-340)
            tt = U + 1;
            ww = W + 1;
            m = 3 * n - 2;
            scratch = Mx + (size_t)(LV - 1) * n;
        }
    }
    __syncthreads();

    // logits -> softmax (models.py:352-355, stepfun.py:157).  e = exp(logit - max) held in cw[1..m].
    float *cw = scratch;          // m+1
    float *cen = cw + (m + 1);    // S
    float mx = -INFINITY;
    for (uint32_t k = lane; k < m; k += 64) {
        const float lg = (tt[k + 1] > tt[k]) ? P.anneal * logf(ww[k] + P.pad) : -INFINITY;
        cw[k + 1] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = nlr_wave_max(mx);
    float se = 0.0f;
    for (uint32_t k = lane; k < m; k += 64) {
        const float e = expf(cw[k + 1] - mx);
        cw[k + 1] = e;
        se += e;
    }
    se = nlr_wave_sum(se);
    __syncthreads();
    // integrate_weights (stepfun.py:123-127): cw_0 = 0, cw_j = min(cumsum(w)[j-1], 1) for j<m, cw_m = 1.
    // Lane l owns the contiguous chunk [l*per, (l+1)*per) of the m weights.
    {
        const uint32_t per = (m + 63) / 64;
        const uint32_t b0 = lane * per;
        float run = 0.0f;
        for (uint32_t i = 0; i < per; ++i) {
            const uint32_t k = b0 + i;
            if (k < m) run += cw[k + 1] / se;
        }
        const float incl = nlr_wave_incl_scan_add(run, lane);
        float base = incl - run;
        // a float tree-scan is not guaranteed monotone to the last ulp; sorted_interp needs xp sorted
        float prevmax = nlr_wave_incl_scan_max(incl, lane);
        prevmax = __shfl_up(prevmax, 1, 64);
        if (lane == 0) prevmax = 0.0f;
        __syncthreads();
        float acc = base;
        float hi = prevmax;
        for (uint32_t i = 0; i < per; ++i) {
            const uint32_t k = b0 + i;
            if (k < m) {
                acc += cw[k + 1] / se;
                hi = fmaxf(hi, acc);
                cw[k + 1] = (k + 1 == m) ? 1.0f : fminf(hi, 1.0f);
            }
        }
        if (lane == 0) cw[0] = 0.0f;
    }
    __syncthreads();

    // invert the CDF at u (stepfun.py:160, math.py:89-108)
    const float jit = P.jitter ? P.jitter[ray] * P.max_jitter : 0.0f;
    for (uint32_t k = lane; k < P.S; k += 64) {
        const float u = P.jitter ? P.u[k] + jit : P.u[k];
        // hi = #{j in [0,m] : cw_j <= u}
        uint32_t lo = 0, hi = m + 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cw[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const uint32_t i0 = lo > 0 ? (lo - 1 > m ? m : lo - 1) : 0;
        const uint32_t i1 = lo > m ? m : lo;
        const float x0 = cw[i0], x1 = cw[i1];
        float off = (u - x0) / (x1 - x0);
        if (off != off) off = 0.0f;
        off = fminf(fmaxf(off, 0.0f), 1.0f);
        cen[k] = tt[i0] + off * (tt[i1] - tt[i0]);
    }
    __syncthreads();

    // fenceposts: midpoints + reflected, clamped ends (stepfun.py:284-293), then s_to_t
    const float nearv = P.near[ray], farv = P.far[ray];
    const float s_near = nlr_warp_fwd(nearv, P.lam, P.lam1, P.c_fwd);
    const float s_far = nlr_warp_fwd(farv, P.lam, P.lam1, P.c_fwd);
    float *so = P.sdist + (size_t)ray * (P.S + 1);
    float *to = P.tdist ? P.tdist + (size_t)ray * (P.S + 1) : nullptr;
    for (uint32_t k = lane; k <= P.S; k += 64) {
        float s;
        if (k == 0) {
            const float mid0 = (cen[1] + cen[0]) / 2.0f;
            s = fmaxf(2.0f * cen[0] - mid0, 0.0f);
        } else if (k == P.S) {
            const float midl = (cen[P.S - 1] + cen[P.S - 2]) / 2.0f;
            s = fminf(2.0f * cen[P.S - 1] - midl, 1.0f);
        } else {
            s = (cen[k] + cen[k - 1]) / 2.0f;
        }
        so[k] = s;
        if (to) to[k] = nlr_warp_inv(s * s_far + (1.0f - s) * s_near, P.lam, P.lam1, P.inv_exp);
    }
}

// torch.linspace in float32 (ATen RangeFactories: symmetric fill from both ends).
static void linspace_f32(float start, float end, uint32_t n, float *out) {
    if (n == 1) {
        out[0] = start;
        return;
    }
    const float step = (end - start) / (float)(n - 1);
    const uint32_t half = n / 2;
    for (uint32_t i = 0; i < n; ++i) out[i] = i < half ? start + step * (float)i : end - step * (float)(n - 1 - i);
}

extern "C" void nlr_sample_u(uint32_t n, int rand, float *u_host, float *max_jitter) {
    const double eps = (double)NLR_EPS;
    if (!rand) {
        const double pad = 1.0 / (2.0 * n);  // python floats (double) then cast by torch.linspace
        linspace_f32((float)pad, (float)(1.0 - pad - eps), n, u_host);
        if (max_jitter) *max_jitter = 0.0f;
    } else {
        const double u_max = eps + (1.0 - eps) / n;
        linspace_f32(0.0f, (float)(1.0 - u_max), n, u_host);
        if (max_jitter) *max_jitter = (float)((1.0 - u_max) / (n - 1) - eps);
    }
}

// Internal launcher shared by nlr_resample_level and nlr_render_rays (u_dev already uploaded).
int nlr_launch_resample(const float *prev_sdist, const float *prev_weights, uint32_t n_prev, float dilation, float anneal,
                        float pad, uint32_t S, const float *u_dev, const float *jitter, float max_jitter, const float *near,
                        const float *far, float lam, uint32_t N, float *sdist, float *tdist, hipStream_t st) {
    NLR_CHECK_ARG(S >= 2, "num_samples must be > 1, is %u", S);  // stepfun.py:271-272
    NLR_CHECK_ARG(S <= 1024 && n_prev <= 512, "resample: S=%u / n_prev=%u beyond the LDS budget", S, n_prev);
    NLR_CHECK_ARG((n_prev == 0) || (prev_sdist && prev_weights), "resample: previous level tensors are NULL");
    ResampleParams P;
    memset(&P, 0, sizeof(P));
    P.prev_sdist = prev_sdist;
    P.prev_weights = prev_weights;
    P.n_prev = n_prev;
    P.dilation = dilation;
    P.anneal = anneal;
    P.pad = pad;
    P.S = S;
    P.u = u_dev;
    P.jitter = jitter;
    P.max_jitter = max_jitter;
    P.near = near;
    P.far = far;
    P.lam = lam;
    P.lam1 = fabsf(lam - 1.0f);
    P.c_fwd = (float)(fabs((double)lam - 1.0) / (double)lam);
    P.inv_exp = (float)(1.0 / (double)lam);
    P.N = N;
    P.sdist = sdist;
    P.tdist = tdist;
    const bool dilate = n_prev > 0 && dilation > 0.0f;
    size_t fl;
    uint32_t m;
    if (n_prev == 0) {
        fl = 4;
        m = 1;
    } else if (!dilate) {
        fl = 2 * (size_t)n_prev + 1;
        m = n_prev;
    } else {
        uint32_t lv = 1;
        while ((2u << (lv - 1)) <= n_prev) ++lv;
        fl = 2 * (size_t)n_prev + 1 + (3 * (size_t)n_prev + 2) + 3 * (size_t)n_prev + (size_t)(lv - 1) * n_prev;
        m = 3 * n_prev - 2;
    }
    fl += (m + 1) + S + 8;
    const size_t lds_bytes = fl * sizeof(float);
    NLR_CHECK_ARG(lds_bytes <= 64 * 1024, "resample: LDS need %zu B > 64 KiB", lds_bytes);
    hipLaunchKernelGGL(nlr_resample_kernel, dim3(N), dim3(64), lds_bytes, st, P);
    NLR_LAUNCH_CHECK("nlr_resample_kernel");
    return NLR_OK;
}

extern "C" int nlr_resample_level(const float *prev_sdist, const float *prev_weights, uint32_t n_prev, float dilation,
                                  float anneal, float resample_padding, uint32_t num_samples, const float *rand_jitter,
                                  const float *near, const float *far, float power_lambda, uint32_t N, float *sdist,
                                  float *tdist, void *stream) {
    NLR_CHECK_ARG(near && far && sdist, "resample_level: NULL tensor");
    if (N == 0) return NLR_OK;
    NLR_CHECK_ARG(num_samples >= 2 && num_samples <= 1024, "num_samples must be in [2,1024], is %u", num_samples);
    hipStream_t st = (hipStream_t)stream;
    float uh[1024], mj = 0.0f;
    nlr_sample_u(num_samples, rand_jitter != nullptr, uh, &mj);
    float *u_dev = nullptr;
    NLR_HIP(hipMallocAsync((void **)&u_dev, num_samples * sizeof(float), st));
    NLR_HIP(hipMemcpyAsync(u_dev, uh, num_samples * sizeof(float), hipMemcpyHostToDevice, st));
    NLR_HIP(hipStreamSynchronize(st));  // uh is a stack buffer
    int rc = nlr_launch_resample(prev_sdist, prev_weights, n_prev, dilation, anneal, resample_padding, num_samples, u_dev,
                                 rand_jitter, mj, near, far, power_lambda, N, sdist, tdist, st);
    (void)hipFreeAsync(u_dev, st);
    return rc;
}
