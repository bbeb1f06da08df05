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
// <= 3*256+1 fenceposts live in LDS, the sort is an in-LDS bitonic network, the CDF is a
// wavefront scan and each sample does a binary search.
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
    uint32_t n2;                // bitonic size (pow2 >= 3 n_prev + 1) when dilating
};

// ZI/coord.py:103-108 with x -> 2x (coord.py:145)
__device__ __forceinline__ float nlr_warp_fwd(float x, float lam, float lam1, float c_fwd) {
    return c_fwd * (powf((x * 2.0f) / lam1 + 1.0f, lam) - 1.0f);
}
// ZI/coord.py:111-118 followed by /2 (coord.py:146)
__device__ __forceinline__ float nlr_warp_inv(float y, float lam, float lam1, float inv_exp) {
    return ((powf(((y * lam) / lam1 + 1.0f) + NLR_EPS, inv_exp) - 1.0f) * lam1) / 2.0f;
}

// LDS carve (floats): t[n+1] | p[n] | U[n2] | W[3n]  (dilate)  then  cw[m+1] | cen[S]
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
            float *U = p + n;        // n2
            float *W = U + P.n2;     // 3n
            const float d = P.dilation;
            // weight_to_pdf (stepfun.py:64-67)
            for (uint32_t j = lane; j < n; j += 64) p[j] = p[j] / fmaxf(t[j + 1] - t[j], NLR_EPS);
            // cat([t, t0, t1]) padded with +inf (stepfun.py:77-79)
            for (uint32_t i = lane; i < P.n2; i += 64) {
                float v;
                if (i <= n) v = t[i];
                else if (i <= 2 * n) v = t[i - (n + 1)] - d;        // t0_j = t_j - d
                else if (i <= 3 * n) v = t[i - (2 * n + 1) + 1] + d; // t1_j = t_{j+1} + d
                else v = INFINITY;
                U[i] = v;
            }
            __syncthreads();
            // bitonic sort, ascending
            for (uint32_t k = 2; k <= P.n2; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = lane; i < P.n2; i += 64) {
                        const uint32_t ixj = i ^ j;
                        if (ixj > i) {
                            const float a = U[i], b = U[ixj];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) {
                                U[i] = b;
                                U[ixj] = a;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            // clip to the domain [0,1] (stepfun.py:80)
            const uint32_t nd = 3 * n + 1;
            for (uint32_t i = lane; i < nd; i += 64) U[i] = fminf(fmaxf(U[i], 0.0f), 1.0f);
            __syncthreads();
            // max over covering intervals (stepfun.py:81-87), then pdf_to_weight (stepfun.py:70-72)
            float part = 0.0f;
            for (uint32_t k = lane; k < 3 * n; k += 64) {
                const float tk = U[k];
                float best = 0.0f;
                for (uint32_t j = 0; j < n; ++j) {
                    const bool in = ((t[j] - d) <= tk) && ((t[j + 1] + d) > tk);
                    best = fmaxf(best, in ? p[j] : 0.0f);
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
            scratch = W + 3 * n;
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

static uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
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
    P.n2 = dilate ? next_pow2(3 * n_prev + 1) : 0;
    size_t fl;
    uint32_t m;
    if (n_prev == 0) {
        fl = 4;
        m = 1;
    } else if (!dilate) {
        fl = 2 * (size_t)n_prev + 1;
        m = n_prev;
    } else {
        fl = 2 * (size_t)n_prev + 1 + P.n2 + 3 * (size_t)n_prev;
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
