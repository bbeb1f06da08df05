// Host side of libnerflidar_hip.so: error plumbing, model object (weight packing), workspace carving and
// the level loop of Model.forward (ZI/models.py:316-557) as a chain of kernel launches on one stream.
#include <stdarg.h>

#include <vector>

#include "nlr_kernels.h"

#include <stdlib.h>
#include "nlr_objects.h"


static thread_local char g_err[512] = "";
void nlr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *nlr_last_error(void) { return g_err; }
extern "C" int nlr_version(void) { return 100; }
extern "C" const char *nlr_kernel_names(void) {
    return "nlr_resample_kernel,nlr_prop8_kernel,nlr_encode8_kernel,nlr_direnc_kernel,nlr_mlp_kernel,nlr_composite_kernel";
}

// ---- optional per-kernel event timing ---------------------------------------------------------------
struct Profile {
    bool armed = false;
    uint32_t mask = 0xffffffffu;  // bit k set: kernels of kind NLR_K_k are bracketed
    std::vector<hipEvent_t> ev;  // start/stop pairs
    std::vector<int> kind;
};
struct ProfScope {  // RAII bracket around one launch
    Profile *p;
    hipStream_t st;
    ProfScope(Profile *prof, int kind, hipStream_t s) : p(prof && prof->armed && ((prof->mask >> kind) & 1u) ? prof : nullptr), st(s) {
        if (!p) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { p = nullptr; return; }
        p->ev.push_back(a);
        p->ev.push_back(b);
        p->kind.push_back(kind);
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (p) (void)hipEventRecord(p->ev.back(), st);
    }
};

// ---- model ------------------------------------------------------------------------------------------
struct LevelModel {
    bool is_prop = false;
    GridParams gp;
    uint32_t F = 0, S = 0;
    int re_weights = 1;
    float density_bias = -1.0f;
    // proposal MLP (fp32 VALU)
    float *p_w1 = nullptr, *p_b1 = nullptr, *p_w2 = nullptr;
    float p_b2 = 0.0f;
    // NerfMLP (matrix cores)
    uint32_t W = 0, WB = 0, D = 0, HT = 0, K = 0, int_row = 0xffffffffu, deg = 0, E = 0;
    bool use_int = false;
    uint32_t prec = NLR_PREC_MIXED;
    void *tape = nullptr;
    uint32_t tape_chunks = 0;
    float rgb_premul = 1.0f, rgb_bias = 0.0f, rgb_padding = 0.001f;
    float *bias_all = nullptr;
    uint32_t bias_count = 0;
    float *u_det = nullptr, *u_rand = nullptr;                            // sample positions [S]
    float max_jitter = 0.0f;
};

struct NlrModel {
    uint32_t num_levels = 0;
    LevelModel lv[NLR_MAX_LEVELS];
    float dilation_multiplier, dilation_bias, anneal_slope, resample_padding, power_lambda, std_scale, bg;
    uint32_t opaque, prec;
    int device = 0;
    uint32_t cus = 0;          // compute units of `device`: the MLP kernel runs one persistent workgroup per CU
    hipStream_t up = nullptr;  // stream the weight uploads of nlr_model_create are enqueued on
    std::vector<void *> allocs;
    mutable Profile prof;
};

static int dev_upload(NlrModel *m, const void *host, size_t bytes, void **out) {
    void *p = nullptr;
    NLR_HIP(hipMalloc(&p, bytes ? bytes : 16));
    m->allocs.push_back(p);
    if (bytes) {  // the sources are temporaries of the caller: complete the copy before returning
        NLR_HIP(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, m->up));
        NLR_HIP(hipStreamSynchronize(m->up));
    }
    *out = p;
    return NLR_OK;
}

static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                                // round to nearest even
    return (uint16_t)(u >> 16);
}

// Dense row-major matrix view used while fusing / slicing the reference's Linear layers.
struct Mat {
    std::vector<float> a;
    uint32_t rows = 0, cols = 0;
    Mat() {}
    Mat(uint32_t r, uint32_t c) : a((size_t)r * c, 0.0f), rows(r), cols(c) {}
    float &at(uint32_t r, uint32_t c) { return a[(size_t)r * cols + c]; }
    float get(uint32_t r, uint32_t c) const { return (r < rows && c < cols) ? a[(size_t)r * cols + c] : 0.0f; }
};

static Mat mat_from(const NlrLinear &l, uint32_t col0, uint32_t ncols) {
    Mat m(l.out_features, ncols);
    for (uint32_t r = 0; r < l.out_features; ++r)
        for (uint32_t c = 0; c < ncols; ++c) m.at(r, c) = l.weight[(size_t)r * l.in_features + col0 + c];
    return m;
}

// ---- weight fragments (layouts: nlr_mlp_kernel.h).  One fragment = 1 KiB = 64 lanes x 16 B, the A operand of one MFMA
// step for 16 output rows; lane = (m = lane & 15, q = lane >> 4).
//   bf16  (v_mfma_f32_16x16x32_bf16): rows 16R + m, k-block g (32 input features): element j = W[16R + m][32g + 16(j>>2) + 4q + (j&3)]
//         - the k order in which two 16-row accumulator tiles of the previous layer, converted pairwise, form the B operand
//   f32   (v_mfma_f32_16x16x4_f32, 4 k-steps per fragment): rows 16R + m, input row block J (16 features): element e = W[16R + m][16J + 4q + e]
// GEMM order on the tape: output unit o (32 rows) -> k-group g -> row block j (R = 2o + j; RH = 1: R = o, only the first 16 rows
// of the unit exist) [-> hi, lo fragment of the split-bf16 form].
enum { TAPE_F32 = 0, TAPE_BF16 = 1, TAPE_X3 = 2 };
#define NLR_TAPE_CHUNK 32768
struct TapeBuilder {
    std::vector<uint8_t> bytes;
    size_t frags() const { return bytes.size() / 1024; }
    void frag_bf16(const Mat &w, uint32_t R, uint32_t g, bool lo_part) {
        uint16_t f[64 * 8];
        for (uint32_t lane = 0; lane < 64; ++lane)
            for (uint32_t j = 0; j < 8; ++j) {
                const float v = w.get(16 * R + (lane & 15), 32 * g + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3));
                const uint16_t hi = f32_to_bf16(v);
                if (!lo_part) {
                    f[lane * 8 + j] = hi;
                } else {  // W = hi + lo
                    uint32_t hb = (uint32_t)hi << 16;
                    float hf;
                    memcpy(&hf, &hb, 4);
                    f[lane * 8 + j] = f32_to_bf16(v - hf);
                }
            }
        bytes.insert(bytes.end(), (const uint8_t *)f, (const uint8_t *)f + sizeof(f));
    }
    void frag_f32(const Mat &w, uint32_t R, uint32_t J) {
        float f[64 * 4];
        for (uint32_t lane = 0; lane < 64; ++lane)
            for (uint32_t e = 0; e < 4; ++e) f[lane * 4 + e] = w.get(16 * R + (lane & 15), 16 * J + 4 * (lane >> 4) + e);
        bytes.insert(bytes.end(), (const uint8_t *)f, (const uint8_t *)f + sizeof(f));
    }
    // out_pad: rows padded to whole units of 32 (RH = 2) or to 16 (RH = 1, a single unit); in_pad: input features padded to 32
    void add(const Mat &w, uint32_t out_pad, uint32_t in_pad, int kind, uint32_t RH = 2) {
        const uint32_t OT = RH == 2 ? out_pad / 32 : 1;
        const uint32_t KG = kind == TAPE_F32 ? in_pad / 16 : in_pad / 32;
        for (uint32_t o = 0; o < OT; ++o)
            for (uint32_t g = 0; g < KG; ++g)
                for (uint32_t j = 0; j < RH; ++j) {
                    const uint32_t R = RH == 2 ? 2 * o + j : o;
                    if (kind == TAPE_F32) {
                        frag_f32(w, R, g);
                    } else {
                        frag_bf16(w, R, g, false);
                        if (kind == TAPE_X3) frag_bf16(w, R, g, true);
                    }
                }
    }
    void pad_to_chunk() { bytes.resize((bytes.size() + NLR_TAPE_CHUNK - 1) / NLR_TAPE_CHUNK * NLR_TAPE_CHUNK, 0); }
};

static int upload_bias(NlrModel *m, const float *b, uint32_t n, uint32_t pad, float **out) {
    std::vector<float> v(pad, 0.0f);
    for (uint32_t i = 0; i < n; ++i) v[i] = b[i];
    return dev_upload(m, v.data(), pad * sizeof(float), (void **)out);
}

static int check_linear(const NlrLinear &l, uint32_t out_f, uint32_t in_f, const char *name) {
    NLR_CHECK_ARG(l.weight && l.bias, "%s: weight/bias is NULL", name);
    NLR_CHECK_ARG(l.out_features == out_f && l.in_features == in_f, "%s: expected [%u,%u], got [%u,%u]", name, out_f, in_f,
                  l.out_features, l.in_features);
    return NLR_OK;
}

static int build_level(NlrModel *m, LevelModel &lv, const NlrMlpDesc &d, uint32_t S, uint32_t prec) {
    int rc = nlr_fill_grid_params(&lv.gp, d.grid.table, d.grid.table_dtype, d.grid.offsets, d.grid.num_levels,
                                  d.grid.level_dim, d.grid.log2_per_level_scale, d.grid.base_resolution, d.grid.gridtype,
                                  (int)d.grid.align_corners, d.grid.interp);
    if (rc) return rc;
    NLR_CHECK_ARG(d.grid.table != nullptr, "mlp: hash table pointer is NULL");
    lv.S = S;
    lv.F = d.grid.num_levels * d.grid.level_dim;
    lv.re_weights = d.re_weights ? 1 : 0;
    lv.density_bias = d.density_bias;
    lv.is_prop = d.disable_rgb != 0;
    // sample positions u (stepfun.py:203-216)
    {
        std::vector<float> u(S);
        float mj = 0.0f;
        nlr_sample_u(S, 0, u.data(), nullptr);
        if ((rc = dev_upload(m, u.data(), S * sizeof(float), (void **)&lv.u_det))) return rc;
        nlr_sample_u(S, 1, u.data(), &mj);
        lv.max_jitter = mj;
        if ((rc = dev_upload(m, u.data(), S * sizeof(float), (void **)&lv.u_rand))) return rc;
    }
    if ((rc = check_linear(d.density0, 64, lv.F, "density_layer.0"))) return rc;
    if (lv.is_prop) {
        if ((rc = check_linear(d.density2, 1, 64, "density_layer.2 (PropMLP)"))) return rc;
        if ((rc = dev_upload(m, d.density0.weight, (size_t)64 * lv.F * 4, (void **)&lv.p_w1))) return rc;
        if ((rc = dev_upload(m, d.density0.bias, 64 * 4, (void **)&lv.p_b1))) return rc;
        if ((rc = dev_upload(m, d.density2.weight, 64 * 4, (void **)&lv.p_w2))) return rc;
        lv.p_b2 = d.density2.bias[0];
        return NLR_OK;
    }
    // ---- NerfMLP
    lv.W = d.net_width_viewdirs;
    lv.WB = d.bottleneck_width;
    lv.D = d.net_depth_viewdirs;
    lv.deg = d.deg_view;
    lv.E = 3 + 6 * d.deg_view;
    lv.rgb_premul = d.rgb_premultiplier;
    lv.rgb_bias = d.rgb_bias;
    lv.rgb_padding = d.rgb_padding;
    if (lv.D < 2 || d.skip_layer_dir != 0 || lv.D > NLR_MAX_VIEW_DEPTH)
        NLR_FAIL(NLR_ERR_UNSUPPORTED, "view MLP: fused path needs net_depth_viewdirs in [2,%d] and skip_layer_dir = 0 (got %u, %u)",
                 NLR_MAX_VIEW_DEPTH, lv.D, d.skip_layer_dir);
    if (lv.W % 32 || lv.WB % 32) NLR_FAIL(NLR_ERR_UNSUPPORTED, "view MLP: widths must be multiples of 32");
    if (lv.F % 4) NLR_FAIL(NLR_ERR_UNSUPPORTED, "NerfMLP: L*C = %u must be a multiple of 4", lv.F);
    if ((rc = check_linear(d.density2, lv.WB, 64, "density_layer.2"))) return rc;
    // use_semantic with no_sem_layer (models.py:1133): the logits are channels [1, 1+K) of the bottleneck.  They reach the
    // softmax through the same two head GEMMs as a learned sem_layer, with pass-through weights that are exact in the
    // split-bf16 / f32 arithmetic: hidden rows c and 32 + c hold relu(+x_{1+c}) and relu(-x_{1+c}), output row c their difference.
    const bool sem_layer = d.use_semantic && !d.no_sem_layer;
    const bool sem_pass = d.use_semantic && d.no_sem_layer;
    if (sem_pass) NLR_CHECK_ARG(d.class_num <= 32 && 1 + d.class_num <= d.bottleneck_width, "no_sem_layer: class_num %u does not fit", d.class_num);
    lv.K = d.use_semantic ? d.class_num : 0;
    lv.use_int = d.use_intensity != 0;
    lv.HT = (d.use_semantic ? 2 : 0) + (lv.use_int ? 2 : 0);
    NLR_CHECK_ARG(lv.K + (lv.use_int ? 1 : 0) <= 32, "class_num %u (+intensity) exceeds one 32-row output tile", lv.K);

    lv.prec = prec;
    const int crit = (prec == NLR_PREC_FAST) ? TAPE_X3 : TAPE_F32;   // density trunk + heads
    const int view = (prec == NLR_PREC_F32) ? TAPE_F32 : TAPE_BF16;  // view MLP
    // The tile's program (nlr_mlp_kernel.h): bf16 view MLP: [T V0 V1 | T V0 V1 | hidden.. | RGB] (the trunk + heads T and view
    // layers 0/1 run once per 32-sample half); exact-f32 chain: [T V0 V1 | hidden.. | RGB], consumed once per half.
    std::vector<float> bias;  // b_d0 | b_d2 | b_h1 | b_h2 | view0 | view1 | view2.. | rgb, each padded to whole units of 32
    auto push_bias = [&](const float *b, uint32_t n, uint32_t pad) {
        for (uint32_t i = 0; i < pad; ++i) bias.push_back(i < n ? b[i] : 0.0f);
    };
    // the kernel instances take two 32-feature k-blocks of grid features: up to 64, fewer are zero-padded (16 levels x 2 = 32 features,
    // the GridEncoder's own default, Z/gridencoder/grid.py:96-110, included)
    const uint32_t Fp32 = lv.F <= 64 ? 64 : ((lv.F + 31) / 32) * 32;
    TapeBuilder trunk;
    // density trunk
    trunk.add(mat_from(d.density0, 0, lv.F), 64, Fp32, crit);
    push_bias(d.density0.bias, 64, 64);
    trunk.add(mat_from(d.density2, 0, 64), lv.WB, 64, crit);
    push_bias(d.density2.bias, lv.WB, lv.WB);
    // heads: [sem0 ; int0] stacked, then a block-diagonal [sem2 | 0 ; 0 | int2] into one 32-row unit
    if (lv.HT) {
        const uint32_t HH = lv.HT * 32;
        Mat h1(HH, lv.WB), h2(32, HH);
        std::vector<float> b1(HH, 0.0f), b2(32, 0.0f);
        uint32_t r0 = 0;
        if (sem_layer) {
            if ((rc = check_linear(d.sem0, 64, lv.WB, "sem_layer.0"))) return rc;
            if ((rc = check_linear(d.sem2, d.class_num, 64, "sem_layer.2"))) return rc;
            for (uint32_t r = 0; r < 64; ++r) {
                for (uint32_t c = 0; c < lv.WB; ++c) h1.at(r0 + r, c) = d.sem0.weight[(size_t)r * lv.WB + c];
                b1[r0 + r] = d.sem0.bias[r];
            }
            for (uint32_t r = 0; r < d.class_num; ++r) {
                for (uint32_t c = 0; c < 64; ++c) h2.at(r, r0 + c) = d.sem2.weight[(size_t)r * 64 + c];
                b2[r] = d.sem2.bias[r];
            }
            r0 += 64;
        }
        if (sem_pass) {
            for (uint32_t c = 0; c < d.class_num; ++c) {
                h1.at(r0 + c, 1 + c) = 1.0f;
                h1.at(r0 + 32 + c, 1 + c) = -1.0f;
                h2.at(c, r0 + c) = 1.0f;
                h2.at(c, r0 + 32 + c) = -1.0f;
            }
            r0 += 64;
        }
        if (lv.use_int) {
            if ((rc = check_linear(d.int0, 64, lv.WB, "intensity_layer.0"))) return rc;
            if ((rc = check_linear(d.int2, 1, 64, "intensity_layer.2"))) return rc;
            lv.int_row = lv.K;
            for (uint32_t r = 0; r < 64; ++r) {
                for (uint32_t c = 0; c < lv.WB; ++c) h1.at(r0 + r, c) = d.int0.weight[(size_t)r * lv.WB + c];
                b1[r0 + r] = d.int0.bias[r];
            }
            for (uint32_t c = 0; c < 64; ++c) h2.at(lv.int_row, r0 + c) = d.int2.weight[c];
            b2[lv.int_row] = d.int2.bias[0];
        }
        trunk.add(h1, HH, lv.WB, crit);
        push_bias(b1.data(), HH, HH);
        trunk.add(h2, 32, HH, crit);
        push_bias(b2.data(), 32, 32);
    } else {
        push_bias(nullptr, 0, 32);  // keeps the block layout of the kernel (OB_H2 slot)
    }
    // view MLP.  Input of layer 0 = [bottleneck (WB) | dir_enc (E)]; of layer 1 = [x (W) | bottleneck | dir_enc]
    // (models.py:1223-1228).  The E dir-encoding columns are zero-padded to one 32-feature k-block.
    const uint32_t in0 = lv.WB + lv.E, in1 = lv.W + in0;
    if ((rc = check_linear(d.view[0], lv.W, in0, "lin_second_stage_0"))) return rc;
    if ((rc = check_linear(d.view[1], lv.W, in1, "lin_second_stage_1"))) return rc;
    NLR_CHECK_ARG(lv.E <= 32, "deg_view %u gives %u > 32 direction features -- no fused path", lv.deg, lv.E);
    TapeBuilder v01;
    v01.add(mat_from(d.view[0], 0, in0), lv.W, lv.WB + 32, view);
    push_bias(d.view[0].bias, lv.W, lv.W);
    v01.add(mat_from(d.view[1], 0, in1), lv.W, lv.W + lv.WB + 32, view);
    push_bias(d.view[1].bias, lv.W, lv.W);
    TapeBuilder tb;
    auto append = [&](const TapeBuilder &x) { tb.bytes.insert(tb.bytes.end(), x.bytes.begin(), x.bytes.end()); };
    if (prec == NLR_PREC_F32) {
        append(trunk);
        append(v01);
    } else {
        append(trunk);
        append(v01);
        append(trunk);
        append(v01);
    }
    for (uint32_t l = 2; l < lv.D; ++l) {
        char nm[64];
        snprintf(nm, sizeof(nm), "lin_second_stage_%u", l);
        if ((rc = check_linear(d.view[l], lv.W, lv.W, nm))) return rc;
        tb.add(mat_from(d.view[l], 0, lv.W), lv.W, lv.W, view);
        push_bias(d.view[l].bias, lv.W, lv.W);
    }
    if ((rc = check_linear(d.rgb_layer, 3, lv.W, "rgb_layer"))) return rc;
    tb.add(mat_from(d.rgb_layer, 0, lv.W), 16, lv.W, view, 1);
    push_bias(d.rgb_layer.bias, 3, 32);
    tb.pad_to_chunk();
    lv.bias_count = (uint32_t)bias.size();
    NLR_CHECK_ARG(lv.bias_count <= 3072, "bias block of %u floats exceeds the 3072-float LDS reservation (view MLP too deep / wide)", lv.bias_count);
    if ((rc = dev_upload(m, bias.data(), bias.size() * 4, (void **)&lv.bias_all))) return rc;
    lv.tape_chunks = (uint32_t)(tb.bytes.size() / NLR_TAPE_CHUNK);
    tb.bytes.resize(tb.bytes.size() + 3 * NLR_TAPE_CHUNK, 0);  // slack: the kernel prefetches up to 3 chunks past the end
    if ((rc = dev_upload(m, tb.bytes.data(), tb.bytes.size(), &lv.tape))) return rc;
    return NLR_OK;
}

extern "C" int nlr_model_create(const NlrModelDesc *desc, NlrModel **out, void *stream) {
    NLR_CHECK_ARG(desc && out, "model_create: NULL argument");
    NLR_CHECK_ARG(desc->num_levels >= 1 && desc->num_levels <= NLR_MAX_LEVELS, "num_levels %u outside [1,%d]", desc->num_levels,
                  NLR_MAX_LEVELS);
    NLR_CHECK_ARG(desc->mlp_precision <= NLR_PREC_FAST, "unknown mlp_precision %u", desc->mlp_precision);
    NlrModel *m = new NlrModel();
    m->up = (hipStream_t)stream;
    {
        int n = 0;
        if (hipGetDevice(&m->device) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, m->device) != hipSuccess || n <= 0) {
            delete m;
            NLR_FAIL(NLR_ERR_HIP, "model_create: cannot query the current device");
        }
        m->cus = (uint32_t)n;
        // diagnostic: run the persistent MLP grid on fewer workgroups than CUs (power / clock experiments); explicit
        // switch, not an environment variable (nlr_debug_set)
        const int w = nlr_debug_get(NLR_DBG_MLP_WORKGROUPS);
        if (w > 0 && w < n) m->cus = (uint32_t)w;
    }
    m->num_levels = desc->num_levels;
    m->dilation_multiplier = desc->dilation_multiplier;
    m->dilation_bias = desc->dilation_bias;
    m->anneal_slope = desc->anneal_slope;
    m->resample_padding = desc->resample_padding;
    m->power_lambda = desc->power_lambda;
    m->std_scale = desc->std_scale;
    m->bg = desc->bg_intensity;
    m->opaque = desc->opaque_background;
    m->prec = desc->mlp_precision;
    for (uint32_t l = 0; l < desc->num_levels; ++l) {
        int rc = NLR_OK;
        if (!desc->mlps[l]) {
            nlr_set_error("model_create: mlps[%u] is NULL", l);
            rc = NLR_ERR_INVALID;
        } else if (desc->num_samples[l] < 2 || desc->num_samples[l] > 512) {
            nlr_set_error("num_samples must be in [2,512], is %u", desc->num_samples[l]);
            rc = NLR_ERR_INVALID;
        } else {
            rc = build_level(m, m->lv[l], *desc->mlps[l], desc->num_samples[l], desc->mlp_precision);
            const bool last = (l + 1 == desc->num_levels);
            if (!rc && last && m->lv[l].is_prop) {
                nlr_set_error("the last level must be a NerfMLP (disable_rgb = 0)");
                rc = NLR_ERR_INVALID;
            }
            if (!rc && !last && !m->lv[l].is_prop) {
                nlr_set_error("single_mlp / NerfMLP as proposal network has no fused path (level %u)", l);
                rc = NLR_ERR_UNSUPPORTED;
            }
        }
        if (rc) {
            nlr_model_destroy(m);
            return rc;
        }
    }
    if (hipStreamSynchronize(m->up) != hipSuccess) {
        nlr_model_destroy(m);
        NLR_FAIL(NLR_ERR_HIP, "model_create: weight upload failed");
    }
    *out = m;
    return NLR_OK;
}

extern "C" void nlr_model_destroy(NlrModel *m) {
    if (!m) return;
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
}

extern "C" int nlr_model_set_table(NlrModel *m, uint32_t level, const void *table_dev, int table_dtype) {
    NLR_CHECK_ARG(m && level < m->num_levels && table_dev, "model_set_table: bad argument");
    NLR_CHECK_ARG(table_dtype == 0 || table_dtype == 1, "model_set_table: table_dtype must be 0 or 1");
    m->lv[level].gp.table = table_dev;
    m->lv[level].gp.table_dtype = table_dtype;
    return NLR_OK;
}

extern "C" int nlr_profile_begin_kinds(NlrModel *m, uint32_t kind_mask) {
    NLR_CHECK_ARG(m != nullptr, "profile_begin: NULL model");
    for (hipEvent_t e : m->prof.ev) (void)hipEventDestroy(e);
    m->prof.ev.clear();
    m->prof.kind.clear();
    m->prof.mask = kind_mask;
    m->prof.armed = true;
    return NLR_OK;
}
extern "C" int nlr_profile_begin(NlrModel *m) { return nlr_profile_begin_kinds(m, 0xffffffffu); }

extern "C" int nlr_profile_end(NlrModel *m, void *stream, float *total_ms, uint32_t *launches) {
    NLR_CHECK_ARG(m && total_ms && launches, "profile_end: NULL argument");
    m->prof.armed = false;
    NLR_HIP(hipStreamSynchronize((hipStream_t)stream));
    for (int k = 0; k < NLR_K_COUNT; ++k) {
        total_ms[k] = 0.0f;
        launches[k] = 0;
    }
    for (size_t i = 0; i < m->prof.kind.size(); ++i) {
        float ms = 0.0f;
        NLR_HIP(hipEventElapsedTime(&ms, m->prof.ev[2 * i], m->prof.ev[2 * i + 1]));
        total_ms[m->prof.kind[i]] += ms;
        launches[m->prof.kind[i]] += 1;
    }
    for (hipEvent_t e : m->prof.ev) (void)hipEventDestroy(e);
    m->prof.ev.clear();
    m->prof.kind.clear();
    return NLR_OK;
}

// ---- workspace ------------------------------------------------------------------------------------
static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

struct Carve {
    char *base;
    size_t off = 0, cap;
    Carve(void *p, size_t c) : base((char *)p), cap(c) {}
    float *take(size_t nfloat) {
        size_t b = al(nfloat * sizeof(float));
        float *r = base ? (float *)(base + off) : nullptr;
        off += b;
        return r;
    }
};

static size_t level_ws(const LevelModel &lv, uint32_t N, bool last) {
    size_t S = lv.S, b = 0;
    b += 2 * al((size_t)N * (S + 1) * 4);  // sdist, tdist
    b += 2 * al((size_t)N * S * 4);        // weights, density
    if (!lv.is_prop) {
        b += al((size_t)N * S * lv.F * 4 + 1024);        // features (+ slack: the MLP kernel stages 64-sample windows)
        b += al((size_t)N * S * 3 * 4);                  // rgb
        b += al((size_t)N * S * (lv.K ? lv.K : 1) * 4);  // semantic
        b += al((size_t)N * S * 4);                      // intensity
        b += al((size_t)N * 32 * 4);                     // per-ray direction encoding
        b += al((size_t)N * 4);                          // |directions| (compositing mode)
        b += al((size_t)N * S * 4);                      // segment records (compositing mode: N * S / 32 records of 32 floats)
    }
    (void)last;
    return b;
}

extern "C" size_t nlr_workspace_bytes(const NlrModel *m, uint32_t N) {
    if (!m) return 0;
    size_t b = 256 + al(64 * sizeof(uint32_t));  // (+ the per-level vote counters of render_impl)
    for (uint32_t l = 0; l < m->num_levels; ++l) b += level_ws(m->lv[l], N, l + 1 == m->num_levels);
    return b;
}

// Run encode + MLP of one NerfMLP level (or the fused proposal kernel) for given tdist.
// seg != NULL: compositing mode (rgb / sem / inten are not written; see nlr_mlp_kernel.h) - needs dnorm [N] scratch
static int run_mlp_level(const NlrModel *m, const LevelModel &lv, const NlrRays *rays, const float *tdist, uint32_t N,
                         uint32_t n, uint32_t mloops, const float *rand_deg, float *feat, float *raybias, float *density,
                         float *rgb, float *sem, float *inten, float *prop_feat, hipStream_t st, bool internal_feat = false,
                         float *seg = nullptr, float *dnorm = nullptr, uint32_t ray_groups = 0, const uint32_t *votes = nullptr) {
    // features in the workspace (never seen by the caller) take the piece-major layout when the fast kernels apply
    const int piece_major = (internal_feat && !lv.is_prop && n <= 8 && lv.F % 4 == 0) ? 1 : 0;  // any level_dim in {1, 2, 4, 8}: nlr_feat_ptr
    CastParams cp;
    int rc = nlr_fill_cast_params(&cp, rays, tdist, rand_deg, N, lv.S, n, mloops, m->std_scale);
    if (rc) return rc;
    cp.ray_groups = ray_groups;  // (nlr_kernels.h: which samples share a wave of the fused cast + encode kernels)
    cp.votes = votes;
    cp.vote_min = (uint32_t)(((uint64_t)(N ? N - 1 : 0) * lv.S) / 2);
    if (lv.is_prop) {
        ProfScope ps(&m->prof, NLR_K_PROP, st);
        return nlr_launch_prop(cp, lv.gp, lv.p_w1, lv.p_b1, lv.p_w2, lv.p_b2, lv.density_bias, lv.re_weights, density, prop_feat, st);
    }
    {
        ProfScope ps(&m->prof, NLR_K_ENCODE, st);
        if ((rc = nlr_launch_encode(cp, lv.gp, lv.re_weights, feat, piece_major, st))) return rc;
    }
    if (rgb || seg) {
        NLR_CHECK_ARG(rays->viewdirs != nullptr, "NerfMLP: viewdirs is NULL");
        DirEncParams dp;
        dp.viewdirs = rays->viewdirs;
        dp.dirs = seg ? rays->directions : nullptr;
        dp.dnorm = seg ? dnorm : nullptr;
        dp.N = N;
        dp.deg = lv.deg;
        dp.E = lv.E;
        dp.out = raybias;
        ProfScope ps(&m->prof, NLR_K_DIRBIAS, st);
        if ((rc = nlr_launch_direnc(dp, st))) return rc;
    }
    MlpParams P;
    memset(&P, 0, sizeof(P));
    P.feat = feat;
    P.feat_piece_major = piece_major;
    P.M = N * lv.S;
    P.S = lv.S;
    P.F = lv.F;
    P.tape = (const uint4 *)lv.tape;
    P.tape_chunks = lv.tape_chunks;
    P.bias_all = lv.bias_all;
    P.bias_count = lv.bias_count;
    P.enc = raybias;
    P.depth = lv.D;
    P.K = (sem || seg) ? lv.K : 0;
    P.int_row = lv.int_row;
    P.tdist = tdist;
    P.dnorm = dnorm;
    P.seg = seg;
    P.opaque = m->opaque;
    P.density_bias = lv.density_bias;
    P.rgb_premul = lv.rgb_premul;
    P.rgb_bias = lv.rgb_bias;
    P.rgb_padding = lv.rgb_padding;
    P.density = density;
    P.rgb = seg ? seg : rgb;  // (compositing mode: non-NULL = run the view MLP; nothing is stored through it)
    P.sem = sem;
    P.inten = (lv.use_int && (inten || seg)) ? (seg ? seg : inten) : nullptr;  // (compositing mode: non-NULL = the records carry the intensity)
    ProfScope ps(&m->prof, NLR_K_MLP, st);
    return nlr_launch_mlp(P, lv.W, lv.WB, lv.HT, lv.prec, m->cus, st);
}

extern "C" int nlr_mlp_level(const NlrModel *m, uint32_t level, const NlrRays *rays, const float *tdist, uint32_t N,
                             uint32_t sample_n, uint32_t sample_m, const float *rand_deg, float *features, float *density,
                             float *rgb, float *semantic, float *intensity, void *workspace, size_t workspace_bytes,
                             void *stream) {
    NLR_CHECK_ARG(m && level < m->num_levels && rays && tdist && density, "mlp_level: bad argument");
    if (N == 0) return NLR_OK;
    const LevelModel &lv = m->lv[level];
    hipStream_t st = (hipStream_t)stream;
    if (lv.is_prop)
        return run_mlp_level(m, lv, rays, tdist, N, sample_n, sample_m, rand_deg, nullptr, nullptr, density, nullptr, nullptr,
                             nullptr, features, st);
    Carve c(workspace, workspace_bytes);
    float *feat = features ? features : c.take((size_t)N * lv.S * lv.F + 256);
    float *rb = c.take((size_t)N * 32);
    float *sem = semantic ? semantic : (lv.K ? c.take((size_t)N * lv.S * lv.K) : nullptr);
    if (c.off > workspace_bytes || (!workspace && c.off))
        NLR_FAIL(NLR_ERR_WORKSPACE, "mlp_level: workspace %zu B < needed %zu B", workspace_bytes, c.off);
    return run_mlp_level(m, lv, rays, tdist, N, sample_n, sample_m, rand_deg, feat, rb, density, rgb, sem, intensity, nullptr, st,
                         features == nullptr);
}

// ---- Model.forward ----------------------------------------------------------------------------------
// The dynamic-object branch of one call (ZI/models.py:401-477): between the MLP and the compositing of every level the samples
// inside a track's box take that track's ObjMLP results.
struct DynScene {
    const NlrObjects *objs;
    const float *box_params;
    uint32_t n_obj;
    int32_t *const *winner;  // optional per-level owner maps
};

static int render_impl(const NlrModel *m, const NlrRays *rays, uint32_t N, const NlrRenderCfg *cfg, const NlrOut *out, void *workspace,
                       size_t workspace_bytes, void *stream, const DynScene *dyn) {
    NLR_CHECK_ARG(m && rays && cfg && out, "render_rays: NULL argument");
    NLR_CHECK_ARG(rays->origins && rays->directions && rays->near && rays->far && rays->radii, "render_rays: ray batch has NULL tensors");
    if (N == 0) return NLR_OK;
    size_t need = nlr_workspace_bytes(m, N), obj_ws = 0;
    if (dyn) {
        uint32_t smax = 0;
        for (uint32_t l = 0; l < m->num_levels; ++l) smax = m->lv[l].S > smax ? m->lv[l].S : smax;
        obj_ws = nlr_objects_workspace_bytes(dyn->objs, N, smax);
        need += obj_ws;
    }
    if (!workspace || workspace_bytes < need)
        NLR_FAIL(NLR_ERR_WORKSPACE, "render_rays: workspace %zu B < %zu B (nlr_workspace_bytes%s)", workspace_bytes, need,
                 dyn ? " + nlr_objects_workspace_bytes" : "");
    hipStream_t st = (hipStream_t)stream;
    void *obj_space = dyn ? (char *)workspace + (workspace_bytes - obj_ws) : nullptr;  // the tail of the workspace
    Carve c(workspace, workspace_bytes - obj_ws);
    const uint32_t n = cfg->sample_n ? cfg->sample_n : 7, mloops = cfg->sample_m ? cfg->sample_m : 3;
    uint32_t *votes = (uint32_t *)c.take(64);  // per level: coherent (adjacent rays, sample) pairs, see nlr_slot_sample
    if (int rc0 = nlr_launch_ray_vote(nullptr, 0, 0, votes, st)) return rc0;  // zero the counters (a kernel: a memset node slows a captured graph down)

    const float *prev_s = nullptr, *prev_w = nullptr;
    uint32_t n_prev = 0;
    double prod = 1.0;
    for (uint32_t l = 0; l < m->num_levels; ++l) {
        const LevelModel &lv = m->lv[l];
        const bool last = (l + 1 == m->num_levels);
        const uint32_t S = lv.S;
        const NlrLevelOut &ho = out->history[l];
        float *sdist = ho.sdist ? ho.sdist : c.take((size_t)N * (S + 1));
        float *tdist = ho.tdist ? ho.tdist : c.take((size_t)N * (S + 1));
        float *weights = ho.weights ? ho.weights : c.take((size_t)N * S);
        float *density = ho.density ? ho.density : c.take((size_t)N * S);
        // models.py:322-346
        const bool use_dil = m->dilation_bias > 0 || m->dilation_multiplier > 0;
        const float dilation = (l > 0 && use_dil) ? (float)(m->dilation_bias + m->dilation_multiplier * 1.0 / prod) : 0.0f;
        prod *= (double)S;
        const float tf = cfg->train_frac;
        const float anneal = m->anneal_slope > 0 ? (float)(((double)m->anneal_slope * tf) / (((double)m->anneal_slope - 1.0) * tf + 1.0)) : 1.0f;
        const float *jit = cfg->rand_jitter[l];
        int rc;
        {
            ProfScope ps(&m->prof, NLR_K_RESAMPLE, st);
            rc = nlr_launch_resample(prev_s, prev_w, n_prev, dilation, anneal, m->resample_padding, S, jit ? lv.u_rand : lv.u_det,
                                     jit, lv.max_jitter, rays->near, rays->far, m->power_lambda, N, sdist, tdist, st);
        }
        if (rc) return rc;
        float *feat = nullptr, *rb = nullptr, *rgb = nullptr, *sem = nullptr, *inten = nullptr, *seg = nullptr, *dnorm = nullptr;
        // Compositing mode: nobody asked for the per-sample heads of the last level (ray_history), so the MLP kernel composites
        // inside its 32-sample segments and 24 floats per sample never go to HBM.
        const bool fuse = !dyn && last && !lv.is_prop && !ho.rgb && !ho.semantic && !ho.intensity && lv.gp.C == 4 && n <= 8 &&
                          nlr_mlp_can_composite(lv.W, lv.WB, lv.HT, lv.prec, lv.F, S, lv.K, lv.use_int, (uint64_t)N * S);
        if (!lv.is_prop) {
            feat = c.take((size_t)N * S * lv.F + 256);
            rb = c.take((size_t)N * 32);
            if (fuse) {
                dnorm = c.take((size_t)N);
                seg = c.take((size_t)N * S);
            } else {
                rgb = ho.rgb ? ho.rgb : c.take((size_t)N * S * 3);
                sem = lv.K ? (ho.semantic ? ho.semantic : c.take((size_t)N * S * lv.K)) : nullptr;
                inten = lv.use_int ? (ho.intensity ? ho.intensity : c.take((size_t)N * S)) : nullptr;
            }
        }
        // which samples share a wave of the encode kernels: consecutive rays of a sweep / an image tile are neighbours in space, and whether
        // their samples are depends on the field - counted here per level unless the caller says the batch is shuffled
        uint32_t groups = cfg->shuffled_rays ? 0u : 2u;
        if (const int force = nlr_debug_get(NLR_DBG_RAY_GROUPS)) groups = force == 1 ? 1u : 0u;
        if (groups == 2u && n > 8) groups = 0u;  // (the sample-parallel kernels of sample_n > 8 have one order)
        if (groups == 2u && (rc = nlr_launch_ray_vote(tdist, N, S, votes + l, st))) return rc;
        rc = run_mlp_level(m, lv, rays, tdist, N, n, mloops, cfg->rand_deg[l], feat, rb, density, rgb, sem, inten, nullptr, st, true, seg,
                           dnorm, groups, votes + l);
        if (rc) return rc;
        if (dyn) {  // (the object networks have no intensity head: DynamicModel refuses use_intensity, like the reference's merge)
            rc = nlr_objects_apply_impl(dyn->objs, rays, tdist, dyn->box_params, N, S, dyn->n_obj, density, rgb, sem, lv.K,
                                        dyn->winner ? dyn->winner[l] : nullptr, obj_space, obj_ws, st);
            if (rc) return rc;
        }
        {
            ProfScope ps(&m->prof, NLR_K_COMPOSITE, st);
            if (fuse)
                rc = nlr_composite_segments(density, tdist, rays->directions, seg, lv.K, lv.use_int ? 1 : 0, rays->far, rays->origins, N, S,
                                            (int)m->opaque, m->bg, (int)cfg->compute_extras, cfg->scale_factor, weights, out, ho.depth, st);
            else if (last)
                rc = nlr_composite_level(density, tdist, rays->directions, rgb, sem, inten, rays->far, rays->origins, N, S, lv.K,
                                         (int)m->opaque, m->bg, (int)cfg->compute_extras, cfg->scale_factor, weights, out, ho.depth, st);
            else {  // a level before the last: weights for the next resampling, its depth, and (on request) the rest of its rendering
                NlrOut lo;
                memset(&lo, 0, sizeof(lo));
                lo.rgb = ho.r_rgb;
                lo.acc = ho.r_acc;
                lo.distance_mean = ho.r_distance_mean;
                lo.distance_median = ho.r_distance_median;
                lo.distance_percentile_5 = ho.r_distance_percentile_5;
                lo.distance_percentile_95 = ho.r_distance_percentile_95;
                const bool extras = cfg->compute_extras && (lo.distance_mean || lo.distance_median || lo.distance_percentile_5 || lo.distance_percentile_95);
                rc = nlr_composite_level(density, tdist, rays->directions, nullptr, nullptr, nullptr, rays->far, rays->origins, N, S, 0,
                                         (int)m->opaque, m->bg, extras ? 1 : 0, 0.0f, weights, &lo, ho.depth, st);
            }
        }
        if (rc) return rc;
        prev_s = sdist;
        prev_w = weights;
        n_prev = S;
    }
    if (c.off > workspace_bytes - obj_ws) NLR_FAIL(NLR_ERR_WORKSPACE, "render_rays: carved %zu B > workspace %zu B", c.off, workspace_bytes - obj_ws);
    return NLR_OK;
}

extern "C" int nlr_render_rays(const NlrModel *m, const NlrRays *rays, uint32_t N, const NlrRenderCfg *cfg, const NlrOut *out,
                               void *workspace, size_t workspace_bytes, void *stream) {
    return render_impl(m, rays, N, cfg, out, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int nlr_render_rays_dynamic(const NlrModel *m, const NlrObjects *o, const NlrRays *rays, const float *box_params, uint32_t n_obj,
                                       uint32_t N, const NlrRenderCfg *cfg, const NlrOut *out, int32_t *const *winner, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    NLR_CHECK_ARG(m && o && (box_params || n_obj == 0), "render_rays_dynamic: NULL argument");
    for (uint32_t l = 0; l < m->num_levels; ++l)
        NLR_CHECK_ARG(!m->lv[l].use_int, "render_rays_dynamic: the object networks have no intensity head (ZI/models.py:469 assigns None)");
    DynScene d{o, box_params, n_obj, winner};
    return render_impl(m, rays, N, cfg, out, workspace, workspace_bytes, stream, &d);
}
