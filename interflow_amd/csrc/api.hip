// api.hip -- the C ABI of libinterflow_hip.so (see include/interflow_hip.h) and the host-side drivers that walk
// FlowUpsamplerNet.layers natively (no Python dispatch per layer, no allocation, no synchronisation).
#include "ifl_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

void ifl_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ifl_last_error(void) { return g_err; }

// ---- opt-in per-kernel timing with HIP events on the caller's stream (used by bench.py for the roofline figure) ----
#include <vector>
namespace {
struct ProfRec { int id; hipEvent_t a, b; };
unsigned g_prof_mask = 0;      // bit i set: time kernel id i
std::vector<ProfRec> g_prof;
struct ProfScope {
    hipStream_t st; hipEvent_t b = nullptr;
    ProfScope(int id, hipStream_t s) : st(s) {
        if (!((g_prof_mask >> id) & 1u)) return;
        ProfRec r{id, nullptr, nullptr};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        (void)hipEventRecord(r.a, st);
        b = r.b;
        g_prof.push_back(r);
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, st); }
};
}  // namespace

extern "C" int ifl_prof_enable(int mask)
{
    if (!mask) {
        for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        g_prof.clear();
    }
    g_prof_mask = (unsigned)mask;
    return 0;
}

extern "C" int ifl_prof_read(int kernel_id, double* total_ms, int64_t* count)
{
    double tot = 0; int64_t n = 0;
    for (auto& r : g_prof) {
        if (r.id != kernel_id) continue;
        float ms = 0;
        IFL_HIP(hipEventSynchronize(r.b));
        IFL_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        tot += ms; ++n;
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    return 0;
}
extern "C" int ifl_version(void) { return IFL_VERSION; }

extern "C" int64_t ifl_step_param_floats(int C, int has_coupling) { return ifl_step_off(C, has_coupling).total; }

extern "C" int64_t ifl_step_field_off(int C, int field)
{
    IflStepOff o = ifl_step_off(C, 1);
    switch (field) {
        case IFL_F_ANIC_FWD: return o.anic_fwd;
        case IFL_F_ANIC_INV: return o.anic_inv;
        case IFL_F_W1A: return o.w1[0];
        case IFL_F_W1F: return o.w1[1];
        case IFL_F_AN1A: return o.an1[0];
        case IFL_F_AN1F: return o.an1[1];
        case IFL_F_W2A: return o.w2[0];
        case IFL_F_W2F: return o.w2[1];
        case IFL_F_AN2A: return o.an2[0];
        case IFL_F_AN2F: return o.an2[1];
        case IFL_F_W3A: return o.w3[0];
        case IFL_F_W3F: return o.w3[1];
        case IFL_F_B3A: return o.b3[0];
        case IFL_F_B3F: return o.b3[1];
    }
    return -1;
}

// ---- tiling ---------------------------------------------------------------------------------------------------
// Pick the band width that minimises (MFMA lane slots, then staged patch pixels) subject to the LDS patch bound.
static IflGeom choose_geom(int H, int W)
{
    IflGeom best{};
    double best_cost = -1;
    for (int bw = 1; bw <= W && bw <= 256; ++bw) {
        IflGeom g{H, W, bw, ifl_cdiv(W, bw), ifl_cdiv(H * bw, IFL_MT), 0};
        g.ntiles = g.nbands * g.tiles_per_band;
        bool ok = true;
        double patch = 0;
        int active = 0;
        for (int t = 0; t < g.ntiles && ok; ++t) {
            IflTile tl = ifl_tile(g, t);
            if (!tl.active) continue;
            ++active;
            if (tl.PP > IFL_PP_MAX) ok = false;
            patch += tl.PP;
        }
        if (!ok) continue;
        double cost = (double)active * IFL_MT * 8.0 + patch;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
}

// ---- workspace ------------------------------------------------------------------------------------------------
int ifl_ws_layout(void* base, int B, int H, int W, IflWs* out)
{
    IflWs w{};
    w.B = B; w.H = H; w.W = W;
    int64_t p = 0;
    auto take = [&](int64_t nfloats) { int64_t r = p; p += (nfloats * 4 + 255) / 256 * 256; return (float*)((char*)base + r); };
    int slots = 0;
    for (int l = 0; l < IFL_MAX_LEVELS; ++l) {
        IflLevelWs& lv = w.lv[l];
        lv.H = H >> (l + 1); lv.W = W >> (l + 1);
        lv.C = 3 << (2 * (l + 1));
        int64_t ring = (int64_t)B * (lv.H + 2) * (lv.W + 2);
        lv.ft = take(ring * IFL_FT);
        lv.z = take(ring * lv.C);
        lv.h2 = take(ring * IFL_H2);
        lv.geom = choose_geom(lv.H, lv.W);
        slots += 2 * 16 * lv.geom.ntiles;
    }
    w.ld_slots = ifl_roundup(slots, 64);
    w.ldpart = take((int64_t)B * w.ld_slots);
    w.bytes = p;
    *out = w;
    return 0;
}

static int check_shape(int B, int H, int W, int div = 4)
{
    IFL_CHECK_ARG(B >= 1 && H >= div && W >= div, IFL_EINVAL, "bad shape B=%d H=%d W=%d", B, H, W);
    IFL_CHECK_ARG(H % div == 0 && W % div == 0, IFL_EINVAL, "H=%d, W=%d must be multiples of %d (flow.py:172)", H, W, div);
    IFL_CHECK_ARG((int64_t)(H / 2 + 2) * (W / 2 + 2) * IFL_FT < (int64_t)1 << 31, IFL_ENOTSUP,
                  "image plane too large for 32-bit tile offsets: tile the image");
    IFL_CHECK_ARG(B <= 65535, IFL_ENOTSUP, "B=%d exceeds the grid's y extent", B);
    return 0;
}

extern "C" int64_t ifl_ws_bytes(int B, int H, int W)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;             // negative IFL_E* code
    IflWs w;
    ifl_ws_layout(nullptr, B, H, W, &w);
    return w.bytes;
}

extern "C" int ifl_ws_init(void* ws, int64_t ws_bytes, int B, int H, int W, void* stream)
{
    int rc = check_shape(B, H, W, 2);     // ifl_flow_step alone accepts planes with H % 4 == 2
    if (rc) return rc;
    IFL_CHECK_ARG(ws != nullptr, IFL_EINVAL, "null workspace");
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    IFL_CHECK_ARG(ws_bytes >= w.bytes, IFL_ESMALLWS, "workspace %lld B < required %lld B", (long long)ws_bytes, (long long)w.bytes);
    IFL_HIP(hipMemsetAsync(ws, 0, (size_t)ws_bytes, (hipStream_t)stream));
    return 0;
}

extern "C" int ifl_load_cond(void* ws, int B, int H, int W, int level, const float* ft_nchw, void* stream)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    IFL_CHECK_ARG(ws && ft_nchw, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(level >= 1 && level <= IFL_MAX_LEVELS, IFL_EINVAL, "level %d outside 1..%d", level, IFL_MAX_LEVELS);
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    const IflLevelWs& lv = w.lv[level - 1];
    return ifl_k_nchw_to_ring(ft_nchw, lv.ft, B, IFL_FT, lv.H, lv.W, (hipStream_t)stream);
}

// ---- one FlowStep on a level's ringed buffers -------------------------------------------------------------------
static int step_encode(const ifl_layer& L, const float* params, const IflWs& w, const IflLevelWs& lv, int& slots,
                       double& constant, float eps, hipStream_t st)
{
    IflStepOff off = ifl_step_off(L.C, L.kind == 2);
    const float* prm = params + L.param_off;
    int rc;
    { ProfScope ps(IFL_K_ANIC, st); rc = ifl_k_anic(lv.z, prm + off.anic_fwd, w.B, L.C, lv.H, lv.W, 0, st); }
    if (rc) return rc;
    double px = (double)lv.H * lv.W;
    constant += (double)L.sum_logs * px + (double)L.sum_log_s * px;
    if (L.kind == 2) {
        { ProfScope ps(lv.C == 12 ? IFL_K_CONV1_ENC_L1 : IFL_K_CONV1_ENC_L2, st);
          rc = ifl_k_conv1(lv.ft, lv.z, lv.h2, prm, off, lv.geom, w.B, L.C, 2, st); }
        if (rc) return rc;
        { ProfScope ps(IFL_K_CONV3_ENC, st);
          rc = ifl_k_conv3(lv.h2, lv.z, w.ldpart, w.ld_slots, slots, prm, off, lv.geom, w.B, L.C, 0, eps, st); }
        if (rc) return rc;
        slots += lv.geom.ntiles;
    }
    return 0;
}

static int step_decode(const ifl_layer& L, const float* params, const IflWs& w, const IflLevelWs& lv, int& slots,
                       double& constant, float eps, hipStream_t st)
{
    IflStepOff off = ifl_step_off(L.C, L.kind == 2);
    const float* prm = params + L.param_off;
    int rc;
    if (L.kind == 2) {
        const int kid = lv.C == 12 ? IFL_K_CONV1_DEC_L1 : IFL_K_CONV1_DEC_L2;
        { ProfScope ps(kid, st); rc = ifl_k_conv1(lv.ft, lv.z, lv.h2, prm, off, lv.geom, w.B, L.C, 1, st); }
        if (rc) return rc;
        { ProfScope ps(IFL_K_CONV3_DEC, st);
          rc = ifl_k_conv3(lv.h2, lv.z, w.ldpart, w.ld_slots, slots, prm, off, lv.geom, w.B, L.C, 1, eps, st); }
        if (rc) return rc;
        slots += lv.geom.ntiles;
        { ProfScope ps(kid, st); rc = ifl_k_conv1(lv.ft, lv.z, lv.h2, prm, off, lv.geom, w.B, L.C, 0, st); }
        if (rc) return rc;
        { ProfScope ps(IFL_K_CONV3_DEC, st);
          rc = ifl_k_conv3(lv.h2, lv.z, w.ldpart, w.ld_slots, slots, prm, off, lv.geom, w.B, L.C, 2, eps, st); }
        if (rc) return rc;
        slots += lv.geom.ntiles;
    }
    { ProfScope ps(IFL_K_ANIC, st); rc = ifl_k_anic(lv.z, prm + off.anic_inv, w.B, L.C, lv.H, lv.W, 1, st); }
    if (rc) return rc;
    double px = (double)lv.H * lv.W;
    constant -= (double)L.sum_logs * px + (double)L.sum_log_s * px;
    return 0;
}

static int check_layers(const ifl_layer* layers, int n_layers)
{
    IFL_CHECK_ARG(layers && n_layers > 0, IFL_EINVAL, "no layers");
    IFL_CHECK_ARG(layers[0].kind == 0, IFL_EINVAL, "the first layer must be a squeeze (FlowUpsamplerNet.py:94-96)");
    int lvl = 0, C = 3, cond = 0;
    for (int i = 0; i < n_layers; ++i) {
        const ifl_layer& L = layers[i];
        IFL_CHECK_ARG(L.kind >= 0 && L.kind <= 2, IFL_EINVAL, "layer %d: kind %d", i, L.kind);
        if (L.kind == 0) { ++lvl; C *= 4; cond = 0; }
        IFL_CHECK_ARG(lvl <= IFL_MAX_LEVELS, IFL_ENOTSUP, "more than %d squeeze levels", IFL_MAX_LEVELS);
        IFL_CHECK_ARG(L.C == C, IFL_EINVAL, "layer %d: C=%d, expected %d", i, L.C, C);
        if (L.kind == 2) ++cond;
        IFL_CHECK_ARG(cond <= 16, IFL_ENOTSUP, "more than 16 conditional steps per level");
    }
    IFL_CHECK_ARG(lvl == IFL_MAX_LEVELS, IFL_ENOTSUP, "expected %d squeeze levels, got %d", IFL_MAX_LEVELS, lvl);
    return 0;
}

extern "C" int ifl_flow_encode(const ifl_layer* layers, int n_layers, const float* params, void* ws,
                               const float* x, float* z, const float* logdet_in, float* logdet_out,
                               int B, int H, int W, float eps, void* stream)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    if ((rc = check_layers(layers, n_layers))) return rc;
    IFL_CHECK_ARG(params && ws && x && z && logdet_out, IFL_EINVAL, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    int lvl = -1, slots = 0;
    double constant = 0.0;
    for (int i = 0; i < n_layers; ++i) {
        const ifl_layer& L = layers[i];
        if (L.kind == 0) {
            ++lvl;
            if (lvl == 0) rc = ifl_k_squeeze_in(x, w.lv[0].z, B, 3, H, W, st);
            else rc = ifl_k_squeeze_ring(w.lv[lvl - 1].z, w.lv[lvl].z, B, w.lv[lvl - 1].C, w.lv[lvl - 1].H, w.lv[lvl - 1].W, st);
        } else {
            rc = step_encode(L, params, w, w.lv[lvl], slots, constant, eps, st);
        }
        if (rc) return rc;
    }
    const IflLevelWs& last = w.lv[lvl];
    if ((rc = ifl_k_ring_to_nchw(last.z, z, B, last.C, last.H, last.W, st))) return rc;
    return ifl_k_ld_finalize(w.ldpart, slots, w.ld_slots, logdet_in, logdet_out, constant, 1.0, B, st);
}

extern "C" int ifl_flow_decode(const ifl_layer* layers, int n_layers, const float* params, void* ws,
                               const float* z, float* x, const float* logdet_in, float* logdet_out,
                               int B, int H, int W, float eps, void* stream)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    if ((rc = check_layers(layers, n_layers))) return rc;
    IFL_CHECK_ARG(params && ws && x && z && logdet_out, IFL_EINVAL, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    int lvl = IFL_MAX_LEVELS - 1, slots = 0;
    double constant = 0.0;
    const IflLevelWs& last = w.lv[lvl];
    if ((rc = ifl_k_nchw_to_ring(z, last.z, B, last.C, last.H, last.W, st))) return rc;
    for (int i = n_layers - 1; i >= 0; --i) {
        const ifl_layer& L = layers[i];
        if (L.kind == 0) {
            if (lvl == 0) rc = ifl_k_unsqueeze_out(w.lv[0].z, x, B, 3, H, W, st);
            else rc = ifl_k_unsqueeze_ring(w.lv[lvl].z, w.lv[lvl - 1].z, B, w.lv[lvl - 1].C, w.lv[lvl - 1].H, w.lv[lvl - 1].W, st);
            --lvl;
        } else {
            rc = step_decode(L, params, w, w.lv[lvl], slots, constant, eps, st);
        }
        if (rc) return rc;
    }
    return ifl_k_ld_finalize(w.ldpart, slots, w.ld_slots, logdet_in, logdet_out, constant, 1.0, B, st);
}

extern "C" int ifl_flow_step(const ifl_layer* layer, const float* params, void* ws, const float* z_in, float* z_out,
                             const float* ft, const float* logdet_in, float* logdet_out, int B, int h, int w_, int reverse,
                             float eps, void* stream)
{
    IFL_CHECK_ARG(layer && params && ws && z_in && z_out && logdet_out, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(layer->kind == 1 || layer->kind == 2, IFL_EINVAL, "layer kind %d is not a FlowStep", layer->kind);
    IFL_CHECK_ARG(layer->C == 12 || layer->C == 48, IFL_ENOTSUP, "C=%d not supported (12 or 48)", layer->C);
    IFL_CHECK_ARG(layer->kind == 1 || ft, IFL_EINVAL, "conditional step needs ft");
    int lvl = layer->C == 12 ? 0 : 1;
    int H = h << (lvl + 1), W = w_ << (lvl + 1);
    IFL_CHECK_ARG(h >= 1 && w_ >= 1, IFL_EINVAL, "bad shape h=%d w=%d", h, w_);
    int rc = check_shape(B, H, W, 2);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    const IflLevelWs& lv = w.lv[lvl];
    if ((rc = ifl_k_nchw_to_ring(z_in, lv.z, B, lv.C, lv.H, lv.W, st))) return rc;
    if (layer->kind == 2 && (rc = ifl_k_nchw_to_ring(ft, lv.ft, B, IFL_FT, lv.H, lv.W, st))) return rc;
    int slots = 0;
    double constant = 0.0;
    rc = reverse ? step_decode(*layer, params, w, lv, slots, constant, eps, st)
                 : step_encode(*layer, params, w, lv, slots, constant, eps, st);
    if (rc) return rc;
    if ((rc = ifl_k_ring_to_nchw(lv.z, z_out, B, lv.C, lv.H, lv.W, st))) return rc;
    return ifl_k_ld_finalize(w.ldpart, slots, w.ld_slots, logdet_in, logdet_out, constant, 1.0, B, st);
}

// ---- stand-alone operators ---------------------------------------------------------------------------------------
extern "C" int ifl_squeeze2d(const float* in, float* out, int B, int C, int H, int W, int factor, void* stream)
{
    IFL_CHECK_ARG(factor >= 1, IFL_EINVAL, "factor %d", factor);
    IFL_CHECK_ARG(H % factor == 0 && W % factor == 0, IFL_EINVAL, "(%d, %d, %d)", H, W, factor);     // flow.py:172
    IFL_CHECK_ARG((in && out) || (int64_t)B * C * H * W == 0, IFL_EINVAL, "null pointer");
    return ifl_k_squeeze_nchw(in, out, B, C, H, W, factor, 0, (hipStream_t)stream);
}

extern "C" int ifl_unsqueeze2d(const float* in, float* out, int B, int C, int H, int W, int factor, void* stream)
{
    // `in` is the squeezed tensor [B, C, H, W]; out is [B, C/f^2, H*f, W*f]                            flow.py:179-193
    IFL_CHECK_ARG(factor >= 1, IFL_EINVAL, "factor %d", factor);
    IFL_CHECK_ARG(C % (factor * factor) == 0, IFL_EINVAL, "%d", C);                                     // flow.py:189
    IFL_CHECK_ARG((in && out) || (int64_t)B * C * H * W == 0, IFL_EINVAL, "null pointer");
    return ifl_k_squeeze_nchw(in, out, B, C / (factor * factor), H * factor, W * factor, factor, 1, (hipStream_t)stream);
}

extern "C" int ifl_gaussian_logp(const float* z, const float* mean, const float* cov_inv, float logdet_cov,
                                 float* scratch, float* out, int B, int C, int H, int W, void* stream)
{
    IFL_CHECK_ARG(z && scratch && out, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(C >= 1 && C <= 64, IFL_ENOTSUP, "C=%d outside 1..64", C);
    IFL_CHECK_ARG(cov_inv == nullptr || mean != nullptr, IFL_EINVAL, "cov_inv without mean");
    IFL_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, IFL_EINVAL, "bad shape");
    return ifl_k_logp(z, mean, cov_inv, logdet_cov, scratch, out, B, C, H * W, (hipStream_t)stream);
}

extern "C" int ifl_lerp(const float* za, const float* zb, const float* wa, const float* wb, float* out, int B,
                        int64_t n_per_sample, void* stream)
{
    IFL_CHECK_ARG(za && zb && wa && wb && out, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 0 && n_per_sample >= 0, IFL_EINVAL, "bad size");
    return ifl_k_lerp(za, zb, wa, wb, out, B, n_per_sample, (hipStream_t)stream);
}
