// api.hip -- the C ABI of libinterflow_hip.so (see include/interflow_hip.h) and the host-side drivers that walk
// FlowUpsamplerNet.layers natively (no Python dispatch per layer, no allocation, no synchronisation).
#include "ifl_common.h"
#include <map>
#include <tuple>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

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
#include <atomic>
#include <mutex>
#include <vector>
namespace {
struct ProfRec { int id; hipEvent_t a, b; };
std::atomic<unsigned> g_prof_mask{0};      // bit i set: time kernel id i
std::mutex g_prof_mu;                      // guards g_prof (drivers may be called from several host threads)
std::vector<ProfRec> g_prof;
struct ProfScope {
    hipStream_t st; hipEvent_t b = nullptr;
    ProfScope(int id, hipStream_t s) : st(s) {
        if (!((g_prof_mask.load(std::memory_order_relaxed) >> id) & 1u)) return;
        ProfRec r{id, nullptr, nullptr};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        (void)hipEventRecord(r.a, st);
        b = r.b;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, st); }
};
}  // namespace

extern "C" int ifl_prof_enable(int mask)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
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
    std::lock_guard<std::mutex> lk(g_prof_mu);
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
        case IFL_F_W1A_BF16: return o.w1b[0];
        case IFL_F_W1F_BF16: return o.w1b[1];
        case IFL_F_W2F_BF16: return o.w2b;
        case IFL_F_W1A_WINO: return o.w1w[0];
        case IFL_F_W1F_WINO: return o.w1w[1];
        case IFL_F_W1A_WINO4: return o.w1x[0];
        case IFL_F_W1F_WINO4: return o.w1x[1];
        case IFL_F_W3A_WINO: return o.w3w[0];
        case IFL_F_W3F_WINO: return o.w3w[1];
    }
    return -1;
}

// ---- tiling ---------------------------------------------------------------------------------------------------
// Pick the band width that minimises (MFMA lane slots, then staged patch pixels) subject to the LDS patch bound.
static IflGeom choose_geom_uncached(int H, int W, int tpx)
{
    IflGeom best{};
    double best_cost = -1;
    for (int bw = 1; bw <= W && bw <= 256; ++bw) {
        IflGeom g{H, W, bw, ifl_cdiv(W, bw), ifl_cdiv(H * bw, tpx), 0, tpx};
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
        double cost = (double)active * tpx * 8.0 + patch;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
}

// Memoised: the per-operator entry points (ifl_conv2d, ifl_conv2d_wgrad) ask for a plane's tiling on every call, and the
// training drivers make about a thousand such calls per step.
IflGeom ifl_choose_geom(int H, int W, int tpx)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, IflGeom> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(H, W, tpx);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (cache.size() > 4096) cache.clear();
    return cache[key] = choose_geom_uncached(H, W, tpx);
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
        lv.vg = take(ifl_vg_floats(B, lv.H, lv.W));
        lv.z = take(ring * lv.C);
        lv.z2 = take(ring * lv.C);
        lv.h2_step = ring * IFL_H2;
        lv.h2 = take(lv.h2_step * IFL_MAX_COND);
        lv.ha = take(ring * IFL_HID);
        lv.sf_step = ring * 2 * lv.C;
        lv.sf = take(lv.sf_step * IFL_MAX_COND);
        lv.geom = ifl_choose_geom(lv.H, lv.W);
        const int w3 = ifl_w3_slots(lv.C, lv.H, lv.W, 1);
        lv.ld_step = lv.geom.ntiles > w3 ? lv.geom.ntiles : w3;
        slots += IFL_MAX_COND * lv.ld_step;
    }
    for (int l = 0; l < IFL_MAX_LEVELS; ++l) {          // fFeatures' sums behind the chain's slots
        w.lv[l].ld_f0 = slots;
        slots += IFL_MAX_COND * w.lv[l].ld_step;
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

// a level's conditioning features in both forms: ringed channels-last (direct, F(2x2) and bf16 kernels, training) and the
// Winograd F(4x4,3x3) input transform of every tile (conv1x.hip), computed from the ringed copy
static int load_ft(const IflLevelWs& lv, int B, const float* ft_nchw, hipStream_t st)
{
    int rc = ifl_k_nchw_to_ring(ft_nchw, lv.ft, B, IFL_FT, lv.H, lv.W, st);
    if (rc) return rc;
    return ifl_k_wino4_input(lv.ft, lv.vg, B, lv.H, lv.W, st);
}

extern "C" int ifl_load_cond(void* ws, int B, int H, int W, int level, const float* ft_nchw, void* stream)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    IFL_CHECK_ARG(ws && ft_nchw, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(level >= 1 && level <= IFL_MAX_LEVELS, IFL_EINVAL, "level %d outside 1..%d", level, IFL_MAX_LEVELS);
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    return load_ft(w.lv[level - 1], B, ft_nchw, (hipStream_t)stream);
}

// algorithm of the coupling networks' last conv (fp32): 1 = Winograd F(2x2,3x3) (conv3w.hip, default), 0 = direct (conv3.hip)
static int env_int0(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int& conv3_algo()
{
    static int a = [] { int v = env_int0("IFL_CONV3_WINO", 1); return v < 0 || v > 1 ? 1 : v; }();
    return a;
}

// fAffine's last conv + the coupling(s) of one step of the chain; advances `slots` by the log-det slots it wrote
static int chain_conv3(const ifl_layer& L, const float* params, const IflWs& w, const IflLevelWs& lv, const float* sf, int& slots,
                       int mode, float eps, int bf16, hipStream_t st)
{
    IflStepOff off = ifl_step_off(L.C, 1);
    int rc;
    if (conv3_algo() == 1 || bf16) {          // the bf16 variant exists in the Winograd form only
        IflStepList one{};
        one.off[0] = L.param_off;
        rc = ifl_k_conv3w(mode == 0 ? 1 : 2, lv.ha, 0, IFL_HID, const_cast<float*>(sf), 0, lv.z, w.ldpart, w.ld_slots, slots, params, one, 1,
                          off, w.B, lv.H, lv.W, L.C, eps, bf16, st);
        slots += ifl_w3_slots(L.C, lv.H, lv.W, 0);
    } else {
        rc = ifl_k_conv3(sf, lv.ha, lv.z, w.ldpart, w.ld_slots, slots, params + L.param_off, off, lv.geom, w.B, L.C, mode, eps, st);
        slots += lv.geom.ntiles;
    }
    return rc;
}

// ---- one FlowStep on a level's ringed buffers -------------------------------------------------------------------
// `slab` is the step's index among the level's conditional steps: its h2 slab holds the conditioning-only part of the first
// conv (conv1_ft) and its sf slab fFeatures' shift / scale (conv3f); both are written by phase 1.
// IFL_FUSE_ANIC (bit mask, default 2): in the encode chain a conditional step's ActNorm + 1x1 conv runs inside its conv1_finish
// kernel (ifl_k_conv1_finish_pre: two launches per step instead of three) for steps with C = 12 (bit 0) / C = 48 (bit 1); 0 = the
// separate kernels.  Measured: at C = 48 the fused kernel takes 62 us in situ against 110 + 44; at C = 12 nothing is gained.
static int& fuse_anic()                      // bit 0: steps with C = 12, bit 1: steps with C = 48
{
    static int v = env_int0("IFL_FUSE_ANIC", 2) & 3;
    return v;
}

// swap_z (may be null): the caller lets this step put the mapped latent into lv.z2; *swap_z = 1 tells it that lv.z and lv.z2
// have changed roles (the running latent is now in what was lv.z2)
static int step_encode(const ifl_layer& L, const float* params, const IflWs& w, const IflLevelWs& lv, int slab,
                       int& slots, double& constant, float eps, int bf16, hipStream_t st, int* swap_z = nullptr)
{
    IflStepOff off = ifl_step_off(L.C, L.kind == 2);
    const float* prm = params + L.param_off;
    int rc;
    double px = (double)lv.H * lv.W;
    constant += (double)L.sum_logs * px + (double)L.sum_log_s * px;
    if (L.kind == 2 && swap_z && !bf16 && (fuse_anic() & (L.C == 12 ? 1 : 2))) {
        const float* h2 = lv.h2 + (int64_t)slab * lv.h2_step;
        const float* sf = lv.sf + (int64_t)slab * lv.sf_step;
        { ProfScope ps(IFL_K_CONV1_FINISH, st);
          rc = ifl_k_conv1_finish_pre(lv.z, lv.z2, prm + off.anic_fwd, h2, lv.ha, prm, off, lv.geom, w.B, L.C, st); }
        if (rc) return rc;
        IflLevelWs lz = lv;
        lz.z = lv.z2; lz.z2 = lv.z;
        { ProfScope ps(IFL_K_CONV3_ENC, st);
          rc = chain_conv3(L, params, w, lz, sf, slots, 0, eps, bf16, st); }
        *swap_z = 1;
        return rc;
    }
    { ProfScope ps(IFL_K_ANIC, st); rc = ifl_k_anic(lv.z, prm + off.anic_fwd, w.B, L.C, lv.H, lv.W, 0, nullptr, st); }
    if (rc) return rc;
    if (L.kind == 2) {
        const float* h2 = lv.h2 + (int64_t)slab * lv.h2_step;
        const float* sf = lv.sf + (int64_t)slab * lv.sf_step;
        { ProfScope ps(IFL_K_CONV1_FINISH, st); rc = ifl_k_conv1_finish(lv.z, h2, lv.ha, prm, off, lv.geom, w.B, L.C, bf16, st); }
        if (rc) return rc;
        { ProfScope ps(IFL_K_CONV3_ENC, st);
          rc = chain_conv3(L, params, w, lv, sf, slots, 0, eps, bf16, st); }
        if (rc) return rc;
    }
    return 0;
}

// fapply_done: the previous kernel of the chain already applied this step's z = (z + shiftFt) * scaleFt; sf_next: apply the
// next step's (the one the chain runs after this one) in this step's last kernel.
// pend_inv (may be null; chain fusion, C = 48): in: the inverse ActNorm + 1x1 block of the step the chain ran before this one,
// whose map is still due -- it is applied, with this step's fapply, inside this step's conv1_finish (the mapped latent goes to
// lv.z2, *swap_z = 1); out (may_defer: the chain's next layer is a conditional step too): this step's own block, left to the
// next step instead of a kernel of its own.
static int step_decode(const ifl_layer& L, const float* params, const IflWs& w, const IflLevelWs& lv, int slab,
                       int& slots, double& constant, float eps, int bf16, hipStream_t st, bool fapply_done = false,
                       const float* sf_next = nullptr, const float** pend_inv = nullptr, bool may_defer = false,
                       int* swap_z = nullptr)
{
    IflStepOff off = ifl_step_off(L.C, L.kind == 2);
    const float* prm = params + L.param_off;
    int rc;
    IflLevelWs lz = lv;
    if (L.kind == 2) {
        const float* h2 = lv.h2 + (int64_t)slab * lv.h2_step;
        const float* sf = lv.sf + (int64_t)slab * lv.sf_step;
        if (pend_inv && *pend_inv) {
            { ProfScope ps(IFL_K_CONV1_FINISH, st);
              rc = ifl_k_conv1_finish_pre(lv.z, lv.z2, *pend_inv, h2, lv.ha, prm, off, lv.geom, w.B, L.C, st, sf); }
            if (rc) return rc;
            lz.z = lv.z2; lz.z2 = lv.z;
            *swap_z = 1;
            *pend_inv = nullptr;
        } else {
            if (!fapply_done) {
                ProfScope ps(IFL_K_CONV3_DEC, st);
                if ((rc = ifl_k_fapply(sf, lv.z, w.B, L.C, lv.H, lv.W, st))) return rc;
            }
            { ProfScope ps(IFL_K_CONV1_FINISH, st); rc = ifl_k_conv1_finish(lv.z, h2, lv.ha, prm, off, lv.geom, w.B, L.C, bf16, st); }
            if (rc) return rc;
        }
        { ProfScope ps(IFL_K_CONV3_DEC, st);
          rc = chain_conv3(L, params, w, lz, sf, slots, 2, eps, bf16, st); }
        if (rc) return rc;
    }
    if (pend_inv && may_defer && L.kind == 2) *pend_inv = prm + off.anic_inv;
    else {
        ProfScope ps(IFL_K_ANIC, st);
        rc = ifl_k_anic(lz.z, prm + off.anic_inv, w.B, L.C, lv.H, lv.W, 1, sf_next, st);
    }
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

// ---- stream plan ------------------------------------------------------------------------------------------------------
// A call is split into two kinds of work (DESIGN.md "schedule"):
//   phase 1  conv1_ft_kernel: the z-independent ~98 % of the FLOPs, for all conditional steps, launched in chunks of
//            IFL_P1_CHUNK steps on an auxiliary stream in the order the chain will need them;
//   chain    everything that depends on the running z (ActNorm+1x1, fAffine's z1 taps, the last conv + coupling), step by
//            step.  A step waits (event) for the phase-1 chunk that holds its slab, so the light, latency-bound chain
//            runs underneath the next heavy chunk instead of leaving the chip idle between small kernels.
// Samples are independent: the chain may further be cut into contiguous sample groups, each on its
// own stream, forked from and joined back to the caller's stream with events.
// The auxiliary streams and events are created once per device on first use and kept for the life of the process.
#define IFL_MAX_GROUPS 4
#ifndef IFL_CONV1_L2_DEFAULT
#define IFL_CONV1_L2_DEFAULT 2
#endif
#define IFL_P1_CHUNK 4         // level 1: 4 steps x 2 networks = one (network, step) pair per XCD and launch (conv1x.hip)
#define IFL_P1_CHUNK_EARLY 8      // level 1 of a pass started with IFL_FLOW_COND_EARLY: its first slabs are ready before its chain can start anyway,
                                  // so longer launches cost nothing (same patches/s) and the dominant launch runs at 0.63 instead of 0.61 in situ
#define IFL_P1_CHUNK_L2 6         // deeper levels: 12 F(4x4) steps = two launches of 1 200 workgroups (4.7 rounds), the 4 tail steps one of their own
namespace {
int env_int(const char* name, int dflt);
constexpr int MAX_P1_EVENTS = IFL_MAX_LEVELS * IFL_MAX_COND;
struct ForkCtx {
    std::mutex mu;       // held while a driver call enqueues its work: the events below are re-recorded by every call, and a
                         // wait captures an event's state when it is enqueued, so calls from several host threads / caller
                         // streams on one device are correct as long as their enqueue sequences do not interleave
    bool ok = false;
    hipStream_t aux, s[IFL_MAX_GROUPS - 1];
    int n_s = 0;         // group streams created so far
    hipEvent_t fork, aux_done, early, join[IFL_MAX_GROUPS - 1], p1[MAX_P1_EVENTS];
};
ForkCtx g_fork[IFL_MAX_DEVICES];

// the context of the current device; the caller holds f->mu from here to the end of its enqueue sequence
int fork_ctx(ForkCtx** out, std::unique_lock<std::mutex>* lock)
{
    int dev = 0;
    IFL_HIP(hipGetDevice(&dev));
    IFL_CHECK_ARG(dev >= 0 && dev < IFL_MAX_DEVICES, IFL_ENOTSUP, "device index %d", dev);
    ForkCtx& f = g_fork[dev];
    *lock = std::unique_lock<std::mutex>(f.mu);
    if (!f.ok) {
        int lo = 0, hi = 0;
        IFL_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = least urgent
        // the bulk stream gets the high priority: its long workgroups then take a freed CU before the chain's short ones,
        // which measures 1 % faster than the opposite choice (IFL_AUX_PRIO_HIGH=0) with the F(4x4) bulk kernel
        IFL_HIP(hipStreamCreateWithPriority(&f.aux, hipStreamNonBlocking, env_int("IFL_AUX_PRIO_HIGH", 1) ? hi : lo));
        // (the sample groups' streams are created when a call first uses them: HIP deals its streams to 4 hardware queues
        // (GPU_MAX_HW_QUEUES) in creation order, and two busy streams on one queue serialise -- a sixth stream for the early
        // conditioning loads, sharing the bulk stream's queue, cost 7 % of the step)
        for (int i = 0; i < IFL_MAX_GROUPS - 1; ++i) IFL_HIP(hipEventCreateWithFlags(&f.join[i], hipEventDisableTiming));
        for (int i = 0; i < MAX_P1_EVENTS; ++i) IFL_HIP(hipEventCreateWithFlags(&f.p1[i], hipEventDisableTiming));
        IFL_HIP(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
        IFL_HIP(hipEventCreateWithFlags(&f.aux_done, hipEventDisableTiming));
        IFL_HIP(hipEventCreateWithFlags(&f.early, hipEventDisableTiming));
        f.ok = true;
    }
    *out = &f;
    return 0;
}

int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// schedule knobs: defaults, overridable by the environment (read once) and by ifl_set_schedule()
struct Sched { int aux, groups, chunk, chunk2, chunk_early; };      // chunk: level-1 steps per bulk launch, chunk2: deeper levels,
                                                                   // chunk_early: level 1 of a pass whose bulk work starts early
Sched& sched()
{
    auto clamp = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };      // as ifl_set_schedule validates
    static Sched s{env_int("IFL_AUX_STREAM", 1) != 0, clamp(env_int("IFL_GROUPS", 2), 1, IFL_MAX_GROUPS),
                   clamp(env_int("IFL_P1_CHUNK_STEPS", IFL_P1_CHUNK), 1, IFL_MAX_COND),
                   clamp(env_int("IFL_P1_CHUNK_STEPS_L2", IFL_P1_CHUNK_L2), 1, IFL_MAX_COND),
                   clamp(env_int("IFL_P1_CHUNK_STEPS_EARLY", env_int("IFL_P1_CHUNK_STEPS", IFL_P1_CHUNK_EARLY)), 1, IFL_MAX_COND)};
    return s;
}

// algorithm of the bulk kernel (fp32) per flow level: 2 = Winograd F(4x4,3x3), 1 = Winograd F(2x2,3x3), 0 = direct implicit GEMM.
// Default 2 at the first level; at the deeper levels IFL_CONV1_WINO_L2 (default: see ifl_set_conv1_algo_level in the header).
int& conv1_algo(int level = 1)
{
    static int a[2] = {[] { int v = env_int("IFL_CONV1_WINO", 2); return v < 0 || v > 2 ? 2 : v; }(),
                       [] { int v = env_int("IFL_CONV1_WINO_L2", env_int("IFL_CONV1_WINO", IFL_CONV1_L2_DEFAULT)); return v < 0 || v > 2 ? 2 : v; }()};
    return a[level <= 1 ? 0 : 1];
}

// The LAST conditional steps of the LAST level by a more accurate form: {number of steps, algorithm}.  The decode runs those
// steps first, and everything behind them amplifies what they add: in an fp32 emulation of the stack, 62 % of the rounding-noise
// variance that F(4x4) adds to a decoded image enters in the last 4 of the 32 conditional steps, 86 % in the last 8, and
// nothing measurable at the first level (profiles/r03_notes.md).  F(2x2) there costs 2.7 % of a pass.
int* conv1_tail()
{
    static int t[2] = {[] { int v = env_int("IFL_CONV1_TAIL_STEPS", 4); return v < 0 ? 0 : v > IFL_MAX_COND ? IFL_MAX_COND : v; }(),
                       [] { int v = env_int("IFL_CONV1_TAIL_ALGO", 1); return v < 0 || v > 2 ? 1 : v; }()};
    return t;
}

// The z-independent work of the conditional steps [s0, s0 + nsteps) of a level (`steps` lists their parameter blocks):
// both networks' first conv over the conditioning features into the h2 slabs, then fFeatures' last conv -> sf slabs and
// its log-det sums.
int conv1_steps(const IflWs& w, const IflLevelWs& lv, float* h2, const float* params, const IflStepList& steps, int nsteps,
                const IflStepOff& off, int level, int bf16, int algo, hipStream_t st)
{
    if (nsteps <= 0) return 0;
    if (!bf16 && algo == 2) return ifl_k_conv1_ft_wino4(lv.vg, h2, lv.h2_step, params, steps, nsteps, off, w.B, lv.H, lv.W, level, st);
    if (!bf16 && algo == 1) return ifl_k_conv1_ft_wino(lv.ft, h2, lv.h2_step, params, steps, nsteps, off, w.B, lv.H, lv.W, level, st);
    return ifl_k_conv1_ft(lv.ft, h2, lv.h2_step, params, steps, nsteps, off, lv.geom, w.B, level, bf16, st);
}

// n_level: conditional steps of the level (the steps [n_level - tail, n_level) of the last level take the tail's algorithm)
int bulk_steps(const IflWs& w, const IflLevelWs& lv, int s0, const float* params, const IflStepList& steps, int nsteps,
               const IflStepOff& off, int level, int bf16, float eps, hipStream_t st, int n_level = 0)
{
    float* h2 = lv.h2 + (int64_t)s0 * lv.h2_step;
    int rc;
    { ProfScope ps(level == 1 ? IFL_K_CONV1_FT_L1 : IFL_K_CONV1_FT_L2, st);
      const int algo = conv1_algo(level), talgo = conv1_tail()[1];
      int t0 = nsteps;                          // first step of this launch (relative) that belongs to the tail
      if (level == IFL_MAX_LEVELS && n_level > 0 && !bf16 && talgo != algo) {
          t0 = n_level - conv1_tail()[0] - s0;
          t0 = t0 < 0 ? 0 : t0 > nsteps ? nsteps : t0;
      }
      rc = conv1_steps(w, lv, h2, params, steps, t0, off, level, bf16, algo, st);
      if (!rc && t0 < nsteps) {
          IflStepList tail{};
          for (int s = t0; s < nsteps; ++s) tail.off[s - t0] = steps.off[s];
          rc = conv1_steps(w, lv, h2 + (int64_t)t0 * lv.h2_step, params, tail, nsteps - t0, off, level, bf16, talgo, st);
      } }
    if (rc) return rc;
    ProfScope ps(IFL_K_CONV3_F, st);
    if (conv3_algo() == 1 || bf16)
        return ifl_k_conv3w(0, h2 + IFL_HID, lv.h2_step, IFL_H2, lv.sf + (int64_t)s0 * lv.sf_step, lv.sf_step, nullptr, w.ldpart,
                            w.ld_slots, lv.ld_f0 + s0 * ifl_w3_slots(lv.C, lv.H, lv.W, 1), params, steps, nsteps, off, w.B, lv.H, lv.W,
                            lv.C, eps, bf16, st);
    return ifl_k_conv3f(h2, lv.h2_step, lv.sf + (int64_t)s0 * lv.sf_step, lv.sf_step, w.ldpart, w.ld_slots,
                        lv.ld_f0 + s0 * lv.geom.ntiles, params, steps, nsteps, off, lv.geom, w.B, lv.C, eps, st);
}

// log-det slots per conditional step that fFeatures' kernel of the current algorithm writes
int f_slots(const IflLevelWs& lv, int bf16) { return conv3_algo() == 1 || bf16 ? ifl_w3_slots(lv.C, lv.H, lv.W, 1) : lv.geom.ntiles; }

int n_groups(int B)
{
    int g = sched().groups;
    if (g > IFL_MAX_GROUPS) g = IFL_MAX_GROUPS;
    if (g > B) g = B;
    return g < 1 ? 1 : g;
}

IflWs ws_slice(const IflWs& w, int b0, int nb)
{
    IflWs s = w;
    s.B = nb;
    for (int l = 0; l < IFL_MAX_LEVELS; ++l) {
        IflLevelWs& lv = s.lv[l];
        int64_t ring = (int64_t)b0 * (lv.H + 2) * (lv.W + 2);
        lv.ft += ring * IFL_FT; lv.z += ring * lv.C; lv.z2 += ring * lv.C; lv.h2 += ring * IFL_H2; lv.ha += ring * IFL_HID;     // h2_step, sf_step stay the full-batch slab strides
        lv.sf += ring * 2 * lv.C;
    }
    s.ldpart += (int64_t)b0 * w.ld_slots;
    return s;
}

// the conditional steps of each level, in layer order
struct CondPlan { int n[IFL_MAX_LEVELS]; IflStepList list[IFL_MAX_LEVELS]; int slab_of_layer[256]; };

int make_plan(const ifl_layer* layers, int n_layers, CondPlan* p)
{
    IFL_CHECK_ARG(n_layers <= 256, IFL_ENOTSUP, "more than 256 layers");
    int lvl = -1;
    for (int l = 0; l < IFL_MAX_LEVELS; ++l) p->n[l] = 0;
    for (int i = 0; i < n_layers; ++i) {
        p->slab_of_layer[i] = -1;
        if (layers[i].kind == 0) ++lvl;
        else if (layers[i].kind == 2) {
            p->slab_of_layer[i] = p->n[lvl];
            p->list[lvl].off[p->n[lvl]++] = layers[i].param_off;
        }
    }
    return 0;
}

// phase 1 of one level: chunks of IFL_P1_CHUNK steps in the order given by `reverse`; event e0 + chunk index marks each
int launch_phase1(const IflWs& w, int lvl, const CondPlan& plan, const float* params, bool reverse, hipStream_t st,
                  hipEvent_t* ev, int bf16, float eps, int chunk_override = 0)
{
    const IflLevelWs& lv = w.lv[lvl];
    const int n = plan.n[lvl];
    const int chunk = chunk_override > 0 ? chunk_override : (lvl == 0 ? sched().chunk : sched().chunk2);
    IflStepOff off = ifl_step_off(lv.C, 1);
    const int nchunks = ifl_cdiv(n, chunk);
    for (int c = 0; c < nchunks; ++c) {
        const int ci = reverse ? nchunks - 1 - c : c;             // chunk holding the steps the chain needs next
        const int s0 = ci * chunk, s1 = s0 + chunk < n ? s0 + chunk : n;
        IflStepList sub{};
        for (int s = s0; s < s1; ++s) sub.off[s - s0] = plan.list[lvl].off[s];
        int rc = bulk_steps(w, lv, s0, params, sub, s1 - s0, off, lvl + 1, bf16, eps, st, n);
        if (rc) return rc;
        if (ev) IFL_HIP(hipEventRecord(ev[lvl * IFL_MAX_COND + ci], st));
    }
    return 0;
}

int chain_group(bool decode, const ifl_layer* layers, int n_layers, const float* params, const CondPlan& plan,
                const IflWs& w, const float* src, float* dst, const float* logdet_in, float* logdet_out, float eps, int bf16,
                hipStream_t st, hipEvent_t* ev, int chunk1)
{
    const int B = w.B, H = w.H, W = w.W;
    const int chunk_of[2] = {chunk1, sched().chunk2};
    int rc = 0, slots = 0;
    double constant = 0.0;
    int waited[IFL_MAX_LEVELS][IFL_MAX_COND] = {};
    auto wait_slab = [&](int lvl, int slab) -> int {
        if (!ev) return 0;
        const int ci = slab / chunk_of[lvl == 0 ? 0 : 1];
        if (!waited[lvl][ci]) { IFL_HIP(hipStreamWaitEvent(st, ev[lvl * IFL_MAX_COND + ci], 0)); waited[lvl][ci] = 1; }
        return 0;
    };
    if (!decode) {
        int lvl = -1;
        IflWs wm = w;                             // lv[].z / lv[].z2 change roles after every step that maps z out of place
        for (int i = 0; i < n_layers; ++i) {
            const ifl_layer& L = layers[i];
            if (L.kind == 0) {
                ++lvl;
                if (lvl == 0) rc = ifl_k_squeeze_in(src, wm.lv[0].z, B, 3, H, W, st);
                else rc = ifl_k_squeeze_ring(wm.lv[lvl - 1].z, wm.lv[lvl].z, B, wm.lv[lvl - 1].C, wm.lv[lvl - 1].H, wm.lv[lvl - 1].W, st);
            } else {
                const int slab = plan.slab_of_layer[i];
                if (slab >= 0 && (rc = wait_slab(lvl, slab))) return rc;
                int swapped = 0;
                rc = step_encode(L, params, wm, wm.lv[lvl], slab < 0 ? 0 : slab, slots, constant, eps, (bf16 >> 1) & 1, st, &swapped);
                if (swapped) { float* tz = wm.lv[lvl].z; wm.lv[lvl].z = wm.lv[lvl].z2; wm.lv[lvl].z2 = tz; }
            }
            if (rc) return rc;
        }
        const IflLevelWs& last = wm.lv[lvl];
        if ((rc = ifl_k_ring_to_nchw(last.z, dst, B, last.C, last.H, last.W, st))) return rc;
    } else {
        int lvl = IFL_MAX_LEVELS - 1;
        IflWs wm = w;                             // lv[].z / lv[].z2 change roles after every step that maps z out of place
        const IflLevelWs& last = wm.lv[lvl];
        if ((rc = ifl_k_nchw_to_ring(src, last.z, B, last.C, last.H, last.W, st))) return rc;
        bool fapply_done = false;
        const float* pend_inv = nullptr;          // a deferred inverse ActNorm + 1x1 block (chain fusion)
        for (int i = n_layers - 1; i >= 0; --i) {
            const ifl_layer& L = layers[i];
            if (L.kind == 0) {
                if (lvl == 0) rc = ifl_k_unsqueeze_out(wm.lv[0].z, dst, B, 3, H, W, st);
                else rc = ifl_k_unsqueeze_ring(wm.lv[lvl].z, wm.lv[lvl - 1].z, B, wm.lv[lvl - 1].C, wm.lv[lvl - 1].H, wm.lv[lvl - 1].W, st);
                --lvl;
            } else {
                const int slab = plan.slab_of_layer[i];
                if (slab >= 0 && (rc = wait_slab(lvl, slab))) return rc;
                // the layer the chain runs next, if it is a conditional step of the same level: its first half (fapply) is
                // pointwise on the same pixels as this step's last kernel (ActNorm + 1x1 inverse), so it is fused into it
                const float* sf_next = nullptr;
                if (i > 0 && layers[i - 1].kind == 2) {
                    const int nslab = plan.slab_of_layer[i - 1];
                    if ((rc = wait_slab(lvl, nslab))) return rc;
                    sf_next = wm.lv[lvl].sf + (int64_t)nslab * wm.lv[lvl].sf_step;
                }
                const int cbf = (bf16 >> 1) & 1;
                const bool may_defer = sf_next != nullptr && !cbf && L.C == 48 && (fuse_anic() & 2);
                int swapped = 0;
                rc = step_decode(L, params, wm, wm.lv[lvl], slab < 0 ? 0 : slab, slots, constant, eps, cbf, st, fapply_done, sf_next,
                                 &pend_inv, may_defer, &swapped);
                if (swapped) { float* tz = wm.lv[lvl].z; wm.lv[lvl].z = wm.lv[lvl].z2; wm.lv[lvl].z2 = tz; }
                fapply_done = sf_next != nullptr;
            }
            if (rc) return rc;
        }
    }
    // fFeatures' sums of log(scale) (phase 1): subtracted by the encode direction, added by the decode direction
    return ifl_k_ld_finalize(w.ldpart, slots, w.ld_slots, logdet_in, logdet_out, constant, decode ? 1.0 : -1.0,
                             w.lv[0].ld_f0, plan.n[0] * f_slots(w.lv[0], bf16 & 1), w.lv[1].ld_f0, plan.n[1] * f_slots(w.lv[1], bf16 & 1), B, st);
}

// early: the conditioning was put into `ws` by ifl_load_cond_early (bulk stream); the bulk work then follows it there without
// waiting for the caller's stream -- it reads nothing that the caller's stream produces
int run_flow(bool decode, bool prepared, int bf16, const ifl_layer* layers, int n_layers, const float* params, void* ws,
             const float* src, float* dst, const float* logdet_in, float* logdet_out, int B, int H, int W, float eps,
             hipStream_t st, bool early = false)
{
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    CondPlan plan;
    int rc = make_plan(layers, n_layers, &plan);
    if (rc) return rc;
    const int use_aux = prepared ? 0 : sched().aux;     // prepared: phase 1 was run by ifl_cond_prepare, the slabs are ready
    const int G = n_groups(B);
    const int64_t n_img = (int64_t)3 * H * W, n_lat = (int64_t)48 * (H / 4) * (W / 4);
    const int64_t n_src = decode ? n_lat : n_img, n_dst = decode ? n_img : n_lat;
    ForkCtx* f = nullptr;
    std::unique_lock<std::mutex> fork_lock;
    hipEvent_t* ev = nullptr;
    // (every pass records f->fork at its entry: ifl_load_cond_early orders the bulk stream behind the previous pass's entry)
    if ((rc = fork_ctx(&f, &fork_lock))) return rc;
    IFL_HIP(hipEventRecord(f->fork, st));
    if (use_aux) {
        ev = f->p1;
        if (!early) IFL_HIP(hipStreamWaitEvent(f->aux, f->fork, 0));
    } else if (early) {
        IFL_HIP(hipStreamWaitEvent(st, f->early, 0));             // one-stream schedule: behind the early load, no overlap
    }
    hipStream_t p1s = use_aux ? f->aux : st;
    const int chunk1 = early && use_aux ? sched().chunk_early : sched().chunk;      // level-1 steps per bulk launch of this pass
    // phase 1 in the order the chain consumes the levels
    for (int k = 0; k < IFL_MAX_LEVELS && !prepared; ++k) {
        const int lvl = decode ? IFL_MAX_LEVELS - 1 - k : k;
        if ((rc = launch_phase1(w, lvl, plan, params, decode, p1s, ev, bf16 & 1, eps, lvl == 0 ? chunk1 : 0))) return rc;
    }
    if (use_aux) IFL_HIP(hipEventRecord(f->aux_done, f->aux));
    else if (G > 1) IFL_HIP(hipEventRecord(f->fork, st));        // groups start after phase 1 on the caller's stream
    for (; f->n_s < G - 1; ++f->n_s) IFL_HIP(hipStreamCreateWithFlags(&f->s[f->n_s], hipStreamNonBlocking));
    for (int g = 0; g < G; ++g) {
        const int b0 = (int)((int64_t)B * g / G), b1 = (int)((int64_t)B * (g + 1) / G);
        hipStream_t sg = g == 0 ? st : f->s[g - 1];
        if (g > 0) IFL_HIP(hipStreamWaitEvent(sg, f->fork, 0));
        IflWs wg = ws_slice(w, b0, b1 - b0);
        const float* li = logdet_in ? logdet_in + b0 : nullptr;
        if ((rc = chain_group(decode, layers, n_layers, params, plan, wg, src + b0 * n_src, dst + b0 * n_dst, li,
                              logdet_out + b0, eps, bf16, sg, ev, chunk1))) return rc;
        if (g > 0) {
            IFL_HIP(hipEventRecord(f->join[g - 1], sg));
            IFL_HIP(hipStreamWaitEvent(st, f->join[g - 1], 0));
        }
    }
    if (use_aux) IFL_HIP(hipStreamWaitEvent(st, f->aux_done, 0));     // also joins when no step waited on phase 1
    return 0;
}
}  // namespace

extern "C" int ifl_set_conv1_algo(int algo)
{
    int prev = conv1_algo(1);
    if (algo >= 0 && algo <= 2) conv1_algo(1) = conv1_algo(2) = algo;
    return prev;
}

extern "C" int ifl_set_conv1_tail(int nsteps, int algo)
{
    int prev = conv1_tail()[0];
    if (nsteps >= 0 && nsteps <= IFL_MAX_COND) conv1_tail()[0] = nsteps;
    if (algo >= 0 && algo <= 2) conv1_tail()[1] = algo;
    return prev;
}

extern "C" int ifl_set_conv1_algo_level(int level, int algo)
{
    int prev = conv1_algo(level);
    if (algo >= 0 && algo <= 2) conv1_algo(level) = algo;
    return prev;
}

extern "C" int ifl_set_chain_fusion(int on)
{
    int prev = fuse_anic();
    if (on >= 0 && on <= 3) fuse_anic() = on;
    return prev;
}

extern "C" int ifl_set_conv3_algo(int algo)
{
    int prev = conv3_algo();
    if (algo == 0 || algo == 1) conv3_algo() = algo;
    return prev;
}

extern "C" int ifl_set_schedule(int aux_stream, int groups, int p1_chunk_steps)
{
    IFL_CHECK_ARG(groups <= IFL_MAX_GROUPS && p1_chunk_steps <= IFL_MAX_COND, IFL_EINVAL,
                  "groups <= %d, chunk <= %d", IFL_MAX_GROUPS, IFL_MAX_COND);
    Sched& s = sched();
    if (aux_stream >= 0) s.aux = aux_stream != 0;
    if (groups >= 1) s.groups = groups;
    if (p1_chunk_steps >= 1) s.chunk = s.chunk2 = s.chunk_early = p1_chunk_steps;
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
    return run_flow(false, false, 0, layers, n_layers, params, ws, x, z, logdet_in, logdet_out, B, H, W, eps, (hipStream_t)stream);
}

extern "C" int ifl_flow_decode(const ifl_layer* layers, int n_layers, const float* params, void* ws,
                               const float* z, float* x, const float* logdet_in, float* logdet_out,
                               int B, int H, int W, float eps, void* stream)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    if ((rc = check_layers(layers, n_layers))) return rc;
    IFL_CHECK_ARG(params && ws && x && z && logdet_out, IFL_EINVAL, "null pointer");
    return run_flow(true, false, 0, layers, n_layers, params, ws, z, x, logdet_in, logdet_out, B, H, W, eps, (hipStream_t)stream);
}

extern "C" int ifl_cond_prepare(const ifl_layer* layers, int n_layers, const float* params, void* ws, int B, int H, int W,
                                float eps, int flags, void* stream)
{
    IFL_CHECK_ARG((flags & ~(IFL_FLOW_BF16_CONV1 | IFL_FLOW_BF16_CHAIN)) == 0, IFL_EINVAL, "unknown flags 0x%x", flags);
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    if ((rc = check_layers(layers, n_layers))) return rc;
    IFL_CHECK_ARG(params && ws, IFL_EINVAL, "null pointer");
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    CondPlan plan;
    if ((rc = make_plan(layers, n_layers, &plan))) return rc;
    for (int lvl = 0; lvl < IFL_MAX_LEVELS; ++lvl)      // one launch per level: all conditional steps at once
        if ((rc = launch_phase1(w, lvl, plan, params, false, (hipStream_t)stream, nullptr, (flags & IFL_FLOW_BF16_CONV1) != 0, eps,
                                IFL_MAX_COND))) return rc;
    return 0;
}

extern "C" int ifl_flow_pass(const ifl_layer* layers, int n_layers, const float* params, void* ws, const float* src,
                             float* dst, const float* logdet_in, float* logdet_out, int B, int H, int W, float eps,
                             int decode, int flags, void* stream)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    if ((rc = check_layers(layers, n_layers))) return rc;
    IFL_CHECK_ARG(params && ws && src && dst && logdet_out, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG((flags & ~(IFL_FLOW_COND_PREPARED | IFL_FLOW_BF16_CONV1 | IFL_FLOW_BF16_CHAIN | IFL_FLOW_COND_EARLY)) == 0, IFL_EINVAL, "unknown flags 0x%x", flags);
    IFL_CHECK_ARG(!(flags & IFL_FLOW_BF16_CHAIN) || (flags & IFL_FLOW_BF16_CONV1), IFL_EINVAL, "IFL_FLOW_BF16_CHAIN needs IFL_FLOW_BF16_CONV1");
    IFL_CHECK_ARG(!(flags & IFL_FLOW_COND_EARLY) || !(flags & IFL_FLOW_COND_PREPARED), IFL_EINVAL, "IFL_FLOW_COND_EARLY with IFL_FLOW_COND_PREPARED: nothing left to start early");
    // bf16 mask inside the drivers: bit 0 = the z-independent convolutions (bulk phase), bit 1 = fAffine's z-dependent ones (chain)
    return run_flow(decode != 0, (flags & IFL_FLOW_COND_PREPARED) != 0, ((flags & IFL_FLOW_BF16_CONV1) ? 1 : 0) | ((flags & IFL_FLOW_BF16_CHAIN) ? 2 : 0), layers, n_layers, params, ws, src, dst, logdet_in,
                    logdet_out, B, H, W, eps, (hipStream_t)stream, (flags & IFL_FLOW_COND_EARLY) != 0);
}

extern "C" int ifl_load_cond_early(void* ws, int B, int H, int W, int level, const float* ft_nchw, void* after_event)
{
    int rc = check_shape(B, H, W);
    if (rc) return rc;
    IFL_CHECK_ARG(ws && ft_nchw, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(level >= 1 && level <= IFL_MAX_LEVELS, IFL_EINVAL, "level %d outside 1..%d", level, IFL_MAX_LEVELS);
    IflWs w;
    ifl_ws_layout(ws, B, H, W, &w);
    ForkCtx* f = nullptr;
    std::unique_lock<std::mutex> fork_lock;
    if ((rc = fork_ctx(&f, &fork_lock))) return rc;
    // behind the entry of the previous pass on its calling stream (where, by the caller's promise, the features were complete)
    // -- the bulk stream is already there unless that pass ran on the one-stream schedule -- and behind `after_event`.
    // (On a stream of their own the conversions run beside the previous pass's bulk kernels: +0.5 % of the step, but they
    // stretch the dominant launch from 1.64 to 1.71 ms; measured with 4 streams in all, see fork_ctx -- not kept.)
    IFL_HIP(hipStreamWaitEvent(f->aux, f->fork, 0));
    if (after_event) IFL_HIP(hipStreamWaitEvent(f->aux, (hipEvent_t)after_event, 0));
    if ((rc = load_ft(w.lv[level - 1], B, ft_nchw, f->aux))) return rc;
    IFL_HIP(hipEventRecord(f->early, f->aux));
    return 0;
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
    if (layer->kind == 2 && (rc = load_ft(lv, B, ft, st))) return rc;
    int slots = 0;
    double constant = 0.0;
    if (layer->kind == 2) {
        IflStepList one{};
        one.off[0] = layer->param_off;
        if ((rc = bulk_steps(w, lv, 0, params, one, 1, ifl_step_off(lv.C, 1), lvl + 1, 0, eps, st))) return rc;
    }
    rc = reverse ? step_decode(*layer, params, w, lv, 0, slots, constant, eps, 0, st)
                 : step_encode(*layer, params, w, lv, 0, slots, constant, eps, 0, st);
    if (rc) return rc;
    if ((rc = ifl_k_ring_to_nchw(lv.z, z_out, B, lv.C, lv.H, lv.W, st))) return rc;
    return ifl_k_ld_finalize(w.ldpart, slots, w.ld_slots, logdet_in, logdet_out, constant, reverse ? 1.0 : -1.0,
                             lv.ld_f0, layer->kind == 2 ? f_slots(lv, 0) : 0, 0, 0, B, st);
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

static int conv2d_impl(const ifl_conv_desc* d, int nz, const int64_t* zs, int B, int h, int w, void* stream);

extern "C" int ifl_conv2d(const ifl_conv_desc* d, int B, int h, int w, void* stream)
{
    return conv2d_impl(d, 1, nullptr, B, h, w, stream);
}

extern "C" int ifl_conv2d_batched(const ifl_conv_desc* d, int nz, int64_t src0_z, int64_t src1_z, int64_t w_z, int64_t bias_z,
                                  int64_t out_z, int64_t res1_z, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(nz >= 1 && src0_z % 4 == 0 && src1_z % 4 == 0 && w_z % 4 == 0 && bias_z % 4 == 0 && out_z % 4 == 0 && res1_z % 4 == 0,
                  IFL_EINVAL, "batched conv: %d problems, strides must be multiples of 4 floats", nz);
    IFL_CHECK_ARG(d && !d->res2, IFL_ENOTSUP, "batched conv: no second residual");
    const int64_t zs[6] = {src0_z, src1_z, w_z, bias_z, out_z, res1_z};
    return conv2d_impl(d, nz, zs, B, h, w, stream);
}

static int conv2d_impl(const ifl_conv_desc* d, int nz, const int64_t* zs, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(d && d->w && d->out && d->src[0], IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 1 && B <= 65535 && h >= 1 && w >= 1, IFL_EINVAL, "bad shape B=%d h=%d w=%d", B, h, w);
    IFL_CHECK_ARG(d->nsrc == 1 || (d->nsrc == 2 && d->src[1]), IFL_EINVAL, "nsrc=%d", d->nsrc);
    IflConvArgs a{};
    a.nsrc = d->nsrc;
    for (int s = 0; s < d->nsrc; ++s) {
        IFL_CHECK_ARG(d->src_ch[s] > 0 && d->src_ch[s] % 32 == 0 && d->src_off[s] % 4 == 0 && d->src_stride[s] % 4 == 0, IFL_EINVAL,
                      "source %d: %d channels at offset %d, stride %d", s, d->src_ch[s], d->src_off[s], d->src_stride[s]);
        a.src[s] = d->src[s]; a.src_stride[s] = d->src_stride[s]; a.src_off[s] = d->src_off[s]; a.src_chunks[s] = d->src_ch[s] / 32;
        a.nchunks += d->src_ch[s] / 32;
    }
    IFL_CHECK_ARG(d->out_off % 4 == 0 && d->out_stride % 4 == 0, IFL_EINVAL, "destination offset/stride must be multiples of 4");
    a.w = d->w; a.bias = d->bias;
    a.out = d->out; a.out_stride = d->out_stride; a.out_off = d->out_off; a.cout = d->cout; a.taps = d->taps; a.act = d->act;
    a.res1 = d->res1; a.res1_stride = d->res1_stride; a.res1_off = d->res1_off; a.alpha = d->alpha;
    a.res2 = d->res2; a.res2_stride = d->res2_stride; a.res2_off = d->res2_off; a.beta = d->beta;
    a.g = ifl_choose_geom(h, w);
    if (nz > 1) {
        a.nz = nz; a.Bz = B;
        a.z_src[0] = zs[0]; a.z_src[1] = zs[1]; a.z_w = zs[2]; a.z_bias = zs[3]; a.z_out = zs[4]; a.z_res1 = zs[5];
    }
    return ifl_k_convg(a, B, (hipStream_t)stream);
}

extern "C" int ifl_conv2d_wgrad(const float* x, int x_stride, int x_off, int cin, const float* dy, int dy_stride, int dy_off,
                                int cout, float* dW, int cin_total, int ci0, int taps, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(x && dy && dW, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 1 && h >= 1 && w >= 1, IFL_EINVAL, "bad shape B=%d h=%d w=%d", B, h, w);
    IFL_CHECK_ARG(ci0 >= 0 && (ci0 * 32 + cin) <= cin_total, IFL_EINVAL, "channel range outside dW");
    IFL_CHECK_ARG(x_stride % 4 == 0 && x_off % 4 == 0 && dy_stride % 4 == 0 && dy_off % 4 == 0, IFL_EINVAL,
                  "strides and offsets must be multiples of 4 (16-byte loads)");
    return ifl_k_wgrad(x, x_stride, x_off, cin, dy, dy_stride, dy_off, cout, dW, cin_total, ci0, taps, ifl_choose_geom(h, w), B,
                       (hipStream_t)stream);
}

extern "C" int ifl_conv2d_wgrad_batched(const float* x, int x_stride, int x_off, int cin, int64_t x_zstride, const float* dy, int dy_stride,
                                        int dy_off, int cout, int64_t dy_zstride, float* dW, int cin_total, int ci0, int64_t dw_zstride,
                                        int nz, int taps, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(x && dy && dW, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 1 && h >= 1 && w >= 1, IFL_EINVAL, "bad shape B=%d h=%d w=%d", B, h, w);
    IFL_CHECK_ARG(ci0 >= 0 && (ci0 * 32 + cin) <= cin_total, IFL_EINVAL, "channel range outside dW");
    IFL_CHECK_ARG(x_stride % 4 == 0 && x_off % 4 == 0 && dy_stride % 4 == 0 && dy_off % 4 == 0 && x_zstride % 4 == 0 && dy_zstride % 4 == 0,
                  IFL_EINVAL, "strides and offsets must be multiples of 4 (16-byte loads)");
    return ifl_k_wgrad(x, x_stride, x_off, cin, dy, dy_stride, dy_off, cout, dW, cin_total, ci0, taps, ifl_choose_geom(h, w), B,
                       (hipStream_t)stream, nz, x_zstride, dy_zstride, dw_zstride);
}

extern "C" int ifl_quantize_u8_nhwc(const float* x, uint8_t* out, int B, int C, int H, int W, void* stream)
{
    IFL_CHECK_ARG(B >= 0 && C >= 1 && H >= 0 && W >= 0, IFL_EINVAL, "bad shape");
    IFL_CHECK_ARG((x && out) || (int64_t)B * H * W == 0, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B <= 65535, IFL_ENOTSUP, "B=%d exceeds the grid's y extent", B);
    return ifl_k_quantize_u8(x, out, B, C, (int64_t)H * W, (hipStream_t)stream);
}

extern "C" int ifl_resize_sep(const float* x, float* out, int N, int Hin, int Win, int Hout, int Wout, const int* idxH, const float* wH,
                              int TH, const int* idxW, const float* wW, int TW, void* stream)
{
    IFL_CHECK_ARG((x && out) || (int64_t)N * Hout * Wout == 0, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(idxH && wH && idxW && wW, IFL_EINVAL, "null tap table");
    IFL_CHECK_ARG(N >= 0 && N <= 65535 && Hin >= 1 && Win >= 1 && Hout >= 0 && Wout >= 0, IFL_EINVAL, "bad shape");
    IFL_CHECK_ARG(TH >= 1 && TW >= 1 && TH <= 64 && TW <= 64, IFL_EINVAL, "taps %d, %d outside 1..64", TH, TW);
    IFL_CHECK_ARG(Win <= 8192, IFL_ENOTSUP, "input rows of %d values exceed the LDS row buffer", Win);
    {   // the kernel asks for 4 rows x Win floats of dynamic LDS: 128 KiB at the limit, inside gfx950's 160 KiB per workgroup
        int dev = 0, lds = 0;
        IFL_HIP(hipGetDevice(&dev));
        IFL_HIP(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        IFL_CHECK_ARG(16 * (int64_t)Win <= lds, IFL_ENOTSUP, "input rows of %d values need %lld bytes of LDS, the device has %d per workgroup",
                      Win, (long long)(16 * (int64_t)Win), lds);
    }
    return ifl_k_resize_sep(x, out, N, Hin, Win, Hout, Wout, idxH, wH, TH, idxW, wW, TW, (hipStream_t)stream);
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

extern "C" int ifl_gaussian_logp_static(const float* z, const float* mean, const float* cov_inv, float logdet_cov, float* out,
                                        int B, int C, int H, int W, void* stream)
{
    IFL_CHECK_ARG(z && out, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(C >= 1 && C <= 64, IFL_ENOTSUP, "C=%d outside 1..64", C);
    IFL_CHECK_ARG(cov_inv == nullptr || mean != nullptr, IFL_EINVAL, "cov_inv without mean");
    IFL_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, IFL_EINVAL, "bad shape");
    return ifl_k_logp_static(z, mean, cov_inv, logdet_cov, out, B, C, H * W, (hipStream_t)stream);
}

extern "C" int ifl_lerp(const float* za, const float* zb, const float* wa, const float* wb, float* out, int B,
                        int64_t n_per_sample, void* stream)
{
    IFL_CHECK_ARG(za && zb && wa && wb && out, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 0 && n_per_sample >= 0, IFL_EINVAL, "bad size");
    return ifl_k_lerp(za, zb, wa, wb, out, B, n_per_sample, (hipStream_t)stream);
}
