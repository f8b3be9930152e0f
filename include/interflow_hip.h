/*
 * interflow_hip.h -- C ABI of libinterflow_hip.so: the MI355X (gfx950) implementation of
 * InterFlow's conditional normalizing-flow forward / inverse pass.
 *
 * The reference (dongjinkim9/InterFlow) is pure PyTorch: it has no FFI for this path.  The
 * boundary it exposes is the Python operator
 *   FlowUpsamplerNet.forward(gt, rrdbResults, z, epses, logdet, reverse, ...)   (interflow/models/modules/FlowUpsamplerNet.py:202-215)
 * called from InterpFlow.forward (interflow/models/flow_model.py:83-116).  The entry points below
 * are what a ctypes binding of that operator calls; each one cites the reference code it
 * replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer except `layers` is DEVICE memory owned by the caller; nothing is allocated
 *     or freed inside the library; all work is enqueued on `stream` and nothing synchronises.
 *   - tensors at the boundary are contiguous NCHW float32, exactly as the reference passes them.
 *   - every function returns 0 on success, a negative IFL_E* code on a rejected argument, or a
 *     positive hipError_t; ifl_last_error() gives the text for the calling thread.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - threads: entry points may be called from several host threads.  The flow drivers (ifl_flow_encode/decode/pass) share
 *     one set of library-owned streams and events per device; a per-device mutex serialises their ENQUEUE sequences, so calls
 *     from different threads or on different caller streams are each correct (their bulk phases queue one behind the other on
 *     the shared auxiliary stream).  Two calls must not use the same workspace at the same time.  ifl_set_schedule,
 *     ifl_set_conv1_algo and ifl_prof_enable change process-wide settings: call them while no other thread is inside a driver.
 */
#ifndef INTERFLOW_HIP_H
#define INTERFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFL_VERSION 1

#define IFL_EINVAL   (-1)   /* bad shape / argument (the reference would raise AssertionError) */
#define IFL_ENOTSUP  (-2)   /* configuration outside the shipped model (C not in {12,48}, ...)   */
#define IFL_ESMALLWS (-3)   /* workspace smaller than ifl_ws_bytes() says                        */

#define IFL_HIDDEN   64     /* coupling-network width, FlowAffineCouplingsAblation.py:133        */
#define IFL_FT_CH    320    /* conditioning channels, FlowUpsamplerNet.py:117-120                */
#define IFL_MAX_LEVELS 2    /* flow levels (squeezes), flow_model.py:42 L=2                      */

/* One entry of FlowUpsamplerNet.layers (FlowUpsamplerNet.py:94-105). */
typedef struct ifl_layer {
    int32_t kind;        /* 0 = SqueezeLayer, 1 = FlowStep without coupling, 2 = conditional FlowStep */
    int32_t C;           /* channels of the tensor this layer outputs (12 or 48)                     */
    int64_t param_off;   /* offset, in floats, of this layer's block inside `params`                 */
    float   sum_logs;    /* ActNorm: sum_c logs[c]        (FlowActNorms.py:72)                       */
    float   sum_log_s;   /* InvertibleConv1x1: sum_c log_s[c] (Permutations.py:94)                   */
} ifl_layer;

int         ifl_version(void);
const char* ifl_last_error(void);

/* ---- packed parameter block of one FlowStep (layout documented in DESIGN.md section "params") ----
 * ifl_step_param_floats: size in floats of the block for a step with C channels.
 * ifl_step_field_off:    float offset of a field inside the block, field ids IFL_F_*.          */
enum {
    IFL_F_ANIC_FWD = 0,  /* [b(C)] [exp(logs)(C)] [W(CxC)]          ActNorm then 1x1 conv           */
    IFL_F_ANIC_INV = 1,  /* [b(C)] [exp(-logs)(C)] [Winv(CxC)]      inverse 1x1 conv then ActNorm   */
    IFL_F_W1A = 2, IFL_F_W1F = 3,    /* first 3x3 conv, MFMA A-fragment order                       */
    IFL_F_AN1A = 4, IFL_F_AN1F = 5,  /* its ActNorm (bias, exp(logs)) in accumulator order          */
    IFL_F_W2A = 6, IFL_F_W2F = 7,    /* 1x1 conv                                                    */
    IFL_F_AN2A = 8, IFL_F_AN2F = 9,
    IFL_F_W3A = 10, IFL_F_W3F = 11,  /* Conv2dZeros 3x3, rows grouped as (shift,shift,scale,scale)  */
    IFL_F_B3A = 12, IFL_F_B3F = 13,  /* its bias and exp(3*logs) per accumulator element            */
    IFL_F_W1A_BF16 = 14, IFL_F_W1F_BF16 = 15,   /* bf16 copies of the conditioning part of W1A / W1F (2 per float slot)  */
    IFL_F_W2F_BF16 = 16,                        /* bf16 copy of fFeatures' 1x1 conv, accumulator-as-operand k order       */
    IFL_F_W1A_WINO = 17, IFL_F_W1F_WINO = 18,   /* Winograd F(2x2,3x3) transform G g G^T of the conditioning part of W1A / W1F:
                                                   [column jc of the 4x4 domain][chunk][k-step][row i][m][MFMA A fragment]  */
    IFL_F_W1A_WINO4 = 19, IFL_F_W1F_WINO4 = 20, /* Winograd F(4x4,3x3) transform (6x6 domain) of the same weights:
                                                   [half of the outputs][3x3 block of positions][sub-chunk of 8 inputs][position][MFMA A fragment] */
    IFL_F_W3A_WINO = 21, IFL_F_W3F_WINO = 22,   /* Winograd F(2x2,3x3) transform of the Conv2dZeros weights (rows permuted as W3A / W3F):
                                                   [row tile][sub-chunk of 16 inputs][position][input quad q][row][4]                        */
    IFL_F_COUNT = 23
};
int64_t ifl_step_param_floats(int C, int has_coupling);
int64_t ifl_step_field_off(int C, int field);

/* ---- workspace ----
 * One workspace serves every call with the same (B, H, W).  It holds the zero-ringed
 * channels-last copies of the conditioning features, the running z of each level, the hidden
 * activations of the coupling networks and the per-tile log-det partial sums.
 * ifl_ws_init must run once after allocation (and again if the memory was reused for another
 * shape): it zeroes the padding rings the kernels rely on.  ifl_ws_bytes returns the size in
 * bytes, or a negative IFL_E* code for a shape the library rejects.                            */
int64_t ifl_ws_bytes(int B, int H, int W);
int     ifl_ws_init(void* ws, int64_t ws_bytes, int B, int H, int W, void* stream);

/* Copy the conditioning feature map of one flow level (level 1: H/2 x W/2, level 2: H/4 x W/4;
 * NCHW [B,320,h,w]) into the workspace.  Replaces the per-layer
 * `rrdbResults[self.levelToName[level]]` lookups of FlowUpsamplerNet.py:236-239,278-280; the
 * caller applies the reference's level rule and passes the tensor it selects.                   */
int ifl_load_cond(void* ws, int B, int H, int W, int level, const float* ft_nchw, void* stream);

/* FlowUpsamplerNet.encode (FlowUpsamplerNet.py:217-255): x [B,3,H,W] -> z [B,48,H/4,W/4],
 * logdet_out[b] = logdet_in[b] + sum of the layers' log-determinants.  H, W multiples of 4
 * (flow.py:172).  eps = CondAffineSeparatedAndCond 'eps' (flow_model.py:46).                    */
int ifl_flow_encode(const ifl_layer* layers, int n_layers, const float* params, void* ws,
                    const float* x, float* z, const float* logdet_in, float* logdet_out,
                    int B, int H, int W, float eps, void* stream);

/* FlowUpsamplerNet.decode (FlowUpsamplerNet.py:271-304): z [B,48,H/4,W/4] -> x [B,3,H,W].      */
int ifl_flow_decode(const ifl_layer* layers, int n_layers, const float* params, void* ws,
                    const float* z, float* x, const float* logdet_in, float* logdet_out,
                    int B, int H, int W, float eps, void* stream);

/* Conditioning reuse.  Both coupling networks of a step start with a 3x3 conv over the 320 conditioning channels
 * (FlowAffineCouplingsAblation.py:219-233); that part -- 93 % of the path's FLOPs -- depends only on rrdbResults and the
 * weights, not on the tensor flowing through the stack.  The reference recomputes it in every pass; its own callers run
 * several passes over one conditioning (weighted_interpolate_image, flow_model.py:194-200: two encodes and one decode;
 * generate_images.py:60-81: 19 weights per image; train.py:73,85-86,94: three encodes and one decode share lr_enc).
 * So does the whole of fFeatures (its input is the conditioning alone): its shift / scale and the sums of log(scale).
 * ifl_cond_prepare computes both once for every conditional step into the workspace (after ifl_load_cond of both
 * levels; `eps` must be the one the passes use: it is part of fFeatures' scale); ifl_flow_pass with IFL_FLOW_COND_PREPARED then runs only the part of the stack that depends on the running
 * tensor.  The prepared state lives in `ws` and stays valid until ifl_load_cond, ifl_flow_encode/decode (which prepare
 * on their own), ifl_flow_step or ifl_ws_init touch that workspace, or the parameters change.  Results are identical,
 * bit for bit, to ifl_flow_encode / ifl_flow_decode (same kernels, same order of operations).
 * ifl_flow_pass(decode = 0/1, flags = 0) is ifl_flow_encode / ifl_flow_decode.                                      */
#define IFL_FLOW_COND_PREPARED 1
/* IFL_FLOW_BF16_CONV1 (BASELINE.json configs[4]: "bf16 coupling-NN convs with fp32 ActNorm/logdet"): the convolutions of the
 * coupling networks that depend only on the conditioning features -- the conditioning part of both networks' first conv and
 * fFeatures' 1x1 conv (direct form, v_mfma_f32_32x32x16_bf16) and fFeatures' last conv (Winograd F(2x2,3x3) with the
 * transformed operands rounded, v_mfma_f32_16x16x16_bf16): 95 % of the path's FLOPs -- run on the bf16 matrix cores (operands
 * rounded to bf16, fp32 accumulation).  They are the same function in both directions, so encode and decode still invert each
 * other to fp32 accuracy.  ActNorm, ReLU, sigmoid, the couplings, the invertible 1x1 convs and the log-dets stay fp32.
 * IFL_FLOW_BF16_CHAIN (only together with IFL_FLOW_BF16_CONV1) also runs fAffine's z-dependent convolutions (z1 taps, 1x1,
 * last conv) with bf16 operands.  That is the whole of configs[4]'s arithmetic, at a price no implementation can avoid: the z1
 * that the decode feeds to fAffine equals the encode's only to fp32 rounding, the bf16 rounding of it turns some of those
 * last-bit differences into 2^-8 relative ones, and decode(encode(x)) then matches x to bf16 accuracy only (measured: 8e-3 on
 * the benign fixture, 9e-2 in the wide-scale regime, against <= 1e-4 without this flag).
 * Pass the flags to ifl_cond_prepare and to EVERY ifl_flow_pass of that conditioning (with or without
 * IFL_FLOW_COND_PREPARED).  Against the fp32 reference the outputs differ by the bf16 rounding (tolerances:
 * tests/test_gpu_parity.py).                                                                                              */
#define IFL_FLOW_BF16_CONV1 2
#define IFL_FLOW_BF16_CHAIN 4
/* IFL_FLOW_COND_EARLY + ifl_load_cond_early: passes that follow each other on UNCHANGED conditioning features (no counterpart in
 * the reference, whose passes are ordinary stream-ordered module calls).  A pass's bulk convolutions read the features and the
 * weights only, yet the caller's stream orders them behind the whole previous pass, so the chip runs the tail of that pass's
 * per-step chain (about a tenth of a pass) without them.  ifl_load_cond_early is ifl_load_cond enqueued on the library's bulk
 * stream, behind what that stream already holds and behind `after_event` (a hipEvent_t, or NULL), instead of on a stream of the
 * caller; ifl_flow_pass with IFL_FLOW_COND_EARLY then starts its bulk work behind those loads without waiting for the calling
 * stream (the chain still runs on the calling stream, behind everything it holds).  The caller promises that
 *   - `ft_nchw` was complete, on the stream the previous pass (ifl_flow_pass / encode / decode) on this device was issued to,
 *     when that pass was issued -- e.g. that pass loaded the same buffer -- and nothing has written it since (the loads wait
 *     for that point of that stream): the library cannot see who produces the features, which is why this is a flag;
 *   - `after_event` covers the last work that used `ws` (pass, ifl_cond_prepare, ifl_flow_step): the loads overwrite its
 *     conditioning and the bulk work its slabs.  In practice: two workspaces used in turn, an event recorded behind each pass;
 *   - both levels are loaded with ifl_load_cond_early before the pass, and the pass is the next library call that uses `ws`.
 * Results are identical, bit for bit, to the ordinary order.  With the one-stream schedule (ifl_set_schedule aux_stream = 0)
 * the pass waits for the early loads and nothing overlaps.                                                                */
#define IFL_FLOW_COND_EARLY 8
int ifl_load_cond_early(void* ws, int B, int H, int W, int level, const float* ft_nchw, void* after_event);
int ifl_cond_prepare(const ifl_layer* layers, int n_layers, const float* params, void* ws, int B, int H, int W,
                     float eps, int flags, void* stream);
int ifl_flow_pass(const ifl_layer* layers, int n_layers, const float* params, void* ws, const float* src, float* dst,
                  const float* logdet_in, float* logdet_out, int B, int H, int W, float eps, int decode, int flags,
                  void* stream);

/* How ifl_flow_encode / ifl_flow_decode spread their work over streams (no counterpart in the reference).
 *   aux_stream      1: the z-independent bulk convolutions run on a library-owned stream (high priority; IFL_AUX_PRIO_HIGH=0: low), forked from and
 *                   joined to the caller's stream with events, so the per-step chain overlaps them; 0: everything is
 *                   enqueued on the caller's stream in order.
 *   groups          1..4 contiguous sample groups whose chains run on separate library-owned streams.
 *   p1_chunk_steps  conditional steps per bulk launch (1..16), all levels.
 * A negative / zero value keeps the current setting.  Defaults: 1, 2, 4 at the first level and 8 below (environment IFL_AUX_STREAM, IFL_GROUPS,
 * IFL_P1_CHUNK_STEPS and, for the levels below the first, IFL_P1_CHUNK_STEPS_L2 override them at load; a pass started with
 * IFL_FLOW_COND_EARLY launches 8 first-level steps at a time, IFL_P1_CHUNK_STEPS_EARLY).  Results do not depend on the schedule.  The library-owned streams and
 * events are created once per device at the first call that needs them.                                            */
int ifl_set_schedule(int aux_stream, int groups, int p1_chunk_steps);

/* Algorithm of the z-independent bulk convolution in fp32 (no counterpart in the reference): 2 = Winograd F(4x4,3x3)
 * minimal filtering (default; 4x fewer multiplications than the direct form, same fp32 arithmetic type, sums associated
 * differently), 1 = Winograd F(2x2,3x3) (2.25x fewer), 0 = direct implicit GEMM.  Environment IFL_CONV1_WINO=0/1/2 sets
 * the default at load.  Any other value only queries.
 * Returns the previous setting.  The bf16 variant (IFL_FLOW_BF16_CONV1) always uses the direct form.                */
int ifl_set_conv1_algo(int algo);

/* The same choice for ONE flow level (1 = the first, 80x80 planes of a 160x160 patch; 2 = every deeper level); ifl_set_conv1_algo
 * sets all levels.  Why per level: the rounding noise of the F(4x4) form only shows in a decoded image when it enters at the
 * LAST level -- the decode runs that level first and the remaining steps amplify it; at the first level it adds nothing
 * measurable (profiles/r03_notes.md).  Environment IFL_CONV1_WINO_L2=0/1/2 sets the deeper levels' default at load.
 * Returns the level's previous setting; any other `algo` only queries.  No counterpart in the reference.             */
int ifl_set_conv1_algo_level(int level, int algo);

/* The LAST `nsteps` conditional steps of the LAST flow level are evaluated by `algo` (same codes) whatever the level's setting
 * is: default 4 steps by F(2x2,3x3).  The decode runs those steps first and every later step amplifies their rounding noise;
 * 62 % of the noise variance the F(4x4) form adds to a decoded image enters there (86 % in the last 8; profiles/r03_notes.md),
 * for 2.7 % of a pass.  nsteps = 0 switches it off; a negative nsteps / an algo outside 0..2 leaves that value as it is.
 * Environment IFL_CONV1_TAIL_STEPS / IFL_CONV1_TAIL_ALGO set the defaults at load.  Returns the previous number of steps.
 * Like the other ifl_set_* calls: process-wide, not between ifl_cond_prepare and the passes that use its results
 * (an encode and the decode that inverts it must see the same setting).  No counterpart in the reference.              */
int ifl_set_conv1_tail(int nsteps, int algo);

/* Algorithm of the coupling networks' last convolution (flow.Conv2dZeros 3x3, flow.py:67-82) in fp32 (no counterpart in the
 * reference): 1 = Winograd F(2x2,3x3) minimal filtering (default; 2.25x fewer multiplications, transforms are additions only),
 * 0 = direct implicit GEMM.  Environment IFL_CONV3_WINO=0/1 sets the default at load.  Any other value only queries.  Returns
 * the previous setting.  Do not change it between ifl_cond_prepare and the passes that use the prepared workspace (the two
 * forms lay out fFeatures' log-det partial sums differently).                                                            */
int ifl_set_conv3_algo(int algo);

/* Encode chain: a conditional step's ActNorm2d + InvertibleConv1x1 (FlowActNorms.py:87-92, Permutations.py:114-118) applied
 * inside the kernel that finishes fAffine's first two layers (two launches per step instead of three, the mapped latent written
 * to a second buffer of the workspace; same arithmetic, bit-identical results; no counterpart in the reference).  Bit mask:
 * bit 0 = steps with C = 12 (encode only), bit 1 = steps with C = 48 (encode, and decode: the previous step's inverse pair plus
 * this step's z = (z + shiftFt) * scaleFt, FlowAffineCouplingsAblation.py:170-175); default 2, 0 = separate kernels everywhere.  Environment
 * IFL_FUSE_ANIC=0..3 sets the default at load.  Any other value only queries.  Returns the previous setting.            */
int ifl_set_chain_fusion(int on);

/* ---- stand-alone operators on NCHW tensors ---- */
/* squeeze2d / unsqueeze2d, flow.py:163-193 (pure index permutation, bit-exact). */
int ifl_squeeze2d(const float* in, float* out, int B, int C, int H, int W, int factor, void* stream);
int ifl_unsqueeze2d(const float* in, float* out, int B, int C, int H, int W, int factor, void* stream);

/* One FlowStep on NCHW tensors (FlowStep.py:114-196): z_in [B,C,h,w] -> z_out, logdet updated.
 * ft [B,320,h,w] is read only when the step has a coupling.  Uses `ws` sized by
 * ifl_ws_bytes(B, 2h, 2w) for C = 12 or ifl_ws_bytes(B, 4h, 4w) for C = 48 (it runs the step on
 * the workspace's level-1 / level-2 buffers).                                                    */
int ifl_flow_step(const ifl_layer* layer, const float* params, void* ws, const float* z_in,
                  float* z_out, const float* ft, const float* logdet_in, float* logdet_out,
                  int B, int h, int w, int reverse, float eps, void* stream);

/* Gaussian.logp, flow.py:138-161: out[b] = sum_{h,w} -1/2 (logdet_cov + k ln 2pi + (x-mean)^T cov_inv (x-mean)).
 * cov_inv = NULL means the standard normal (mean ignored, logdet_cov = 0).  k = C <= 64.
 * scratch: B*ifl_logp_scratch_floats(H*W) floats.                                               */
int64_t ifl_logp_scratch_floats(int HW);
int ifl_gaussian_logp(const float* z, const float* mean, const float* cov_inv, float logdet_cov,
                      float* scratch, float* out, int B, int C, int H, int W, void* stream);

/* ifl_gaussian_logp for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels and the same results, bit for bit, as
 * ifl_gaussian_logp; the per-block partial sums live in a static device array inside the library (one copy per GPU,
 * IFL_LOGP_STATIC_MAX_FLOATS floats = 256 KiB, held until the process ends) instead of a caller-owned scratch.  The call
 * allocates and frees nothing.  Limit: B * ifl_logp_scratch_floats(H*W) <= IFL_LOGP_STATIC_MAX_FLOATS, otherwise
 * IFL_ENOTSUP (use ifl_gaussian_logp).  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two
 * streams, two host threads) race; calls on one stream, or otherwise ordered, are safe (each call overwrites every slot it
 * later reads, so nothing carries over from the call before).  The FIRST call on a device looks the array's address up
 * (hipGetSymbolAddress, which may load the code object): make one warm-up call per device outside graph capture.
 * Alignment: z 4 bytes (16 selects the vector form), mean and cov_inv 4 bytes.                                        */
#define IFL_LOGP_STATIC_MAX_FLOATS 65536
int ifl_gaussian_logp_static(const float* z, const float* mean, const float* cov_inv, float logdet_cov, float* out,
                             int B, int C, int H, int W, void* stream);

/* Latent interpolation z = wa[b]*za + wb[b]*zb (flow_model.py:186-200, train.py:88-92).
 * wa/wb are per-sample weights in device memory, n_per_sample = C*H*W.                          */
int ifl_lerp(const float* za, const float* zb, const float* wa, const float* wb, float* out,
             int B, int64_t n_per_sample, void* stream);

/* ---- generic convolution primitives (the building blocks of the encoder and of the training path) ----
 * Tensors are "ringed channels-last" planes [B][h+2][w+2][stride] float32 whose one-pixel ring is zero (the 'same' padding
 * of flow.py:27 / nn.Conv2d(padding=1)); a source or destination is a channel range [off, off+n) of such a buffer.
 * ifl_conv2d: out = epilogue(conv(cat(src0, src1), w) + bias), stride 1, 3x3 (taps 9) or 1x1 (taps 1).  Channel counts are
 * multiples of 32 (pad with zero channels / zero weights).  w is in fragment order: element
 *   [group of 32 outputs][chunk of 32 inputs][tap][kq][half h][i][e] = W[out = group*32 + i][in = chunk*32 + kq*8 + h*4 + e][tap].
 * Epilogue: act (0 none, 1 LeakyReLU 0.2, 2 ReLU), then v = v*alpha + res1,
 * then v = v*beta + res2 (each optional).  Only interior pixels are written.
 * The data gradient of a convolution is ifl_conv2d with the transposed, spatially flipped weights.
 * ifl_conv2d_wgrad: dW[co][ci0*32 + ci][tap] += sum_{b,px} dy[b,px][co] * x[b,px+tap][ci] for the cin channels of one
 * source (dW row-major [cout][cin_total][taps], float atomics: zero it first; cout in {32,64,96}).                       */
typedef struct ifl_conv_desc {
    const float* src[2]; int32_t src_stride[2], src_off[2], src_ch[2]; int32_t nsrc;
    const float* w; const float* bias;
    float* out; int32_t out_stride, out_off, cout, taps, act;
    const float* res1; int32_t res1_stride, res1_off; float alpha;
    const float* res2; int32_t res2_stride, res2_off; float beta;
} ifl_conv_desc;
int ifl_conv2d(const ifl_conv_desc* d, int B, int h, int w, void* stream);
/* nz convolutions of one shape in one launch (the same layer of nz flow steps): problem k uses src[i] + k*src_i_z, w + k*w_z,
 * bias + k*bias_z, res1 + k*res1_z and writes out + k*out_z (strides in floats; B * nz <= 65535; no res2).              */
int ifl_conv2d_batched(const ifl_conv_desc* d, int nz, int64_t src0_z, int64_t src1_z, int64_t w_z, int64_t bias_z,
                       int64_t out_z, int64_t res1_z, int B, int h, int w, void* stream);
int ifl_conv2d_wgrad(const float* x, int x_stride, int x_off, int cin, const float* dy, int dy_stride, int dy_off, int cout,
                     float* dW, int cin_total, int ci0, int taps, int B, int h, int w, void* stream);
/* nz independent weight gradients of one shape in one launch (the same convolution of nz flow steps): problem k reads
 * x + k*x_zstride, dy + k*dy_zstride and adds into dW + k*dw_zstride.                                                    */
int ifl_conv2d_wgrad_batched(const float* x, int x_stride, int x_off, int cin, int64_t x_zstride, const float* dy, int dy_stride,
                             int dy_off, int cout, int64_t dy_zstride, float* dW, int cin_total, int ci0, int64_t dw_zstride,
                             int nz, int taps, int B, int h, int w, void* stream);

/* ---- per-pixel operators of the coupling networks for the training path, forward and backward (train_glue.hip) ----
 * ifl_an_relu_*: y = relu((x + bias) * exp(logs)) on ringed channels-last planes [B][h+2][w+2][64] (flow.Conv2d's ActNorm,
 * flow.py:60-64 / FlowActNorms.py:87-92, and the ReLU behind it, FlowAffineCouplingsAblation.py:250,252); only interior pixels
 * are read and written.  The backward writes dx and per-block partial sums partials[ifl_an_relu_blocks()][128]: [0:64] of
 * d bias, [64:128] of d logs (the caller adds the blocks).
 * ifl_coupling_*: the affine coupling of FlowAffineCouplingsAblation.py:167-207 on channels [c0, c0+n) of a contiguous
 * channels-last z [B][h][w][C] (other channels copied): hs = h * exp(3 logs3) (flow.py:80-82), shift = hs[2i], scale =
 * sigmoid(hs[2i+1] + 2) + eps (:229), h = the raw Conv2dZeros output (conv + bias) in a ringed buffer of `hstride` channels.
 * branch 0: z' = (z + shift) * scale, logdet += sum log scale; branch 1: z' = z / scale - shift, logdet -= sum log scale.
 * ldpart[B][ifl_coupling_blocks()] receives the signed per-block sums.  The backward takes d z', d logdet [B] (or NULL) and
 * writes d z, d h (interior pixels, rows [0, 2n); the buffer's other entries must be zero) and
 * dlogs_part[B * ifl_coupling_blocks()][2n], the per-block sums of d logs3.                                              */
int64_t ifl_an_relu_blocks(int B, int h, int w);
int ifl_an_relu_fwd(const float* x, float* y, const float* bias, const float* logs, int B, int h, int w, int C, void* stream);
int ifl_an_relu_bwd(const float* x, const float* dy, float* dx, const float* bias, const float* logs, float* partials,
                    int B, int h, int w, int C, void* stream);
int ifl_coupling_blocks(int h, int w);
int ifl_coupling_fwd(const float* z, float* zout, const float* h_ring, int hstride, const float* logs3, float eps, int branch,
                     int c0, int n, float* ldpart, int B, int h, int w, int C, void* stream);
int ifl_coupling_bwd(const float* z, const float* dzout, const float* dld, const float* h_ring, int hstride, const float* logs3,
                     float eps, int branch, int c0, int n, float* dz, float* dh_ring, float* dlogs_part, int B, int h, int w,
                     int C, void* stream);

/* ---- gradient-enabled pass of one flow level (training: train.py:73,129 through FlowUpsamplerNet.encode) ----
 * Replaces, for the encode direction, what torch autograd does over FlowStep.normal_flow (FlowStep.py:114-151) in the
 * reference: ifl_train_level_fwd runs the level's steps and keeps the activations in the caller's buffers,
 * ifl_train_level_bwd walks them in reverse.  All buffers are caller-owned device memory; the step array itself is host
 * memory.  "ringed" = [B][h+2][w+2][channels] with a zero one-pixel ring that the kernels never write.
 *
 * ifl_pack_table: one launch over a device-resident table of re-layout jobs (dst[0..nelem) is written):
 *   IFL_PACK_FWD         W [cout_real][real0+real1][taps] -> fragment order of ifl_conv2d with cout_pad outputs and pad0+pad1
 *                        inputs (source 0: real0 of pad0 channels, source 1: real1 of pad1)
 *   IFL_PACK_TRANSPOSED  the data-gradient weights Wt[a][co][tap] = W[co][a][taps-1-tap] for the padded input channels
 *                        a0 + [0, acnt), fragment order with acnt outputs and cout_pad inputs
 *   IFL_PACK_VECTOR      dst[i] = i < cout_real ? src[i] : 0, i < nelem (padded bias)
 *   IFL_PACK_GRAD        dense gradient [cout_pad][pad0+pad1][taps] (ifl_conv2d_wgrad) -> the parameter's
 *                        [cout_real][real0+real1][taps]
 *   IFL_PACK_WINO4       input channels [real0, real0 + 320) of a 3x3 weight [64][cin_src][3][3] -> U = G g G^T in the 6x6
 *                        domain of Winograd F(4x4,3x3) (float64 arithmetic, rounded once), in the fragment order of the
 *                        bulk kernel (conv1x.hip): 64*320*36 floats
 *   IFL_PACK_WINO4_T     the same input channels, for the data gradient wrt the conditioning features (the 'same'
 *                        convolution of the output gradient with the 180-degree rotated taps): U' = G rot180(g) G^T written as
 *                        dst[(p*320 + ci)*acnt + a0 + co], p = 0..35, co = 0..63 (a0 = the (step, network) pair's first column,
 *                        acnt = the row length = 64 * pairs); nelem = 36*320*64
 * cin_src: the parameter's input-channel count (row length of src) when it differs from real0 + real1 (a weight of which
 * only some input channels are packed); 0 = real0 + real1.                                                           */
enum { IFL_PACK_FWD = 0, IFL_PACK_TRANSPOSED = 1, IFL_PACK_VECTOR = 2, IFL_PACK_GRAD = 3, IFL_PACK_WINO4 = 4, IFL_PACK_WINO4_T = 5 };
typedef struct ifl_pack_desc {
    const float* src; float* dst; int64_t nelem;
    int32_t mode, cout_real, cout_pad, taps, real0, pad0, real1, pad1, a0, acnt, cin_src, pad_;
} ifl_pack_desc;
int ifl_pack_table(const ifl_pack_desc* table_dev, int n, int64_t max_elems, void* stream);

/* One coupling network (fAffine or fFeatures, FlowAffineCouplingsAblation.py:248-256) of one step and its coupling. */
typedef struct ifl_train_net {
    const float *w1, *w2, *w3;                   /* packed forward weights: 3x3 (z1|ft -> 64), 1x1, Conv2dZeros 3x3 (64 -> hstride) */
    const float *w1t, *w2t, *w3t;                /* packed data-gradient weights (w1t: towards z1 only, NULL without z1)       */
    const float *an1_bias, *an1_logs, *an2_bias, *an2_logs;   /* the two flow.Conv2d ActNorms, 64 floats each (parameters)      */
    const float *bias3, *logs3;                  /* Conv2dZeros bias padded to hstride floats; its logs [2n] (parameter)        */
    float *p1, *a1, *p2, *a2;                    /* saved: conv outputs and ActNorm+ReLU outputs, ringed, 64 channels            */
    float *hraw;                                 /* saved: raw Conv2dZeros output, ringed, hstride channels                      */
    float *dW1, *dW2, *dW3;                      /* dense weight gradients [64][(32+)ft_ch][9], [64][64][1], [hstride][64][9]:
                                                    accumulated with atomics, zero them before the backward                    */
    float *an1_part, *an2_part;                  /* [ifl_an_relu_blocks][128] partial sums (d bias | d logs)                     */
    float *dlogs_part, *dbias_part;              /* [B * ifl_coupling_blocks][2n] partial sums of d logs3 and of d bias3         */
    float *ldpart;                               /* [B][ifl_coupling_blocks] signed per-block sums of log scale                  */
    float *dp1;                                  /* optional: the gradient wrt p1 is kept here (ringed, 64 channels, zero ring)
                                                    instead of a scratch buffer, for a batched weight gradient afterwards        */
    float *dp2, *dh;                             /* optional, likewise: the gradients wrt p2 (64 channels) and wrt hraw (hstride
                                                    channels, zero outside the interior's rows [0, 2n))                          */
    int32_t c0, n, hstride, has_z1;              /* coupling on channels [c0, c0+n); has_z1: the network also reads z[:, :ch]    */
    int32_t p1_stride, pad_;                     /* pixel stride of p1 (64; 128 when p1 is a channel range of the bulk slab)     */
} ifl_train_net;

/* One FlowStep: z_a = M z_in + c (ActNorm2d then InvertibleConv1x1, M = W diag(exp(logs)), c = M bias), then -- cond != 0 --
 * coupling net[0] (z_a -> z_mid) and coupling net[1] (z_mid -> z_out), in the order the reference applies them.        */
typedef struct ifl_train_step {
    const float *M, *c;                          /* [C][C], [C]                                                                  */
    float *dM_part;                              /* [ifl_affine_px_blocks(B h w)][C*C + C] partial sums of dM (row-major) | dc   */
    float *z_in, *z_a, *z_mid, *z_out;           /* contiguous channels-last [B][h][w][C]; z_in is read, the others written      */
    float *z1;                                   /* optional: this step's z1 plane (ringed, 32 channels, zero outside the interior's
                                                    first ch channels), kept for the backward instead of the level's z1ring      */
    int32_t cond, pad_;
    ifl_train_net net[2];
} ifl_train_step;

typedef struct ifl_train_level {
    int32_t B, h, w, C, ch, nsteps, branch, ft_ch;   /* ch = C/2 channels feed the networks; branch as ifl_coupling_fwd         */
    int32_t skip_step_wgrad;                     /* != 0: the backward leaves ALL weight gradients of the coupling networks to the
                                                    caller (batched launches over the saved dp1 / dp2 / dh / z1 afterwards)       */
    int32_t groups;                              /* 2: samples [0, B/2) and [B/2, B) run as two concurrent chains (the caller's
                                                    stream and a library-owned one); per-block partial sums (an*_part, dM_part)
                                                    then hold group 0's blocks followed by group 1's; else one chain              */
    float eps;  int32_t skip_ft_wgrad;           /* != 0: the backward leaves the first convolutions' weight gradient wrt ft
                                                    to the caller (ifl_wino4_kmajor + GEMM + ifl_wino4_dw over the saved dp1)     */
    const float* ft;                             /* conditioning features, ringed, ft_ch channels (no gradient towards them)     */
    float *z1ring, *dz1ring;                     /* scratch, ringed 32 channels, zero outside the interior's first ch channels   */
    float *dz[4];                                /* scratch, [B][h][w][C]                                                        */
    float *dh[2];                                /* scratch per network slot, ringed hstride channels, zero outside rows [0,2n)  */
    float *d64[2];                               /* scratch, ringed 64 channels, zero ring                                       */
    const ifl_train_step* steps;                 /* host array [nsteps]                                                          */
    /* z-independent network (has_z1 == 0) batched over the steps -- needs raw, skip_step_wgrad and per-step dp1 / dp2 / dh:
     * its ActNorm+ReLU, 1x1 and last convolution run as ONE launch each for all conditional steps before the chain
     * (forward) and after it (backward: data gradients back to dp1), so the sequential chain only applies its coupling.  */
    const float* anp;                            /* that network's ActNorm parameters per conditional step, anp_step floats
                                                    apart: [an1 bias | an1 logs | an2 bias | an2 logs] x 64                   */
    int64_t anp_step;
    float* da_all;                               /* scratch [conditional steps][B][h+2][w+2][64], zero rings                  */
    int32_t f_batched, decode;                   /* f_batched = 2: as 1, but the z-independent network's forward results (raw,
                                                    a1, p2, a2, hraw of that network) are in the buffers already -- a pass over the same
                                                    features with the same weights computed them -- and the forward call skips it.
                                                    decode: the steps are listed in the order FlowUpsamplerNet.decode runs them and each
                                                    runs fFeatures' coupling, fAffine's, then the (inverse) affine map (FlowStep.py:153-196) */
    const float* raw;                            /* optional: ifl_train_bulk_conv1's slabs [conditional step][B][h+2][w+2][128]; then
                                                    w1 of a network with z1 holds the 32 z1 input channels only and p1 = its z1
                                                    taps + raw[.., 0:64]; a network without z1 has p1 = raw + 64, p1_stride 128  */
    int64_t raw_step;                            /* floats between the slabs of consecutive conditional steps                    */
} ifl_train_level;
/* The conditioning part (320 channels of ft) of BOTH networks' first 3x3 convolution for all conditional steps of a level
 * by the Winograd F(4x4,3x3) bulk kernel (the same kernel the inference path uses): raw[step][..][0:64] = fAffine's sums,
 * [64:128] = fFeatures'.  wx: per step two IFL_PACK_WINO4 blocks (fAffine, fFeatures), wx_step floats apart, plus 4096
 * floats of slack behind the last one.  vg: ifl_train_vg_floats(B, h, w) floats of scratch.                              */
/* Weight gradient of those convolutions in the same Winograd domain (36 multiplications per 4x4 outputs instead of 144):
 *   dU_p[row][ci] = sum_tiles dM_p[row][tile] * V_p[ci][tile]  for each of the 36 positions p,   dg = G^T dU G
 * with V = B^T d B of the 6x6 windows of ft and dM = A dY A^T of the 4x4 blocks of the saved gradients (row = pair*64 +
 * output channel).  ifl_wino4_kmajor writes either transform K-major, out[p][row0 + c][tile] (K = ifl_wino4_tiles() tiles
 * per row; mode 0: src = ringed ft with C channels; mode 1: src = ringed gradient planes of C = 64 channels, n_planes of
 * them plane_stride floats apart, rows n*64 + c); the 36 GEMMs C_p = dM_p V_p^T are plain batched GEMMs; ifl_wino4_dw
 * maps dU [36][rows][320] back to 3x3 taps and writes them into each pair's parameter-layout gradient,
 * dst[pair][(co*cin[pair] + zoff[pair] + ci)*9 + tap], from a device table of ifl_wino4_dst records.                     */
typedef struct ifl_wino4_dst { float* dst; int32_t cin, zoff; } ifl_wino4_dst;
int64_t ifl_wino4_tiles(int B, int h, int w);
int ifl_wino4_kmajor(int mode, const float* src, int C, int n_planes, int64_t plane_stride, float* out, int64_t rows_total,
                     int64_t row0, int B, int h, int w, void* stream);
int ifl_wino4_dw(const float* dU, int64_t rows_total, const ifl_wino4_dst* table_dev, int n_pairs, void* stream);
/* The gradient wrt the conditioning features (train.py:129 when the encoder is trained: RRDBGlowNet_arch.py:26-33) of all
 * first convolutions of a level at once, in the Winograd domain: X = ifl_wino4_kmajor(mode 0) of the output-gradient planes
 * [36][64*pairs][tiles], Y_p = U'_p X_p (36 plain GEMMs, U' from IFL_PACK_WINO4_T), and ifl_wino4_untile maps
 * Y [36][C][tiles] back to pixels: out NCHW [B][C][h][w] = A^T Y A (every element written).                              */
int ifl_wino4_untile(const float* Y, float* out_nchw, int C, int B, int h, int w, void* stream);
int64_t ifl_train_vg_floats(int B, int h, int w);
int ifl_train_bulk_conv1(const float* ft_ring, float* vg, const float* wx, int64_t wx_step, int nsteps, float* raw,
                         int64_t raw_step, int B, int h, int w, void* stream);
int64_t ifl_affine_px_blocks(int64_t npx);
int ifl_train_level_fwd(const ifl_train_level* L, void* stream);
/* dz_out: gradient wrt the last step's output; dld: gradient wrt the per-sample log-det [B] (or NULL); dz_in: gradient wrt
 * the first step's input.  Weight gradients, partial sums and dM_part of every step are written as described above.    */
int ifl_train_level_bwd(const ifl_train_level* L, const float* dz_out, const float* dld, float* dz_in, void* stream);

/* ---- conditioning encoder ----
 * RRDBNet.forward(get_steps=True) (RRDBNet_arch.py:85-148) followed by RRDBGlowNet.rrdbPreprocessing
 * (RRDBGlowNet_arch.py:217-246) for the shipped options (scale 4, nf 64, gc 32, stackRRDB concat): x [B,3,H,W] ->
 *   fea_up1 [B,64*(1+n_blocks),H/4,W/4]  (= 'last_lr_fea'),  fea_up2 [..,H/2,W/2],  fea_up4 [..,H,W],  fea_up0 [..,H/8,W/8],
 *   out [B,3,H,W],  blocks_out [n_blocks][B,64,H/4,W/4]  ('block_k' for k in block_idxs)
 * all contiguous NCHW float32; an output pointer may be NULL, and the layers only it needs are skipped.
 * params: ifl_rrdb_param_floats(nb) floats, one record per convolution in execution order (conv_first, RRDB_trunk.k.RDBr.convc,
 * trunk_conv, upconv1, upconv2, HRconv, conv_last); ifl_rrdb_conv_desc gives each record's padded channel counts and the
 * float offsets of its weights (fragment order) and bias.  ws: ifl_rrdb_ws_bytes bytes, ifl_rrdb_ws_init once. */
int     ifl_rrdb_conv_count(int nb);
int64_t ifl_rrdb_param_floats(int nb);
int     ifl_rrdb_conv_desc(int nb, int idx, int* cin, int* cout, int64_t* w_off, int64_t* b_off);
int64_t ifl_rrdb_ws_bytes(int B, int H, int W, int n_blocks);
int     ifl_rrdb_ws_init(void* ws, int64_t ws_bytes, int B, int H, int W, int n_blocks, void* stream);
int     ifl_rrdb_encode(const float* params, int nb, const int* block_idxs, int n_blocks, void* ws, const float* x,
                        float* fea_up1, float* fea_up2, float* fea_up4, float* fea_up0, float* out, float* blocks_out,
                        int B, int H, int W, void* stream);

/* The encoder with gradients (train.py:72,129 with network_G.train_RRDB; RRDBGlowNet_arch.py:26-33 makes its parameters
 * trainable): ifl_rrdb_train_fwd is ifl_rrdb_encode for 'fea_up1' / 'fea_up2' (fea_up2 may be NULL) that keeps every dense
 * block's buffer in `ws` (ifl_rrdb_train_ws_bytes bytes, ifl_rrdb_train_ws_init once: ~100 MB per 160x160 image at nb = 23);
 * ifl_rrdb_train_bwd takes the gradients wrt those two outputs (NCHW, either may be NULL) and writes the gradient of every
 * convolution it ran into d_params: [cout_pad][cin_pad][3][3] at w_off and the bias gradient at b_off of
 * ifl_rrdb_conv_desc (zero d_params first).  params: as ifl_rrdb_encode; params_t: at the same offsets, each weight
 * transposed and flipped for the data gradient (IFL_PACK_TRANSPOSED with a0 = 0, acnt = cin_pad).  One backward per forward. */
int64_t ifl_rrdb_train_ws_bytes(int B, int H, int W, int nb);
int     ifl_rrdb_train_ws_init(void* ws, int64_t ws_bytes, int B, int H, int W, int nb, void* stream);
int     ifl_rrdb_train_fwd(const float* params, int nb, const int* block_idxs, int n_blocks, void* ws, const float* x,
                           float* fea_up1, float* fea_up2, int B, int H, int W, void* stream);
int     ifl_rrdb_train_bwd(const float* params_t, int nb, const int* block_idxs, int n_blocks, void* ws, const float* d_up1,
                           const float* d_up2, float* d_params, int B, int H, int W, void* stream);

/* Separable gather-resize: the antialiased cubic down-scaling of the LR-consistency loss, train.py:102-103 =
 * utils/resize_right.py:32-123 resize(x, 1/4, cubic, antialiasing=True).  x [N][Hin][Win] -> out [N][Hout][Wout], dimension -2
 * first, then -1, as the reference; per output index a field of view of T taps: idx*[out][T] (int32 input indices, clamped for
 * taps on the zero padding) and w*[out][T] (their normalised weights, 0 on the padding), built by the caller with the
 * reference's formulas (interflow_amd/resize.py).  The adjoint (backward pass) is the same call with the transposed tables.   */
int ifl_resize_sep(const float* x, float* out, int N, int Hin, int Win, int Hout, int Wout, const int* idxH, const float* wH,
                   int TH, const int* idxW, const float* wW, int TW, void* stream);

/* Image quantisation of the generation loop, generate_images.py:89-91:
 * out[b,h,w,c] = (uint8)(clip(x[b,c,h,w], 0, 1) * 255)   (NCHW float32 -> NHWC uint8, bit-exact with numpy).  C in {1,3,4}. */
int ifl_quantize_u8_nhwc(const float* x, uint8_t* out, int B, int C, int H, int W, void* stream);

/* ---- opt-in kernel timing (no counterpart in the reference; measurement aid for bench.py) ----
 * ifl_prof_enable(mask): bit i of `mask` turns timing on for kernel id i (an event pair costs a few
 * microseconds of stream time per launch, so time only what you report).  While enabled, the flow
 * drivers bracket each launch of those kernels with a pair of HIP events recorded on the caller's stream (events are created on demand: the "no allocation" rule above holds
 * only while timing is off).  ifl_prof_read synchronises on those events and returns the summed
 * duration and the number of launches of one kernel id; ifl_prof_enable(0) frees the events.       */
enum {
    IFL_K_CONV1_FT_L1 = 0,    /* conv1_ft_kernel (conditioning part of both networks), level 1 (C = 12) */
    IFL_K_CONV1_FT_L2 = 1,    /* same, level 2 (C = 48) */
    IFL_K_CONV1_FINISH = 2,   /* conv1_finish_kernel (fAffine's z1 taps + epilogue) */
    IFL_K_CONV3_F = 3,        /* conv3f_kernel: fFeatures' last conv + shift/scale for the steps of a bulk launch */
    IFL_K_CONV3_ENC = 4,
    IFL_K_CONV3_DEC = 5,
    IFL_K_ANIC = 6
};
int ifl_prof_enable(int mask);
int ifl_prof_read(int kernel_id, double* total_ms, int64_t* count);

/* ---- the optimizer update of the training step (train.py:53-56,130: torch.optim.Adam) as ONE launch over a device table ----
 * Row = a chunk of at most IFL_ADAM_CHUNK elements of one parameter tensor with its gradient and moments (the caller splits
 * tensors into chunks).  t = *step_in + 1; m += (g - m)(1 - beta1); v = beta2 v + (1 - beta2) g^2;
 * p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps); *step_out = t.  *skip != 0 (device flag, may be NULL):
 * nothing is written but *step_out = *step_in (the reference's loss guards train.py:113-126 decided on the device).
 * step_in and step_out are two different device cells the caller alternates.  interflow_amd/training.py:TableAdam.      */
enum { IFL_ADAM_CHUNK = 4096 };
typedef struct ifl_adam_row { float* p; const float* g; float* m; float* v; int64_t n; } ifl_adam_row;
int ifl_adam_table(const ifl_adam_row* table_dev, int n_rows, float lr, float beta1, float beta2, float eps,
                   const float* step_in, float* step_out, const float* skip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INTERFLOW_HIP_H */
