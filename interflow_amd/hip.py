"""ctypes binding of libinterflow_hip.so (C ABI: include/interflow_hip.h).

PyTorch is used only for device memory and streams: every call passes raw device pointers and the
current HIP stream.  There is no CPU or PyTorch fallback: if the shared library is missing or a
tensor is not on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IFL_LIB_PATH") or os.path.join(_HERE, "libinterflow_hip.so")     # IFL_LIB_PATH: experiment builds (tools/build_exp.sh)

IFL_EINVAL, IFL_ENOTSUP, IFL_ESMALLWS = -1, -2, -3
FLOW_COND_PREPARED = 1
FLOW_BF16_CONV1 = 2
FLOW_BF16_CHAIN = 4
FLOW_COND_EARLY = 8

# field ids of the packed per-step parameter block (interflow_hip.h IFL_F_*)
(F_ANIC_FWD, F_ANIC_INV, F_W1A, F_W1F, F_AN1A, F_AN1F, F_W2A, F_W2F, F_AN2A, F_AN2F,
 F_W3A, F_W3F, F_B3A, F_B3F, F_W1A_BF16, F_W1F_BF16, F_W2F_BF16, F_W1A_WINO, F_W1F_WINO,
 F_W1A_WINO4, F_W1F_WINO4, F_W3A_WINO, F_W3F_WINO) = range(23)


class IflLayer(Structure):
    _fields_ = [("kind", c_int32), ("C", c_int32), ("param_off", c_int64),
                ("sum_logs", c_float), ("sum_log_s", c_float)]


class IflConvDesc(Structure):
    _fields_ = [("src", c_void_p * 2), ("src_stride", c_int32 * 2), ("src_off", c_int32 * 2), ("src_ch", c_int32 * 2),
                ("nsrc", c_int32), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p), ("out_stride", c_int32),
                ("out_off", c_int32), ("cout", c_int32), ("taps", c_int32), ("act", c_int32),
                ("res1", c_void_p), ("res1_stride", c_int32), ("res1_off", c_int32), ("alpha", c_float),
                ("res2", c_void_p), ("res2_stride", c_int32), ("res2_off", c_int32), ("beta", c_float)]


class InterflowHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the library once; fail loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise InterflowHipError(
            f"{LIB_PATH} not found: build it with `make -C interflow_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    L = ctypes.CDLL(LIB_PATH)
    L.ifl_version.restype = c_int
    L.ifl_last_error.restype = c_char_p
    L.ifl_step_param_floats.restype = c_int64
    L.ifl_step_param_floats.argtypes = [c_int, c_int]
    L.ifl_step_field_off.restype = c_int64
    L.ifl_step_field_off.argtypes = [c_int, c_int]
    L.ifl_ws_bytes.restype = c_int64
    L.ifl_ws_bytes.argtypes = [c_int, c_int, c_int]
    L.ifl_ws_init.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_void_p]
    L.ifl_load_cond.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.ifl_load_cond_early.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.ifl_load_cond_early.restype = c_int
    flow_args = [POINTER(IflLayer), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                 c_int, c_int, c_int, c_float, c_void_p]
    L.ifl_flow_encode.argtypes = flow_args
    L.ifl_flow_decode.argtypes = flow_args
    L.ifl_cond_prepare.argtypes = [POINTER(IflLayer), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]
    L.ifl_flow_pass.argtypes = flow_args[:-1] + [c_int, c_int, c_void_p]
    L.ifl_flow_step.argtypes = [POINTER(IflLayer), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]
    sq = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_squeeze2d.argtypes = sq
    L.ifl_unsqueeze2d.argtypes = sq
    L.ifl_logp_scratch_floats.restype = c_int64
    L.ifl_logp_scratch_floats.argtypes = [c_int]
    L.ifl_gaussian_logp.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_gaussian_logp_static.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                           c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_gaussian_logp_static.restype = c_int
    L.ifl_lerp.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]
    L.ifl_rrdb_conv_count.argtypes = [c_int]
    L.ifl_rrdb_conv_count.restype = c_int
    L.ifl_rrdb_param_floats.argtypes = [c_int]
    L.ifl_rrdb_param_floats.restype = c_int64
    L.ifl_rrdb_conv_desc.argtypes = [c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int64), POINTER(c_int64)]
    L.ifl_rrdb_conv_desc.restype = c_int
    L.ifl_rrdb_ws_bytes.argtypes = [c_int, c_int, c_int, c_int]
    L.ifl_rrdb_ws_bytes.restype = c_int64
    L.ifl_rrdb_ws_init.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_rrdb_ws_init.restype = c_int
    L.ifl_rrdb_encode.argtypes = [c_void_p, c_int, POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    L.ifl_rrdb_encode.restype = c_int
    L.ifl_conv2d.argtypes = [POINTER(IflConvDesc), c_int, c_int, c_int, c_void_p]
    L.ifl_conv2d.restype = c_int
    L.ifl_conv2d_wgrad.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p]
    L.ifl_conv2d_wgrad.restype = c_int
    L.ifl_quantize_u8_nhwc.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_quantize_u8_nhwc.restype = c_int
    L.ifl_set_schedule.argtypes = [c_int, c_int, c_int]
    L.ifl_set_schedule.restype = c_int
    L.ifl_set_conv1_algo.argtypes = [c_int]
    L.ifl_set_conv1_algo.restype = c_int
    L.ifl_set_conv1_algo_level.argtypes = [c_int, c_int]
    L.ifl_set_conv1_algo_level.restype = c_int
    L.ifl_set_conv1_tail.argtypes = [c_int, c_int]
    L.ifl_set_conv1_tail.restype = c_int
    L.ifl_set_conv3_algo.argtypes = [c_int]
    L.ifl_set_conv3_algo.restype = c_int
    L.ifl_set_chain_fusion.argtypes = [c_int]
    L.ifl_set_chain_fusion.restype = c_int
    L.ifl_an_relu_blocks.argtypes = [c_int, c_int, c_int]
    L.ifl_an_relu_blocks.restype = c_int64
    L.ifl_an_relu_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_an_relu_fwd.restype = c_int
    L.ifl_an_relu_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_an_relu_bwd.restype = c_int
    L.ifl_coupling_blocks.argtypes = [c_int, c_int]
    L.ifl_coupling_blocks.restype = c_int
    L.ifl_coupling_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_int, c_int, c_int, c_void_p,
                                   c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_coupling_fwd.restype = c_int
    L.ifl_coupling_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.ifl_coupling_bwd.restype = c_int
    L.ifl_resize_sep.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int, c_void_p]
    L.ifl_resize_sep.restype = c_int
    L.ifl_adam_table.argtypes = [c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]
    L.ifl_adam_table.restype = c_int
    L.ifl_prof_enable.argtypes = [c_int]
    L.ifl_prof_enable.restype = c_int
    L.ifl_prof_read.argtypes = [c_int, POINTER(ctypes.c_double), POINTER(c_int64)]
    L.ifl_prof_read.restype = c_int
    for name in ("ifl_ws_init", "ifl_load_cond", "ifl_flow_encode", "ifl_flow_decode", "ifl_flow_step",
                 "ifl_cond_prepare", "ifl_flow_pass",
                 "ifl_squeeze2d", "ifl_unsqueeze2d", "ifl_gaussian_logp", "ifl_lerp"):
        getattr(L, name).restype = c_int
    _lib = L
    return L


ADAM_CHUNK = 4096             # IFL_ADAM_CHUNK of the header
EXPORTS = ("ifl_version", "ifl_last_error", "ifl_step_param_floats", "ifl_step_field_off", "ifl_ws_bytes",
           "ifl_ws_init", "ifl_load_cond", "ifl_flow_encode", "ifl_flow_decode", "ifl_flow_step",
           "ifl_squeeze2d", "ifl_unsqueeze2d", "ifl_logp_scratch_floats", "ifl_gaussian_logp", "ifl_lerp",
           "ifl_prof_enable", "ifl_prof_read", "ifl_set_schedule", "ifl_set_conv1_algo", "ifl_set_conv1_algo_level", "ifl_set_conv1_tail", "ifl_cond_prepare", "ifl_flow_pass", "ifl_load_cond_early",
           "ifl_quantize_u8_nhwc", "ifl_rrdb_conv_count", "ifl_rrdb_param_floats", "ifl_rrdb_conv_desc", "ifl_rrdb_ws_bytes",
           "ifl_rrdb_ws_init", "ifl_rrdb_encode", "ifl_conv2d", "ifl_conv2d_wgrad", "ifl_gaussian_logp_static", "ifl_set_conv3_algo", "ifl_set_chain_fusion",
           "ifl_an_relu_blocks", "ifl_an_relu_fwd", "ifl_an_relu_bwd", "ifl_coupling_blocks", "ifl_coupling_fwd", "ifl_coupling_bwd", "ifl_resize_sep",
           "ifl_adam_table", "ifl_wino4_untile", "ifl_rrdb_train_ws_bytes", "ifl_rrdb_train_ws_init", "ifl_rrdb_train_fwd", "ifl_rrdb_train_bwd", "ifl_pack_table", "ifl_affine_px_blocks", "ifl_train_level_fwd", "ifl_train_level_bwd", "ifl_train_vg_floats",
           "ifl_train_bulk_conv1", "ifl_wino4_tiles", "ifl_wino4_kmajor", "ifl_wino4_dw", "ifl_conv2d_wgrad_batched", "ifl_conv2d_batched")

K_CONV1_FT_L1, K_CONV1_FT_L2, K_CONV1_FINISH, K_CONV3_F, K_CONV3_ENC, K_CONV3_DEC, K_ANIC = range(7)


def set_schedule(aux_stream: int = -1, groups: int = -1, p1_chunk_steps: int = -1):
    """Stream plan of the flow drivers (see ifl_set_schedule in interflow_hip.h); -1 keeps a setting."""
    check(lib().ifl_set_schedule(aux_stream, groups, p1_chunk_steps), "ifl_set_schedule")


_cross_pass = [os.environ.get("IFL_CROSS_PASS_OVERLAP", "1") != "0"]


def set_cross_pass_overlap(on=None) -> bool:
    """Passes of FlowUpsamplerNet that follow each other on the SAME conditioning tensors (same objects, not written in
    between: torch's version counters) start their bulk convolutions behind the previous pass's bulk convolutions instead of
    behind its whole per-step chain (IFL_FLOW_COND_EARLY in interflow_hip.h; two workspaces per shape used in turn).  Default
    on (environment IFL_CROSS_PASS_OVERLAP=0: off); `None` only queries.  Results are bit-identical either way.  Turn it off
    if conditioning tensors are written behind torch's back (a kernel of your own on `.data_ptr()`): the version counter is
    all this layer can see.  Returns the previous setting."""
    prev = _cross_pass[0]
    if on is not None:
        _cross_pass[0] = bool(on)
    return prev


CONV1_DIRECT, CONV1_WINOGRAD, CONV1_WINOGRAD4 = 0, 1, 2


def set_conv1_tail(nsteps: int = -1, algo: int = -1) -> int:
    """ifl_set_conv1_tail: the last `nsteps` conditional steps of the last level by `algo` (default 4 by F(2x2)); returns the
    previous number of steps."""
    return int(lib().ifl_set_conv1_tail(nsteps, algo))


def set_conv1_algo_level(level: int, algo: int = -1) -> int:
    """ifl_set_conv1_algo_level: the bulk convolution's algorithm at one flow level (1 = first, 2 = deeper); returns the previous one."""
    return int(lib().ifl_set_conv1_algo_level(level, algo))


def set_conv1_algo(algo: int = -1) -> int:
    """Algorithm of the fp32 bulk convolution (ifl_set_conv1_algo): CONV1_WINOGRAD4 (F(4x4,3x3), default), CONV1_WINOGRAD
    (F(2x2,3x3)) or CONV1_DIRECT; any other value only queries.  Returns the previous setting."""
    return int(lib().ifl_set_conv1_algo(algo))


CONV3_DIRECT, CONV3_WINOGRAD = 0, 1


def set_chain_fusion(on: int = -1) -> int:
    """Encode chain with a conditional step's ActNorm + 1x1 conv inside its conv1_finish kernel (ifl_set_chain_fusion): bit
    mask, 1 = steps with C = 12, 2 = steps with C = 48 (default 2), 0 = off; any other value only queries.  Returns the previous
    setting.  Results are bit-identical whatever the setting."""
    return int(lib().ifl_set_chain_fusion(on))


def set_conv3_algo(algo: int = -1) -> int:
    """Algorithm of the coupling networks' last convolution (ifl_set_conv3_algo): CONV3_WINOGRAD (F(2x2,3x3), default) or
    CONV3_DIRECT; any other value only queries.  Returns the previous setting."""
    return int(lib().ifl_set_conv3_algo(algo))


def prof_enable(kernel_ids=()):
    """Time the given kernel ids (K_*) with HIP events; an empty list turns timing off and frees the events."""
    mask = 0
    for k in kernel_ids:
        mask |= 1 << k
    check(lib().ifl_prof_enable(mask), "ifl_prof_enable")


def prof_read(kernel_id: int):
    """(total milliseconds, launches) of one kernel id since timing was enabled."""
    ms, n = ctypes.c_double(0), c_int64(0)
    check(lib().ifl_prof_read(kernel_id, ctypes.byref(ms), ctypes.byref(n)), "ifl_prof_read")
    return ms.value, n.value


def check(rc: int, what: str):
    if rc == 0:
        return
    msg = lib().ifl_last_error().decode(errors="replace")
    if rc == IFL_EINVAL:
        # the reference signals these conditions with `assert` (flow.py:172, FlowActNorms.py:104-109, ...)
        raise AssertionError(f"{what}: {msg}")
    if rc == IFL_ENOTSUP:
        raise NotImplementedError(f"{what}: {msg}")
    raise InterflowHipError(f"{what}: rc={rc}: {msg}")


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise InterflowHipError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


class Workspace:
    """Device workspace for one (B, H, W); allocated with torch, initialised by the library."""
    _cache = {}

    def __init__(self, B, H, W, device):
        L = lib()
        nbytes = L.ifl_ws_bytes(B, H + (-H) % 4, W + (-W) % 4)     # FlowStep alone may run on H % 4 == 2 planes
        if nbytes < 0:
            check(int(nbytes), "ifl_ws_bytes")
        self.B, self.H, self.W = B, H, W
        self.nbytes = nbytes
        self.generation = 0          # bumped whenever the conditioning maps / slabs in this workspace are overwritten
        self.done = None             # event behind the last work enqueued on this workspace (mark_used)
        with torch.cuda.device(device):
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            check(L.ifl_ws_init(_ptr(self.buf), nbytes, B, H, W, _stream()), "ifl_ws_init")
            self.mark_used()         # the initialisation is the first work on it (an early load must not overtake it)

    @property
    def ptr(self):
        return _ptr(self.buf)

    def mark_used(self):
        """Record, on the current stream, the end of the work just enqueued on this workspace: what an early conditioning load
        of a later pass (ifl_load_cond_early) waits for before it overwrites the workspace from the library's bulk stream."""
        if self.done is None:
            self.done = torch.cuda.Event()
        self.done.record()

    def done_handle(self):
        return c_void_p(self.done.cuda_event) if self.done is not None else c_void_p(0)

    @classmethod
    def get(cls, B, H, W, device, slot=0):
        key = (B, H, W, torch.device(device).index if torch.device(device).index is not None
               else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, slot)
        ws = cls._cache.get(key)
        if ws is None:
            if len(cls._cache) >= 8:          # a few live shapes (two workspaces each) at most; drop the oldest
                cls._cache.pop(next(iter(cls._cache)))
            ws = cls(B, H, W, device)
            cls._cache[key] = ws
        return ws

    @classmethod
    def clear(cls):
        cls._cache.clear()


# ------------------------------------------------------------------------------------------ operators
def squeeze2d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """flow.py:163-176"""
    assert factor >= 1 and isinstance(factor, int)
    if factor == 1:
        return x
    x = _dev_f32(x, "squeeze2d input")
    B, C, H, W = x.shape
    assert H % factor == 0 and W % factor == 0, "{}".format((H, W, factor))
    out = torch.empty(B, C * factor * factor, H // factor, W // factor, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().ifl_squeeze2d(_ptr(x), _ptr(out), B, C, H, W, factor, _stream()), "ifl_squeeze2d")
    return out


def unsqueeze2d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """flow.py:179-193"""
    assert factor >= 1 and isinstance(factor, int)
    if factor == 1:
        return x
    x = _dev_f32(x, "unsqueeze2d input")
    B, C, H, W = x.shape
    assert C % (factor * factor) == 0, "{}".format(C)
    out = torch.empty(B, C // (factor * factor), H * factor, W * factor, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().ifl_unsqueeze2d(_ptr(x), _ptr(out), B, C, H, W, factor, _stream()), "ifl_unsqueeze2d")
    return out


def gaussian_logp(z: torch.Tensor, mean=None, cov_inv=None, logdet_cov: float = 0.0) -> torch.Tensor:
    """flow.py:158-161 with the inverse covariance and its log-determinant precomputed by the caller."""
    z = _dev_f32(z, "gaussian_logp z")
    B, C, H, W = z.shape
    out = torch.empty(B, dtype=torch.float32, device=z.device)
    if B == 0:
        return out
    L = lib()
    scratch = torch.empty(B * L.ifl_logp_scratch_floats(H * W), dtype=torch.float32, device=z.device)
    if cov_inv is not None:
        mean = _dev_f32(mean, "mean")
        cov_inv = _dev_f32(cov_inv, "cov_inv")
        assert mean.numel() == C and cov_inv.shape == (C, C)
    with torch.cuda.device(z.device):
        check(L.ifl_gaussian_logp(_ptr(z), _ptr(mean) if cov_inv is not None else c_void_p(0), _ptr(cov_inv),
                                  float(logdet_cov), _ptr(scratch), _ptr(out), B, C, H, W, _stream()),
              "ifl_gaussian_logp")
    return out


LOGP_STATIC_MAX_FLOATS = 65536


def gaussian_logp_static(z: torch.Tensor, out: torch.Tensor, mean=None, cov_inv=None, logdet_cov: float = 0.0) -> torch.Tensor:
    """ifl_gaussian_logp_static: gaussian_logp into a caller-owned `out` [B], partial sums in the library's static device
    array -- neither this wrapper nor the library allocates.  Inputs must already be contiguous float32 device tensors."""
    B, C, H, W = z.shape
    for t, name in ((z, "z"), (out, "out"), (mean, "mean"), (cov_inv, "cov_inv")):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise InterflowHipError(f"gaussian_logp_static {name}: needs a contiguous float32 GPU tensor")
    assert out.numel() == B
    with torch.cuda.device(z.device):
        check(lib().ifl_gaussian_logp_static(_ptr(z), _ptr(mean) if cov_inv is not None else c_void_p(0), _ptr(cov_inv),
                                             float(logdet_cov), _ptr(out), B, C, H, W, _stream()),
              "ifl_gaussian_logp_static")
    return out


def lerp(za: torch.Tensor, zb: torch.Tensor, wa, wb) -> torch.Tensor:
    """out = wa*za + wb*zb with scalar or per-sample weights (flow_model.py:190,198; train.py:88-92)."""
    za = _dev_f32(za, "lerp za")
    zb = _dev_f32(zb, "lerp zb")
    assert za.shape == zb.shape
    B = za.shape[0]

    def weights(w):
        if isinstance(w, torch.Tensor):
            w = w.to(device=za.device, dtype=torch.float32).reshape(-1)
            return w.expand(B).contiguous() if w.numel() == 1 else w.contiguous()
        return torch.full((B,), float(w), dtype=torch.float32, device=za.device)
    wa_t, wb_t = weights(wa), weights(wb)
    assert wa_t.numel() == B and wb_t.numel() == B
    out = torch.empty_like(za)
    if B == 0:
        return out
    with torch.cuda.device(za.device):
        check(lib().ifl_lerp(_ptr(za), _ptr(zb), _ptr(wa_t), _ptr(wb_t), _ptr(out), B, za[0].numel(), _stream()),
              "ifl_lerp")
    return out


def quantize_u8_nhwc(x: torch.Tensor) -> torch.Tensor:
    """generate_images.py:89-91: `(np.clip(x, 0, 1) * 255).astype(np.uint8).transpose(0, 2, 3, 1)` on the GPU."""
    x = _dev_f32(x, "quantize_u8_nhwc input")
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, C, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().ifl_quantize_u8_nhwc(_ptr(x), _ptr(out), B, C, H, W, _stream()), "ifl_quantize_u8_nhwc")
    return out
