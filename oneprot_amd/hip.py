"""ctypes binding of liboneprot_hip.so (the C ABI in include/oneprot_hip.h).

There is no CPU fallback: if the shared library is missing, or a kernel returns a non-zero status, this module
raises.  torch is used only as the owner of device memory and streams (tensor.data_ptr(), current stream).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ONEPROT_HIP_LIB: another build of the same ABI (development: step-level A/B of a variant library, tools/ab/build_lib.sh)
LIB_PATH = os.environ.get("ONEPROT_HIP_LIB") or os.path.join(_HERE, "liboneprot_hip.so")

ABI_VERSION = 7
EPI_BF16, EPI_F32, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_QKV_ROPE, EPI_GELU_BWD = range(6)
LOG2E = 1.4426950408889634      # the attention kernels take q pre-multiplied by hd^-1/2 * log2(e) (include/oneprot_hip.h)


class HipLibraryMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


P, I, L64, F, SZ = c_void_p, c_int, c_int64, c_float, c_size_t
U64 = ctypes.c_uint64

# name -> (restype, argtypes)    (mirrors include/oneprot_hip.h line by line)
_SIGS = {
    "oneprot_abi_version": (I, []),
    "oneprot_esm_embed_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "oneprot_esm_embed_bwd_workspace": (SZ, [I, I, I]),
    "oneprot_esm_embed_bwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "oneprot_bert_embed_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, F, P]),
    "oneprot_pool_fwd": (I, [P, P, I, P, I, I, I, I, P]),
    "oneprot_pool_bwd": (I, [P, P, I, P, P, I, I, I, I, P]),
    "oneprot_embed_scatter_sorted": (I, [P, P, P, P, L64, I, I, I, P, P]),
    "oneprot_rowsum_f32": (I, [P, P, I, L64, P]),
    "oneprot_attnpool_fwd": (I, [P, P, I, P, P, P, P, I, I, I, P]),
    "oneprot_attnpool_bwd_workspace": (SZ, [I, I]),
    "oneprot_attnpool_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "oneprot_layernorm_fwd": (I, [P, I, P, P, P, P, P, P, L64, I, F, P]),
    "oneprot_layernorm_bwd_workspace": (SZ, [I]),
    "oneprot_layernorm_bwd": (I, [P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, L64, I, I, P]),
    "oneprot_lnpool_fwd": (I, [P, P, I, P, P, P, P, P, P, P, P, I, I, I, F, I, P]),
    "oneprot_gemm_bf16_nt": (I, [P, P, L64, I, I, I, I, I, P, P, P, P, P, P, P, F, I, I, I, P]),
    "oneprot_gemm_ln_pack_weight": (I, [P, P, I, I, P]),
    "oneprot_gemm_bf16_nt_resid_ln": (I, [P, P, L64, I, I, I, P, P, P, P, P, F, P, P, P, P]),
    "oneprot_sched_workspace_bytes": (SZ, [L64]),
    "oneprot_alloc_uncached": (I, [ctypes.POINTER(c_void_p), SZ]),
    "oneprot_free_uncached": (I, [P]),
    "oneprot_sched_workspace_init": (I, [P, SZ, P]),
    "oneprot_dynamic_tiles": (None, [P, SZ]),
    "oneprot_sched_late_draws": (I, [P]),
    "oneprot_sched_epoch": (L64, [P]),
    "oneprot_gemm_bf16_nt_resid_ln8": (I, [P, P, L64, I, I, I, I, P, P, P, P, P, F, P, P, P, SZ, P]),
    "oneprot_gemm_resid_ln8_eligible": (I, [L64, I, I]),
    "oneprot_gemm_resid_ln8_error": (I, [P]),
    "oneprot_gemm_resid_ln8_error_clear": (I, [P, P]),
    "oneprot_gemm_resid_ln8_poll_bound": (None, [I]),
    "oneprot_gemm_ln_form": (None, [I]),
    "oneprot_gemm_ln_form_get": (I, []),
    "oneprot_gemm_force_shape": (None, [I]),
    "oneprot_gemm_tune": (None, [I, I]),
    "oneprot_gemm_bf16_tn_workspace": (SZ, [I, I]),
    "oneprot_gemm_tn_variant": (None, [I]),
    "oneprot_cu_reserve": (None, [I]),
    "oneprot_gemm_bf16_tn": (I, [P, P, L64, I, I, I, I, P, P, P, SZ, I, P]),
    "oneprot_sgemm": (I, [P, P, P, I, I, I, I, I, F, I, P]),
    "oneprot_attn_fwd": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "oneprot_attn_fwd_dropout": (I, [P, P, P, P, P, P, I, I, I, I, F, U64, U64, P]),
    "oneprot_attn_dropout_keep": (I, [P, I, I, I, F, U64, U64, P]),
    "oneprot_attn_bwd_dropout": (I, [P, P, P, P, P, P, P, P, P, F, P, P, I, I, I, I, F, U64, U64, P]),
    "oneprot_attn_bwd_workspace": (SZ, [I, I, I]),
    "oneprot_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, F, P, P, I, I, I, I, P]),
    "oneprot_attn_force_bwd_path": (None, [I]),
    "oneprot_attn_force_fwd_path": (None, [I]),
    "oneprot_gelu_f32": (I, [P, P, L64, P]),
    "oneprot_gelu_bwd_f32": (I, [P, P, P, L64, P]),
    "oneprot_l2norm_fwd": (I, [P, P, P, I, I, F, P]),
    "oneprot_l2norm_bwd": (I, [P, P, P, P, I, I, F, F, P]),
    "oneprot_ce_fwd_bwd": (I, [P, P, P, I, I, I, F, P]),
    "oneprot_siglip_fwd_bwd": (I, [P, P, P, I, F, I, P]),
    "oneprot_siglip_fwd_bwd_dev": (I, [P, P, P, I, P, I, P]),
    "oneprot_diag_rank": (I, [P, P, P, I, P]),
    "oneprot_abs_sum": (I, [P, P, P, L64, F, P]),
    "oneprot_dot_f32": (I, [P, P, P, P, L64, F, P]),
    "oneprot_l1_bwd": (I, [P, P, L64, F, P, I, P]),
    "oneprot_scale_by_device_scalar": (I, [P, L64, P, P]),
    "oneprot_dropout_bf16": (I, [P, P, L64, F, U64, U64, P]),
    "oneprot_dropout_bwd_add_bf16": (I, [P, P, L64, F, U64, U64, P]),
    "oneprot_dropout_bwd_add_f32": (I, [P, P, L64, F, U64, U64, P]),
    "oneprot_dropout_f32": (I, [P, P, L64, F, U64, U64, P]),
    "oneprot_dropout_add_f32": (I, [P, P, P, L64, F, U64, U64, P]),
    "oneprot_dropout_add_layernorm_fwd": (I, [P, P, P, P, P, P, P, P, P, L64, I, F, F, U64, U64, P]),
    "oneprot_key_padding_bias": (I, [P, P, L64, I, P]),
    "oneprot_sumsq_workspace": (SZ, []),
    "oneprot_sumsq": (I, [P, L64, P, P, P]),
    "oneprot_clip_coef": (I, [P, F, P, P, P, P]),
    "oneprot_adam_step": (I, [P, P, P, P, L64, F, F, F, F, F, I, P, P]),
    "oneprot_cast_f32_to_bf16": (I, [P, P, L64, P]),
    "oneprot_transpose_cast_f32_to_bf16": (I, [P, P, I, I, P]),
    "oneprot_transpose_cast_f32_to_bf16_batched": (I, [P, P, I, I, L64, L64, I, P]),
    "oneprot_colsum_workspace": (SZ, [I]),
    "oneprot_colsum_bf16": (I, [P, P, P, L64, I, I, P]),
}

# Expected element type of every pointer argument, in order (f = float32, h = bfloat16, l = int64, i = int32, b = uint8 workspace, * = stated by a
# flag argument / epilogue id).  The C side validates shapes and alignment but cannot see a tensor's dtype, device or strides: a strided view or an
# fp16 tensor would compute garbage silently, so the binding refuses them (HipKernelError) before the launch.
_PTR_DTYPES = {
    "oneprot_esm_embed_fwd": "lfff", "oneprot_esm_embed_bwd": "lfffb", "oneprot_bert_embed_fwd": "lffffffh", "oneprot_pool_fwd": "flf",
    "oneprot_pool_bwd": "flfh", "oneprot_embed_scatter_sorted": "flllf", "oneprot_rowsum_f32": "ff", "oneprot_attnpool_fwd": "flffff",
    "oneprot_attnpool_bwd": "fffffffb", "oneprot_layernorm_fwd": "*ffhfff", "oneprot_layernorm_bwd": "*f*fffffhffb", "oneprot_lnpool_fwd": "flffffffhf",
    "oneprot_gemm_bf16_nt": "hhf**h*ff", "oneprot_gemm_ln_pack_weight": "hh", "oneprot_gemm_bf16_nt_resid_ln": "hhfffffhff", "oneprot_gemm_bf16_nt_resid_ln8": "hhfffffhfb", "oneprot_sched_workspace_init": "b", "oneprot_gemm_resid_ln8_error_clear": "b", "oneprot_gemm_bf16_tn": "hhffb", "oneprot_sgemm": "fff", "oneprot_attn_fwd": "hhhfhf",
    "oneprot_attn_bwd": "hhhfhhfffhb", "oneprot_attn_bwd_dropout": "hhhfhhfffhb", "oneprot_gelu_f32": "ff", "oneprot_gelu_bwd_f32": "fff", "oneprot_l2norm_fwd": "fff", "oneprot_l2norm_bwd": "ffff",
    "oneprot_ce_fwd_bwd": "fff", "oneprot_siglip_fwd_bwd": "fff", "oneprot_siglip_fwd_bwd_dev": "ffff", "oneprot_diag_rank": "fii", "oneprot_abs_sum": "ffb", "oneprot_dot_f32": "fffb", "oneprot_l1_bwd": "fff",
    "oneprot_scale_by_device_scalar": "ff", "oneprot_key_padding_bias": "lf", "oneprot_dropout_bf16": "hh", "oneprot_dropout_bwd_add_bf16": "hh", "oneprot_dropout_bwd_add_f32": "hf", "oneprot_dropout_f32": "ff", "oneprot_dropout_add_f32": "fff", "oneprot_dropout_add_layernorm_fwd": "fffffhfff", "oneprot_attn_fwd_dropout": "hhhfhf", "oneprot_attn_dropout_keep": "b", "oneprot_sumsq": "ffb", "oneprot_clip_coef": "fffb", "oneprot_adam_step": "fffff",
    "oneprot_cast_f32_to_bf16": "fh", "oneprot_transpose_cast_f32_to_bf16": "fh", "oneprot_transpose_cast_f32_to_bf16_batched": "fh", "oneprot_colsum_bf16": "hfb",
}
_DT = {"f": torch.float32, "h": torch.bfloat16, "l": torch.int64, "i": torch.int32, "b": torch.uint8}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """Load (once) and return the ctypes handle.  Raises HipLibraryMissing if the .so was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(oneprot_amd/csrc/build.sh).  There is no CPU fallback for the OneProt hot path.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if h.oneprot_abi_version() != ABI_VERSION:
            raise HipLibraryMissing(f"{LIB_PATH} has ABI version {h.oneprot_abi_version()}, this binding needs {ABI_VERSION}: rebuild it (oneprot_amd/csrc/build.sh)")
        _lib = h
    return _lib


def ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _check_tensors(name, args):
    """device / layout / dtype of every tensor handed to `name` (see _PTR_DTYPES)"""
    kinds = _PTR_DTYPES.get(name)
    slot = 0
    for a in args:
        if not (a is None or isinstance(a, torch.Tensor)):
            continue
        if a is not None:
            if not a.is_cuda:
                raise HipKernelError(f"{name}: pointer argument {slot} is a {a.device} tensor; the HIP path has no CPU fallback")
            if not a.is_contiguous():
                raise HipKernelError(f"{name}: pointer argument {slot} is not contiguous (shape {tuple(a.shape)}, strides {a.stride()}); the kernels take dense row-major memory")
            if kinds is not None and slot < len(kinds) and kinds[slot] != "*" and a.dtype != _DT[kinds[slot]]:
                raise HipKernelError(f"{name}: pointer argument {slot} has dtype {a.dtype}, the entry point takes {_DT[kinds[slot]]}")
        slot += 1


def stream():
    return torch.cuda.current_stream().cuda_stream


_prof = None     # {entry point: (epilogue or None, [(start_event, end_event, scalar args), ...])}


def profile_begin(name, epilogue=None):
    """Bracket every subsequent launch of `name` (optionally: only with this GEMM epilogue id) with HIP events recorded on the
    launch stream; profile_end() returns the per-launch durations in ms.  Used by bench.py for the live roofline figure.
    `name` may be a dict {entry point: epilogue or None} to watch several entry points at once (profile_end then returns a dict of
    [(ms, scalar args), ...] lists)."""
    global _prof
    _prof = {k: (v, []) for k, v in name.items()} if isinstance(name, dict) else {name: (epilogue, [])}
    _prof["__single__"] = None if isinstance(name, dict) else name


def profile_end():
    global _prof
    if _prof is None:
        return []
    torch.cuda.synchronize()
    single = _prof.pop("__single__")
    out = {k: [(a.elapsed_time(b), sc) for a, b, sc in v[1]] for k, v in _prof.items()}
    _prof = None
    return [ms for ms, _ in out[single]] if single is not None else out


def call(name, *args):
    """Invoke an int-returning entry point on the current torch stream; raise on a non-zero status."""
    fn = getattr(lib(), name)
    _check_tensors(name, args)
    cargs = [ptr(a) if isinstance(a, torch.Tensor) or a is None else a for a in args]
    watch = _prof.get(name) if _prof is not None else None
    if watch is not None and (watch[0] is None or args[7] == watch[0]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*cargs, stream())
        e1.record()
        # scalar arguments + the number of tensor arguments (tells e.g. a GEMM launch with the optional second output from one without)
        watch[1].append((e0, e1, tuple(a for a in args if isinstance(a, (int, float))) + (sum(isinstance(a, torch.Tensor) for a in args),)))
    else:
        rc = fn(*cargs, stream())
    if rc != 0:
        raise HipKernelError(f"{name} returned {rc} ({'invalid argument' if rc == -1 else 'launch failure'})")


def query(name, *args):
    return getattr(lib(), name)(*args)


# ---------------------------------------------------------------------------------------------------------------------------------
# The sched workspace (include/oneprot_hip.h, csrc/sched_ws.h): per-device memory through which the persistent kernels hand out work and keep their launch
# bookkeeping on the device.  The host side owns it: one uncached allocation per device, zeroed once, grown when a launch needs more rows.
class _SchedWorkspace:
    def __init__(self, device, rows):
        h = lib()
        self.device, self.rows = device, rows
        self.bytes = h.oneprot_sched_workspace_bytes(rows)
        out = c_void_p()
        with torch.cuda.device(device):
            if h.oneprot_alloc_uncached(ctypes.byref(out), self.bytes) != 0 or not out.value:
                raise HipKernelError(f"oneprot_alloc_uncached({self.bytes} bytes) failed")
            self.ptr = out.value
            if h.oneprot_sched_workspace_init(self.ptr, self.bytes, stream()) != 0:
                raise HipKernelError("oneprot_sched_workspace_init failed")

    def error(self):
        return lib().oneprot_gemm_resid_ln8_error(self.ptr)

    def release(self):
        if self.ptr:
            torch.cuda.synchronize(self.device)
            lib().oneprot_free_uncached(self.ptr)
            self.ptr = 0


_sched = {}      # device index -> _SchedWorkspace
SCHED_MIN_ROWS = 131072


def dynamic_tiles_wanted():
    """ONEPROT_DYNAMIC_TILES=1 / 0; default: on when this process is one of several ranks (WORLD_SIZE > 1), off otherwise.  On = the tiles of the persistent NT GEMMs
    and the slabs of the attention forward are drawn from work queues: a co-resident kernel that holds CUs -- an RCCL channel of the overlapped gradient
    all-reduce -- then costs its share of the chip instead of 1.47 x per launch (profiles/r06_cu_occupier.txt); bit-identical results either way, a tie in
    time on a GPU that runs nothing else."""
    v = os.environ.get("ONEPROT_DYNAMIC_TILES")
    if v is not None:
        return v != "0"
    return int(os.environ.get("WORLD_SIZE", "1")) > 1


def cu_reserve_wanted():
    """ONEPROT_CU_RESERVE=<n>; default 16 when this process is one of several ranks (the overlapped gradient all-reduce's RCCL channels hold CUs), else 0"""
    v = os.environ.get("ONEPROT_CU_RESERVE")
    if v is not None:
        return int(v)
    return 16 if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 0


def sched_workspace(rows=0, device=None):
    """(pointer, bytes) of the current device's sched workspace, sized for at least `rows` rows of row statistics"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    ws = _sched.get(dev)
    if ws is None or ws.rows < rows:
        new_rows = max(rows, SCHED_MIN_ROWS, 2 * ws.rows if ws is not None else 0)
        if ws is not None:
            if ws.error() != 0:
                raise HipKernelError("oneprot_gemm_bf16_nt_resid_ln8: a bounded wait ran out in an earlier launch (NaN rows were written)")
            lib().oneprot_dynamic_tiles(None, 0)
            ws.release()
        ws = _sched[dev] = _SchedWorkspace(dev, new_rows)
        lib().oneprot_dynamic_tiles(ws.ptr if dynamic_tiles_wanted() else None, ws.bytes)
        lib().oneprot_cu_reserve(cu_reserve_wanted())
    return ws.ptr, ws.bytes


def _sched_of(device):
    if not _sched:                                       # nothing has launched through a workspace yet (also: no GPU in this process)
        return None
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    return _sched.get(dev)


def sched_error(device=None):
    """host-synchronous: 1 when a launch on this device's workspace wrote NaN rows because a bounded wait ran out (0 when no workspace exists yet)"""
    ws = _sched_of(device)
    return 0 if ws is None else ws.error()


def sched_late_draws(device=None):
    ws = _sched_of(device)
    return 0 if ws is None else lib().oneprot_sched_late_draws(ws.ptr)


def sched_error_clear(device=None):
    ws = _sched_of(device)
    if ws is not None:
        lib().oneprot_gemm_resid_ln8_error_clear(ws.ptr, stream())


def sched_ptr_or_none(device=None):
    ws = _sched_of(device)
    return None if ws is None else ws.ptr
