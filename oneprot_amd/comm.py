"""ctypes binding of liboneprot_comm.so (include/oneprot_comm.h): RCCL communicator behind a C ABI.

The default transport of the hot path is torch.distributed (backend "nccl" = RCCL).  `RcclComm` is the same exchange without torch.distributed:
a host passes plain device pointers and a stream.  `gather_features` uses it when `oneprot_amd.loss.set_feature_comm(comm)` has installed one
(e.g. a host that bootstraps ranks itself); tests drive it directly.  No CPU fallback: a missing library raises."""
import ctypes
import os
from ctypes import c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboneprot_comm.so")
ID_BYTES = 128
F32, BF16 = 0, 1
SUM, AVG = 0, 1
_SIGS = {
    "oneprot_comm_unique_id": (c_int, [c_void_p]),
    "oneprot_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "oneprot_comm_destroy": (c_int, [c_void_p]),
    "oneprot_comm_nranks": (c_int, [c_void_p]),
    "oneprot_comm_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oneprot_comm_reduce_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oneprot_comm_all_reduce": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "oneprot_comm_send_recv": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_size_t, c_int, c_void_p]),
    "oneprot_comm_group_begin": (c_int, []),
    "oneprot_comm_group_end": (c_int, []),
}
_lib = None


class CommLibraryMissing(RuntimeError):
    pass


class CommError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CommLibraryMissing(f"{LIB_PATH} not found: build it with oneprot_amd/csrc/build.sh")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def _dtype(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise CommError(f"unsupported dtype {t.dtype}")


def _check(rc, what):
    if rc != 0:
        raise CommError(f"{what} returned {rc} ({'invalid argument' if rc == -1 else 'RCCL error'})")


def unique_id() -> bytes:
    buf = ctypes.create_string_buffer(ID_BYTES)
    _check(lib().oneprot_comm_unique_id(buf), "oneprot_comm_unique_id")
    return buf.raw


class RcclComm:
    """One RCCL communicator for this process's current GPU.  `uid` = bytes from unique_id() on rank 0, shared by the host."""

    def __init__(self, nranks: int, rank: int, uid: bytes):
        self.nranks, self.rank = nranks, rank
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != ID_BYTES:      # the C side copies ID_BYTES: a short id would be read past its end
            raise CommError(f"unique id must be {ID_BYTES} bytes (oneprot_comm.unique_id() on rank 0), got {len(uid) if hasattr(uid, '__len__') else type(uid)}")
        h = c_void_p()
        _check(lib().oneprot_comm_init(ctypes.byref(h), nranks, rank, ctypes.c_char_p(uid)), "oneprot_comm_init")
        self._h = h

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        send = send.contiguous()
        out = torch.empty((self.nranks,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        _check(lib().oneprot_comm_all_gather(self._h, send.data_ptr(), out.data_ptr(), send.numel(), _dtype(send), self._stream()), "oneprot_comm_all_gather")
        return out

    def reduce_scatter(self, send: torch.Tensor) -> torch.Tensor:
        """send [nranks, ...] -> sum over ranks of send[rank]"""
        send = send.contiguous()
        out = torch.empty(tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        _check(lib().oneprot_comm_reduce_scatter(self._h, send.data_ptr(), out.data_ptr(), out.numel(), _dtype(send), self._stream()), "oneprot_comm_reduce_scatter")
        return out

    def all_reduce_(self, buf: torch.Tensor, average: bool = False) -> torch.Tensor:
        assert buf.is_contiguous()
        _check(lib().oneprot_comm_all_reduce(self._h, buf.data_ptr(), buf.numel(), _dtype(buf), AVG if average else SUM, self._stream()), "oneprot_comm_all_reduce")
        return buf

    def exchange(self, pairs, stream=None):
        """pairs: [(send or None, to_rank, recv or None, from_rank), ...] -- all of them in flight together as one RCCL group on `stream`
        (default: the current stream).  Every tensor of a pair has the same element count and dtype."""
        st = self._stream() if stream is None else stream
        _check(lib().oneprot_comm_group_begin(), "oneprot_comm_group_begin")
        try:
            for send, to, recv, frm in pairs:
                ref = send if send is not None else recv
                assert (send is None or send.is_contiguous()) and (recv is None or recv.is_contiguous())
                _check(lib().oneprot_comm_send_recv(self._h, None if send is None else send.data_ptr(), int(to), None if recv is None else recv.data_ptr(), int(frm),
                                                    ref.numel(), _dtype(ref), st), "oneprot_comm_send_recv")
        finally:
            _check(lib().oneprot_comm_group_end(), "oneprot_comm_group_end")

    def send_recv(self, send, to_rank, recv, from_rank, stream=None):
        self.exchange([(send, to_rank, recv, from_rank)], stream)

    def destroy(self):
        if self._h:
            _check(lib().oneprot_comm_destroy(self._h), "oneprot_comm_destroy")
            self._h = None
