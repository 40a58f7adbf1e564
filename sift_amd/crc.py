"""Host-side mirror of sift-persist/src/crc.rs over the C ABI.

`compute` and `Digest` keep the reference's names, argument meaning and
state convention (crc.rs:28-53).  Inputs may be host bytes (bytes, bytearray,
memoryview, numpy uint8) or a CUDA/HIP torch.uint8 tensor; either way the
bytes are reduced on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _host_view(buf):
    a = buf if isinstance(buf, np.ndarray) else np.frombuffer(buf, dtype=np.uint8)
    if a.dtype != np.uint8:
        a = a.view(np.uint8)
    if not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a)
    return a


def _stream_ptr(device_index: int) -> int:
    return torch.cuda.current_stream(device_index).cuda_stream


def compute(crc: int, buf, device: int | None = None) -> int:
    """crc.rs:28-34 -- compute(crc, buf) -> crc.  Running state in and out."""
    out = C.c_uint32(0)
    if isinstance(buf, torch.Tensor):
        if not buf.is_cuda:
            raise TypeError("tensor input must live on the GPU; pass bytes/numpy for host data")
        if buf.dtype != torch.uint8 or not buf.is_contiguous():
            raise TypeError("tensor input must be contiguous uint8")
        dev = buf.device.index
        check(lib.sift_crc32_compute_device_sync(
            dev, crc & 0xFFFFFFFF, buf.data_ptr(), buf.numel(), C.byref(out), _stream_ptr(dev)))
        return out.value
    a = _host_view(buf)
    dev = torch.cuda.current_device() if device is None else device
    check(lib.sift_crc32_compute_host(
        dev, crc & 0xFFFFFFFF, a.ctypes.data if a.size else None, a.size, C.byref(out)))
    return out.value


def compute_async(crc, buf: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor | None = None,
                  stream: int | None = None) -> None:
    """Asynchronous form on the current stream: result lands in out[0] (int32/uint32
    device tensor).  `crc` is an int or a 1-element device tensor holding the state.
    `stream`: a raw hipStream_t the caller already holds (a loop of calls need not
    ask torch for the current stream every time)."""
    dev = buf.device.index
    state_imm, state_ptr = (0, crc.data_ptr()) if isinstance(crc, torch.Tensor) else (crc & 0xFFFFFFFF, None)
    ws = workspace.data_ptr() if workspace is not None else None
    check(lib.sift_crc32_compute_device(
        dev, state_imm, state_ptr, buf.data_ptr(), buf.numel(), out.data_ptr(), ws,
        _stream_ptr(dev) if stream is None else stream))


def workspace_bytes() -> int:
    return int(lib.sift_crc32_workspace_bytes())


class Digest:
    """crc.rs:36-53 -- Digest::new / update / finalize."""

    def __init__(self, device: int | None = None):
        self._d = _lib.SiftDigest()
        dev = torch.cuda.current_device() if device is None else device
        lib.sift_digest_new(C.byref(self._d), dev)

    @property
    def value(self) -> int:
        return self._d.value

    def update(self, buf) -> None:
        if isinstance(buf, torch.Tensor):
            if not buf.is_cuda or buf.dtype != torch.uint8 or not buf.is_contiguous():
                raise TypeError("tensor input must be a contiguous uint8 GPU tensor")
            self._d.device = buf.device.index
            check(lib.sift_digest_update_device(
                C.byref(self._d), buf.data_ptr(), buf.numel(), _stream_ptr(buf.device.index)))
            return
        a = _host_view(buf)
        check(lib.sift_digest_update(C.byref(self._d), a.ctypes.data if a.size else None, a.size))

    def finalize(self) -> int:
        return int(lib.sift_digest_finalize(C.byref(self._d)))


def crc32(buf) -> int:
    """The reference's test helper (crc.rs:59-63): new -> update -> finalize."""
    d = Digest()
    d.update(buf)
    return d.finalize()


def concat(state_a: int, state_b0: int, len_b: int) -> int:
    """compute(s, A|B) from compute(s, A), compute(0, B) and len(B)."""
    return int(lib.sift_crc32_concat(state_a & 0xFFFFFFFF, state_b0 & 0xFFFFFFFF, len_b))


def _descriptors(base: torch.Tensor, offsets, lens):
    """(offsets_ptr, lens_ptr, n, keepalive) for host sequences or device int64 tensors."""
    if isinstance(offsets, torch.Tensor) and offsets.is_cuda:
        if not (isinstance(lens, torch.Tensor) and lens.is_cuda):
            raise TypeError("offsets and lens must both be host sequences or both device tensors")
        if offsets.dtype != torch.int64 or lens.dtype != torch.int64:
            raise TypeError("device offsets / lens must be int64")
        off, ln = offsets.contiguous(), lens.contiguous()
        assert off.shape == ln.shape and off.ndim == 1
        return off.data_ptr(), ln.data_ptr(), off.numel(), (off, ln)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    assert off.shape == ln.shape and off.ndim == 1
    numel = np.uint64(base.numel())
    if off.size and bool(((ln > numel) | (off > numel - np.minimum(ln, numel))).any()):  # no u64 wrap
        raise ValueError("a buffer reaches past the end of base")
    return off.ctypes.data, ln.ctypes.data, off.size, (off, ln)


def compute_batch(crc: int, base: torch.Tensor, offsets, lens, xor_out: int = 0,
                  out: torch.Tensor | None = None) -> torch.Tensor:
    """compute(crc, buffer i) ^ xor_out for n buffers inside the device tensor
    `base` (crc.rs:28-34 applied n times, one wave per buffer).  `offsets` and
    `lens` are host sequences, or int64 device tensors (used in place; the caller
    vouches that every buffer lies inside `base`).  Returns an int32 device
    tensor (bit pattern of the u32 results); asynchronous on the current stream."""
    po, pl, n, keep = _descriptors(base, offsets, lens)
    dev = base.device.index
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=base.device)
    check(lib.sift_crc32_batch_device(dev, crc & 0xFFFFFFFF, xor_out & 0xFFFFFFFF, base.data_ptr(),
                                      po, pl, n, out.data_ptr(), _stream_ptr(dev)))
    return out
