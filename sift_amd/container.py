"""Host-side mirror of sift-persist/src/container.rs over the C ABI.

Same names and error cases as the reference: `Chunk`, `read_header`,
`write_header`, `read_chunk`, `write_chunk`, `Error` (container.rs:4-18).
Readers/writers are byte buffers instead of `impl Read` / `impl Write`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import lib

HEADER = bytes([0x89, 0x53, 0x49, 0x46, 0x54, 0x0D, 0x0A, 0x1A, 0x0A])  # container.rs:104


class Error(Exception):
    """container.rs:4-18.  `.kind` is the reference's variant name."""

    KINDS = {
        _lib.ERR_READ: "Read",
        _lib.ERR_WRITE: "Write",
        _lib.ERR_READ_INVALID_CHUNK_SIZE: "ReadInvalidChunkSize",
        _lib.ERR_WRITE_INVALID_CHUNK_SIZE: "WriteInvalidChunkSize",
        _lib.ERR_READ_CRC_MISMATCH: "ReadCrcMismatch",
    }

    def __init__(self, status: int):
        self.status = status
        self.kind = self.KINDS.get(status, "Device")
        super().__init__(lib.sift_strerror(status).decode())


def _raise(neg: int):
    status = -int(neg)
    if status in Error.KINDS:
        raise Error(status)
    raise _lib.SiftError(status)


def _dev(device):
    return torch.cuda.current_device() if device is None else device


@dataclass
class Chunk:
    """container.rs:35-45."""

    chunk_type: bytes
    data: bytes

    def expect_type(self, expected_type: bytes) -> None:
        """container.rs:86-95."""
        if lib.sift_expect_type(self.chunk_type, expected_type) != _lib.OK:
            raise Error(_lib.ERR_READ)


def write_header(writer: bytearray) -> None:
    """container.rs:124-126."""
    buf = (C.c_uint8 * _lib.HEADER_LEN)()
    n = lib.sift_write_header(buf, _lib.HEADER_LEN)
    if n < 0:
        _raise(n)
    writer += bytes(buf)


def read_header(reader, pos: int = 0) -> int:
    """container.rs:107-121.  Returns the position after the header."""
    a = np.frombuffer(reader, dtype=np.uint8)
    n = lib.sift_read_header(a.ctypes.data + pos if a.size else None, max(a.size - pos, 0))
    if n < 0:
        _raise(n)
    return pos + int(n)


def write_chunk(chunk: Chunk, writer: bytearray, device: int | None = None) -> None:
    """container.rs:68-77 / :134-136."""
    data = np.frombuffer(chunk.data, dtype=np.uint8)
    out = np.empty(data.size + 12, dtype=np.uint8)
    n = lib.sift_write_chunk(_dev(device), bytes(chunk.chunk_type),
                             data.ctypes.data if data.size else None, data.size,
                             out.ctypes.data, out.size)
    if n < 0:
        _raise(n)
    writer += out[:n].tobytes()


def read_chunk(reader, pos: int = 0, device: int | None = None):
    """container.rs:47-66 / :129-131.  Returns (Chunk, position after it)."""
    a = np.frombuffer(reader, dtype=np.uint8)
    t = (C.c_uint8 * 4)()
    off = C.c_uint64(0)
    ln = C.c_uint32(0)
    avail = max(a.size - pos, 0)
    n = lib.sift_read_chunk(_dev(device), a.ctypes.data + pos if a.size else None, avail,
                            t, C.byref(off), C.byref(ln))
    if n < 0:
        _raise(n)
    start = pos + off.value
    return Chunk(bytes(t), a[start:start + ln.value].tobytes()), pos + int(n)


def write_document(payload, writer: bytearray, device: int | None = None) -> None:
    """document.rs:53-60 with `payload` standing for encode_document's bytes:
    header, AMRG chunk, empty SEND chunk."""
    data = np.frombuffer(payload, dtype=np.uint8)
    out = np.empty(data.size + _lib.HEADER_LEN + 24, dtype=np.uint8)
    n = lib.sift_write_document(_dev(device), data.ctypes.data if data.size else None, data.size,
                                out.ctypes.data, out.size)
    if n < 0:
        _raise(n)
    writer += out[:n].tobytes()


def read_document(reader, device: int | None = None, decode=None):
    """document.rs:62-77.  Returns the AMRG payload (or decode(payload) if a decoder
    is given; it runs where the reference runs decode_document, before the SEND
    chunk's verdict is raised)."""
    a = np.frombuffer(reader, dtype=np.uint8)
    off, ln, stage = C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    n = lib.sift_read_document(_dev(device), a.ctypes.data if a.size else None, a.size,
                               C.byref(off), C.byref(ln), C.byref(stage))
    payload = None
    if stage.value >= 3:
        payload = a[off.value:off.value + ln.value].tobytes()
        if decode is not None:
            payload = decode(payload)
    if n < 0:
        _raise(n)
    return payload


def _windows(base: torch.Tensor, offsets, avail):
    """Descriptor pointers for host sequences or int64 device tensors."""
    from .crc import _descriptors
    if avail is None:
        if isinstance(offsets, torch.Tensor) and offsets.is_cuda:
            avail = base.numel() - offsets
        else:
            off = np.ascontiguousarray(offsets, dtype=np.uint64)
            avail = np.uint64(base.numel()) - off
    return _descriptors(base, offsets, avail)


def _batch(fn, base: torch.Tensor, offsets, avail):
    po, pa, n, keep = _windows(base, offsets, avail)
    dev = base.device.index
    status = torch.empty(n, dtype=torch.int32, device=base.device)
    dlen = torch.empty(n, dtype=torch.int32, device=base.device)
    st = fn(dev, base.data_ptr(), po, pa, n, status.data_ptr(),
            dlen.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if st != _lib.OK:
        raise _lib.SiftError(st)
    return status, dlen


def read_documents_device(base: torch.Tensor, offsets, avail=None):
    """The framing of read_document (document.rs:62-77) on n save files resident
    in `base`: header, AMRG chunk, SEND chunk, CRCs verified; the automerge
    decode is not done.  Returns (status, amrg_len) int32 device tensors."""
    return _batch(lib.sift_read_documents_device, base, offsets, avail)


def read_chunks_device(base: torch.Tensor, offsets, avail=None):
    """Chunk::read (container.rs:47-66) on n serialized chunks resident in the
    device tensor `base`, one wave per chunk.  `offsets` is a host sequence of
    chunk starts; `avail[i]` the bytes readable from offsets[i] (default: to the
    end of `base`).  Returns (status, data_len) int32 device tensors; status is
    0 or the reference's error as in include/sift_amd.h.  Asynchronous."""
    return _batch(lib.sift_read_chunks_device, base, offsets, avail)


def write_chunk_crcs_device(base: torch.Tensor, offsets, avail=None) -> torch.Tensor:
    """Chunk::write's CRC step (container.rs:75) for n chunks serialized in place
    in `base` (length | type | data | 4-byte slot): fills every slot with the
    big-endian CRC of type | data.  Returns the int32 status tensor."""
    po, pa, n, keep = _windows(base, offsets, avail)
    dev = base.device.index
    status = torch.empty(n, dtype=torch.int32, device=base.device)
    st = lib.sift_write_chunk_crcs_device(dev, base.data_ptr(), po, pa, n,
                                          status.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if st != _lib.OK:
        raise _lib.SiftError(st)
    return status
