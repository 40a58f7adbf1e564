"""ctypes loader for the C oracle (oracle/_build/liboracle.so).

CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sift_oracle.c")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p = C.c_char_p
        L.oracle_crc_compute.restype = C.c_uint32
        L.oracle_crc_compute.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.oracle_crc_repeat.restype = C.c_uint32
        L.oracle_crc_repeat.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_chunk_crc.restype = C.c_uint32
        L.oracle_chunk_crc.argtypes = [u8p, C.c_void_p, C.c_size_t]
        L.oracle_write_header.restype = C.c_int64
        L.oracle_write_header.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_read_header.restype = C.c_int64
        L.oracle_read_header.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_write_chunk.restype = C.c_int64
        L.oracle_write_chunk.argtypes = [u8p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
        L.oracle_read_chunk.restype = C.c_int64
        L.oracle_read_chunk.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.oracle_write_document.restype = C.c_int64
        L.oracle_write_document.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
        L.oracle_read_document.restype = C.c_int64
        L.oracle_read_document.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.oracle_crc_table.restype = C.POINTER(C.c_uint32)
        _lib = L
    return _lib


def _ptr(buf):
    """Address + keepalive for bytes / bytearray / numpy array / memoryview."""
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a)
    return a.ctypes.data if a.size else None, a


def compute(crc: int, buf) -> int:
    """crc.rs:28-34 compute(crc, buf) via the C oracle."""
    p, keep = _ptr(buf)
    return int(lib().oracle_crc_compute(crc & 0xFFFFFFFF, p, keep.size))


def crc32(buf) -> int:
    """Digest::new -> update -> finalize (crc.rs:59-63 test helper)."""
    return compute(0xFFFFFFFF, buf) ^ 0xFFFFFFFF


def chunk_crc(chunk_type: bytes, data) -> int:
    p, keep = _ptr(data)
    return int(lib().oracle_chunk_crc(chunk_type, p, keep.size))


def write_chunk(chunk_type: bytes, data) -> bytes:
    p, keep = _ptr(data)
    out = C.create_string_buffer(keep.size + 12)
    n = lib().oracle_write_chunk(chunk_type, p, keep.size, out, len(out))
    if n < 0:
        raise ValueError(f"oracle_write_chunk error {-n}")
    return out.raw[:n]


def read_chunk(buf, pos: int = 0):
    """Returns (status, chunk_type, data_off, data_len, consumed); status 0 = ok."""
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8)
    t = C.create_string_buffer(4)
    off = C.c_uint64(0)
    ln = C.c_uint32(0)
    base = a.ctypes.data + pos if a.size else None
    n = lib().oracle_read_chunk(base, a.size - pos, t, C.byref(off), C.byref(ln))
    if n < 0:
        return int(-n), None, 0, 0, 0
    return 0, t.raw, pos + off.value, ln.value, int(n)


def write_document(payload) -> bytes:
    """document.rs:53-60 around an opaque payload."""
    p, keep = _ptr(payload)
    out = C.create_string_buffer(keep.size + 9 + 12 + 12)
    n = lib().oracle_write_document(p, keep.size, out, len(out))
    if n < 0:
        raise ValueError(f"oracle_write_document error {-n}")
    return out.raw[:n]


def read_document(buf):
    """document.rs:62-77 framing.  Returns (status, stage, payload_off, payload_len, consumed)."""
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8)
    off, ln, stage = C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    n = lib().oracle_read_document(a.ctypes.data if a.size else None, a.size, C.byref(off), C.byref(ln),
                                   C.byref(stage))
    if n < 0:
        return int(-n), stage.value, off.value, ln.value, 0
    return 0, stage.value, off.value, ln.value, int(n)
