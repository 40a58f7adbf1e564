"""ctypes binding of libsift_amd.so -- the C ABI in include/sift_amd.h.

There is no fallback: if the HIP library has not been built, importing this
module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C sift_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIFT_AMD_LIB selects another build of the same library (A/B runs of -DSIFT_* variants)
LIB_PATH = os.environ.get("SIFT_AMD_LIB") or os.path.join(_HERE, "libsift_amd.so")

# status codes (include/sift_amd.h)
OK = 0
ERR_READ = 1
ERR_WRITE = 2
ERR_READ_INVALID_CHUNK_SIZE = 3
ERR_WRITE_INVALID_CHUNK_SIZE = 4
ERR_READ_CRC_MISMATCH = 5
ERR_NO_DEVICE = 16
ERR_HIP = 17
ERR_INVALID_ARG = 18
ERR_TOO_MANY_THREADS = 19

TUNE_SINGLE_LAUNCH = 1
TUNE_BATCH_CUT = 2
TUNE_BATCH_WAVE_MAX = 3
TUNE_BATCH_PIECE_LOG2 = 4
TUNE_BATCH_PLAN = 5
TUNE_HOST_COPY_THREADS = 6

HEADER_LEN = 9
CHUNK_SIZE_LIMIT = 1024 * 1024 * 1024
CRC_INIT = 0xFFFFFFFF
STATIC_MAX_LEN = 1 << 38

# every symbol include/sift_amd.h declares; tests check the .so exports them all
EXPORTS = (
    "sift_strerror", "sift_last_error", "sift_init", "sift_shutdown", "sift_reset_workspace", "sift_release_stream",
    "sift_profile_end_samples", "sift_profile_end_kernels",
    "sift_profile_begin", "sift_profile_end", "sift_tune", "sift_crc32_workspace_bytes", "sift_crc32_compute_device",
    "sift_crc32_compute_device_static", "sift_crc32_compute_device_sync", "sift_crc32_compute_host",
    "sift_digest_new", "sift_digest_update", "sift_digest_update_device",
    "sift_digest_finalize", "sift_crc32_concat", "sift_chunk_crc_device", "sift_crc32_batch_device", "sift_read_chunks_device", "sift_read_documents_device", "sift_write_chunk_crcs_device",
    "sift_write_header", "sift_read_header", "sift_write_chunk", "sift_read_chunk",
    "sift_expect_type", "sift_write_document", "sift_read_document",
)


class SiftDigest(C.Structure):
    _fields_ = [("value", C.c_uint32), ("device", C.c_int32)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(run __graft_entry__.build() or `make -C sift_amd/csrc`). "
            "sift_amd has no CPU fallback."
        )
    # torch ships its own libamdhip64.so.7; importing it first makes the loader
    # reuse that runtime for our DT_NEEDED entry, so device pointers and streams
    # made by torch are valid in our calls.
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_int64
    L.sift_strerror.restype = C.c_char_p
    L.sift_strerror.argtypes = [i32]
    L.sift_last_error.restype = C.c_char_p
    L.sift_last_error.argtypes = []
    L.sift_init.restype = i32
    L.sift_init.argtypes = [i32]
    L.sift_shutdown.restype = None
    L.sift_shutdown.argtypes = []
    L.sift_reset_workspace.restype = i32
    L.sift_reset_workspace.argtypes = [i32]
    L.sift_release_stream.restype = i32
    L.sift_release_stream.argtypes = [i32, vp]
    L.sift_profile_end_samples.restype = i32
    L.sift_profile_end_samples.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(C.c_float), i32]
    L.sift_profile_end_kernels.restype = i32
    L.sift_profile_end_kernels.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(C.c_float), i32]
    L.sift_profile_begin.restype = i32
    L.sift_profile_begin.argtypes = [i32, i32, i32]
    L.sift_profile_end.restype = i32
    L.sift_profile_end.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(i32)]
    L.sift_tune.restype = i32
    L.sift_tune.argtypes = [i32, i32]
    L.sift_crc32_workspace_bytes.restype = C.c_size_t
    L.sift_crc32_workspace_bytes.argtypes = []
    L.sift_crc32_compute_device.restype = i32
    L.sift_crc32_compute_device.argtypes = [i32, u32, vp, vp, u64, vp, vp, vp]
    L.sift_crc32_compute_device_static.restype = i32
    L.sift_crc32_compute_device_static.argtypes = [i32, u32, vp, vp, u64, vp, vp]
    L.sift_crc32_compute_device_sync.restype = i32
    L.sift_crc32_compute_device_sync.argtypes = [i32, u32, vp, u64, C.POINTER(u32), vp]
    L.sift_crc32_compute_host.restype = i32
    L.sift_crc32_compute_host.argtypes = [i32, u32, vp, u64, C.POINTER(u32)]
    L.sift_digest_new.restype = None
    L.sift_digest_new.argtypes = [C.POINTER(SiftDigest), i32]
    L.sift_digest_update.restype = i32
    L.sift_digest_update.argtypes = [C.POINTER(SiftDigest), vp, u64]
    L.sift_digest_update_device.restype = i32
    L.sift_digest_update_device.argtypes = [C.POINTER(SiftDigest), vp, u64, vp]
    L.sift_digest_finalize.restype = u32
    L.sift_digest_finalize.argtypes = [C.POINTER(SiftDigest)]
    L.sift_crc32_concat.restype = u32
    L.sift_crc32_concat.argtypes = [u32, u32, u64]
    L.sift_chunk_crc_device.restype = i32
    L.sift_chunk_crc_device.argtypes = [i32, vp, u64, vp, vp, vp]
    L.sift_crc32_batch_device.restype = i32
    L.sift_crc32_batch_device.argtypes = [i32, u32, u32, vp, vp, vp, u32, vp, vp]
    L.sift_read_chunks_device.restype = i32
    L.sift_read_chunks_device.argtypes = [i32, vp, vp, vp, u32, vp, vp, vp]
    L.sift_read_documents_device.restype = i32
    L.sift_read_documents_device.argtypes = [i32, vp, vp, vp, u32, vp, vp, vp]
    L.sift_write_chunk_crcs_device.restype = i32
    L.sift_write_chunk_crcs_device.argtypes = [i32, vp, vp, vp, u32, vp, vp]
    L.sift_write_header.restype = i64
    L.sift_write_header.argtypes = [vp, C.c_size_t]
    L.sift_read_header.restype = i64
    L.sift_read_header.argtypes = [vp, C.c_size_t]
    L.sift_write_chunk.restype = i64
    L.sift_write_chunk.argtypes = [i32, C.c_char_p, vp, u64, vp, C.c_size_t]
    L.sift_read_chunk.restype = i64
    L.sift_read_chunk.argtypes = [i32, vp, C.c_size_t, vp, C.POINTER(u64), C.POINTER(u32)]
    L.sift_write_document.restype = i64
    L.sift_write_document.argtypes = [i32, vp, u64, vp, C.c_size_t]
    L.sift_read_document.restype = i64
    L.sift_read_document.argtypes = [i32, vp, C.c_size_t, C.POINTER(u64), C.POINTER(u32), C.POINTER(i32)]
    L.sift_expect_type.restype = i32
    L.sift_expect_type.argtypes = [C.c_char_p, C.c_char_p]
    return L


lib = _load()


class SiftError(RuntimeError):
    """A non-zero status from the C ABI; `.status` holds the code."""

    def __init__(self, status: int):
        self.status = int(status)
        detail = lib.sift_last_error().decode() if status >= ERR_NO_DEVICE else ""
        msg = lib.sift_strerror(self.status).decode()
        super().__init__(f"{msg}" + (f" ({detail})" if detail else ""))


def check(status: int) -> None:
    if status != OK:
        raise SiftError(status)
