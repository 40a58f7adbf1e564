"""sift_amd -- MI355X (gfx950) CRC-32 / container hot path of matta/sift.

Scope: SURVEY.md section 8 only (crc.rs + container.rs of sift-persist).  Hand-written
HIP kernels behind a C ABI (include/sift_amd.h); this package is the thin
Python mirror of the reference's interface used by the tests and bench.
Importing it requires the built library -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError if the .so is missing)
from .crc import Digest, compute, compute_async, compute_batch, concat, crc32, workspace_bytes  # noqa: F401
from . import container  # noqa: F401
