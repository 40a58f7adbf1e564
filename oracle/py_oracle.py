"""Pure-Python restatement of the reference CRC and container framing.

CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Small inputs only (pure-Python loops).
Follows /root/reference/sift-persist/src/crc.rs:2-53 and
/root/reference/sift-persist/src/container.rs:20-136 statement by statement.
Pinned by the reference's own known-answer vectors (crc.rs:71-96, :121-122,
:127-128) in tests/test_oracle.py.
"""
from __future__ import annotations

import struct

# crc.rs:2-20 -- CRC_TABLE
CRC_TABLE = []
for _n in range(256):
    _c = _n
    for _ in range(8):
        _c = (0xEDB88320 ^ (_c >> 1)) if (_c & 1) else (_c >> 1)
    CRC_TABLE.append(_c)


def compute(crc: int, buf: bytes) -> int:
    """crc.rs:28-34."""
    c = crc
    for byte in buf:
        c = CRC_TABLE[(c ^ byte) & 0xFF] ^ (c >> 8)
    return c


class Digest:
    """crc.rs:36-53."""

    def __init__(self) -> None:
        self.value = 0xFFFFFFFF

    def update(self, buf: bytes) -> None:
        self.value = compute(self.value, buf)

    def finalize(self) -> int:
        return self.value ^ 0xFFFFFFFF


class ContainerError(Exception):
    """container.rs:4-18 -- kind is the enum variant name."""

    def __init__(self, kind: str, *args):
        super().__init__(kind, *args)
        self.kind = kind


HEADER = bytes([0x89, 0x53, 0x49, 0x46, 0x54, 0x0D, 0x0A, 0x1A, 0x0A])  # container.rs:104
CHUNK_SIZE_LIMIT = 1024 * 1024 * 1024  # container.rs:49


def chunk_crc(chunk_type: bytes, data: bytes) -> int:
    """container.rs:79-84."""
    d = Digest()
    d.update(chunk_type)
    d.update(data)
    return d.finalize()


def write_header() -> bytes:
    return HEADER


def read_header(buf: bytes, pos: int = 0) -> int:
    """container.rs:107-121.  Returns the new position."""
    if len(buf) - pos < len(HEADER):
        raise ContainerError("Read")
    if buf[pos:pos + len(HEADER)] != HEADER:
        raise ContainerError("Read")
    return pos + len(HEADER)


def write_chunk(chunk_type: bytes, data: bytes) -> bytes:
    """container.rs:68-77."""
    if len(data) > 0xFFFFFFFF:
        raise ContainerError("WriteInvalidChunkSize", len(data))
    return struct.pack(">I", len(data)) + chunk_type + data + struct.pack(">I", chunk_crc(chunk_type, data))


def read_chunk(buf: bytes, pos: int = 0):
    """container.rs:47-66.  Returns (chunk_type, data, new_pos)."""
    if len(buf) - pos < 4:
        raise ContainerError("Read")
    (length,) = struct.unpack_from(">I", buf, pos)
    if length > CHUNK_SIZE_LIMIT:
        raise ContainerError("ReadInvalidChunkSize", length, CHUNK_SIZE_LIMIT)
    if len(buf) - pos < 8:
        raise ContainerError("Read")
    chunk_type = buf[pos + 4:pos + 8]
    if len(buf) - pos < 8 + length + 4:
        raise ContainerError("Read")
    data = buf[pos + 8:pos + 8 + length]
    (crc,) = struct.unpack_from(">I", buf, pos + 8 + length)
    if crc != chunk_crc(chunk_type, data):
        raise ContainerError("ReadCrcMismatch")
    return chunk_type, data, pos + 12 + length


AUTOMERGE_CHUNK = b"AMRG"  # document.rs:50
END_CHUNK = b"SEND"        # document.rs:51


def write_document(payload: bytes) -> bytes:
    """document.rs:53-60; `payload` stands for encode_document's bytes."""
    return write_header() + write_chunk(AUTOMERGE_CHUNK, payload) + write_chunk(END_CHUNK, b"")


def read_document(buf: bytes) -> bytes:
    """document.rs:62-77 without decode_document: returns the AMRG payload."""
    pos = read_header(buf)
    chunk_type, data, pos = read_chunk(buf, pos)
    if chunk_type != AUTOMERGE_CHUNK:
        raise ContainerError("Read")
    end_type, _, pos = read_chunk(buf, pos)
    if end_type != END_CHUNK:
        raise ContainerError("Read")
    return data
