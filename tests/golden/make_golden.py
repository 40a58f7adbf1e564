"""Writes tests/golden/crc_kat.json and crc_random.json.

crc_kat.json: the known-answer vectors held by the reference's own tests
(sift-persist/src/crc.rs:71-96, :121-122, :127-128) -- data, not code.
crc_random.json: seeded random buffers (numpy default_rng(seed), uint8) with
expected running states from Python's stdlib zlib.crc32, an independent
implementation of the same CRC-32/ISO-HDLC; the reference is
Rust and cannot run here.  compute(state, buf) = zlib.crc32(buf, state^~0)^~0.
Run: python tests/golden/make_golden.py
"""
import json
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
M = 0xFFFFFFFF

kat = {
    "source": "sift-persist/src/crc.rs:71-96,121-122,127-128",
    "reference_kat": [
        {"data": "", "crc": "0x00000000"},
        {"data": "The quick brown fox jumps over the lazy dog", "crc": "0x414fa339"},
        {"data": "123456789", "crc": "0xcbf43926"},
        {"data": "various CRC algorithms input data", "crc": "0x9bd366ae"},
        {"data": "Test vector from febooti.com", "crc": "0x0c877f61"},
        {"data": "IEND", "crc": "0xae426082"},
    ],
}
with open(os.path.join(HERE, "crc_kat.json"), "w") as f:
    json.dump(kat, f, indent=1)

cases = []
lens = [0, 1, 3, 4, 15, 16, 17, 64, 1000, 4096, 16383, 16384, 16385, 65537, 1 << 20,
        (4 << 20) + 1, (9 << 20) + 4093]
for i, n in enumerate(lens):
    seed = 0x51F7 + i
    buf = np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()
    for state in (M, 0, 0xDEADBEEF):
        out = zlib.crc32(buf, state ^ M) ^ M
        cases.append({"len": n, "seed": seed, "state": f"0x{state:08x}", "compute": f"0x{out:08x}"})
with open(os.path.join(HERE, "crc_random.json"), "w") as f:
    json.dump({"generator": "numpy.random.default_rng(seed).integers(0,256,len,uint8)",
               "expected_from": "zlib.crc32", "cases": cases}, f, indent=1)
print(len(cases), "cases")

# Container framing cases (sift-persist/src/container.rs has no test of its own,
# SURVEY.md section 4): serialized bytes built here independently of the oracle and
# of the product, with struct.pack + zlib.crc32, from the format as written in
# container.rs:47-77,104: u32 BE length | type | data | u32 BE crc(type|data).
import struct

HEADER = bytes([0x89, 0x53, 0x49, 0x46, 0x54, 0x0D, 0x0A, 0x1A, 0x0A])


def chunk(t: bytes, d: bytes) -> bytes:
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))


rng = np.random.default_rng(0x51F7)
ccases = []
for t, n in [(b"SEND", 0), (b"AMRG", 1), (b"AMRG", 11), (b"AMRG", 12), (b"AMRG", 27), (b"AMRG", 28),
             (b"AMRG", 29), (b"AMRG", 300), (b"AMRG", 4096), (b"tEXt", 20000)]:
    d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    ccases.append({"type": t.decode(), "data": d.hex(), "serialized": chunk(t, d).hex(),
                   "crc": f"0x{zlib.crc32(t + d):08x}"})
doc = HEADER + chunk(b"AMRG", bytes(range(256)) * 3) + chunk(b"SEND", b"")
with open(os.path.join(HERE, "container_cases.json"), "w") as f:
    json.dump({"source": "format of container.rs:47-77,104; crc from zlib.crc32",
               "header": HEADER.hex(), "chunks": ccases,
               "document_framing": {"note": "document.rs:53-60 layout: header, AMRG chunk, empty SEND chunk",
                                    "amrg_data": (bytes(range(256)) * 3).hex(), "file": doc.hex()}},
              f, indent=1)
print(len(ccases), "container cases; empty SEND crc", ccases[0]["crc"])
