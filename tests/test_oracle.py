"""CPU: pins the oracle (C and pure-Python restatements of crc.rs /
container.rs) against the reference's own known answers, the committed golden
vectors and zlib.crc32."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import py_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
M = 0xFFFFFFFF


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_table_matches_reference_construction(oracle):
    # crc.rs:2-20: entry n is n shifted right 8 times with the reflected poly
    t = oracle.lib().oracle_crc_table()
    assert [t[i] for i in range(256)] == py_oracle.CRC_TABLE
    assert t[0] == 0 and t[1] == 0x77073096 and t[255] == 0x2D02EF8D  # zlib's table ends


def test_reference_known_answers(oracle):
    # crc.rs:71-96 (test_crc32), :111-117, :119-123 (test_check), :125-131 (test_iend)
    for case in _load("crc_kat.json")["reference_kat"]:
        data = case["data"].encode()
        want = int(case["crc"], 16)
        assert oracle.crc32(data) == want
        d = py_oracle.Digest()
        d.update(data)
        assert d.finalize() == want
    assert "0x{:08x}".format(oracle.crc32(b"123456789")) == "0xcbf43926"


def test_golden_random_vectors(oracle):
    for case in _load("crc_random.json")["cases"]:
        buf = np.random.default_rng(case["seed"]).integers(0, 256, size=case["len"], dtype=np.uint8)
        assert oracle.compute(int(case["state"], 16), buf) == int(case["compute"], 16), case


def test_against_zlib_random(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = int(rng.integers(0, 5000))
        buf = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        s = int(rng.integers(0, 1 << 32))
        want = zlib.crc32(buf, s ^ M) ^ M
        assert oracle.compute(s, buf) == want
        if n < 600:
            assert py_oracle.compute(s, buf) == want


def test_digest_streaming_order(oracle):
    # container.rs:81-82: update(type) then update(data) == one pass over both
    t, d = b"AMRG", bytes(range(200))
    assert oracle.chunk_crc(t, d) == zlib.crc32(t + d)
    assert oracle.chunk_crc(b"SEND", b"") == 0x91E6D779  # SURVEY.md section 8(c), derived


def test_container_golden(oracle):
    g = _load("container_cases.json")
    for c in g["chunks"]:
        t, d, ser = c["type"].encode(), bytes.fromhex(c["data"]), bytes.fromhex(c["serialized"])
        assert oracle.write_chunk(t, d) == ser
        assert py_oracle.write_chunk(t, d) == ser
        st, typ, off, ln, used = oracle.read_chunk(ser)
        assert (st, typ, ln, used) == (0, t, len(d), len(ser)) and ser[off:off + ln] == d
        assert py_oracle.read_chunk(ser) == (t, d, len(ser))
    doc = bytes.fromhex(g["document_framing"]["file"])
    pos = py_oracle.read_header(doc)
    t, d, pos = py_oracle.read_chunk(doc, pos)
    assert t == b"AMRG" and d.hex() == g["document_framing"]["amrg_data"]
    t, d, pos = py_oracle.read_chunk(doc, pos)
    assert (t, d, pos) == (b"SEND", b"", len(doc))


def test_container_errors(oracle):
    L = oracle.lib()
    good = oracle.write_chunk(b"AMRG", b"hello world")
    # CRC mismatch: flip one data bit (container.rs:60-63)
    bad = bytearray(good)
    bad[9] ^= 1
    assert oracle.read_chunk(bytes(bad))[0] == 5
    # stored CRC altered
    bad = bytearray(good)
    bad[-1] ^= 0x80
    assert oracle.read_chunk(bytes(bad))[0] == 5
    # truncated at every length: always Error::Read (read_exact fails)
    for cut in range(len(good)):
        assert oracle.read_chunk(good[:cut] or b"\0"[:0] or bytes(0))[0] == 1 if cut else True
    # declared length above the 1 GiB cap is rejected before anything else (container.rs:49-52)
    over = struct.pack(">I", (1 << 30) + 1) + b"AMRG"
    assert oracle.read_chunk(over)[0] == 3
    at_cap = struct.pack(">I", 1 << 30) + b"AMRG"
    assert oracle.read_chunk(at_cap)[0] == 1  # allowed size, but the data is not there
    with pytest.raises(py_oracle.ContainerError) as e:
        py_oracle.read_chunk(over)
    assert e.value.kind == "ReadInvalidChunkSize"
    # header: bad magic and short input (container.rs:107-121)
    hdr = bytearray(9)
    assert L.oracle_write_header((__import__("ctypes").c_char * 9).from_buffer(hdr), 9) == 9
    assert bytes(hdr) == py_oracle.HEADER
    for i in range(9):
        b = bytearray(py_oracle.HEADER)
        b[i] ^= 0x20
        with pytest.raises(py_oracle.ContainerError):
            py_oracle.read_header(bytes(b))
    with pytest.raises(py_oracle.ContainerError):
        py_oracle.read_header(py_oracle.HEADER[:8])


def test_document_framing_oracles_agree_with_golden(oracle):
    # document.rs:53-77 with an opaque payload: C oracle == Python oracle == the fixture built
    # independently with struct.pack + zlib (tests/golden/make_golden.py)
    import json
    import os
    from oracle import py_oracle
    with open(os.path.join(os.path.dirname(__file__), "golden", "container_cases.json")) as f:
        g = json.load(f)["document_framing"]
    payload, want = bytes.fromhex(g["amrg_data"]), bytes.fromhex(g["file"])
    assert oracle.write_document(payload) == want == py_oracle.write_document(payload)
    assert oracle.read_document(want) == (0, 5, 17, len(payload), len(want))
    assert py_oracle.read_document(want) == payload
    assert oracle.write_document(b"") == py_oracle.write_document(b"")
    # the stage at which each damaged file stops, and with which error
    cases = {
        "magic": (bytes([want[0] ^ 1]) + want[1:], 1, 0),
        "short header": (want[:5], 1, 0),
        "payload bit": (want[:40] + bytes([want[40] ^ 8]) + want[41:], 5, 1),
        "oversize": (want[:9] + b"\x40\x00\x00\x01" + want[13:], 3, 1),
        "amrg type": (py_oracle.HEADER + py_oracle.write_chunk(b"AMRX", payload) + py_oracle.write_chunk(b"SEND", b""), 1, 2),
        "send cut": (want[:-1], 1, 3),
        "send crc": (want[:-1] + bytes([want[-1] ^ 1]), 5, 3),
        "send type": (py_oracle.HEADER + py_oracle.write_chunk(b"AMRG", payload) + py_oracle.write_chunk(b"AMRG", b""), 1, 4),
    }
    for name, (f_, status, stage) in cases.items():
        got = oracle.read_document(f_)
        assert got[:2] == (status, stage), (name, got)
        with pytest.raises(py_oracle.ContainerError) as e:
            py_oracle.read_document(f_)
        assert {"Read": 1, "ReadInvalidChunkSize": 3, "ReadCrcMismatch": 5}[e.value.kind] == status, name
