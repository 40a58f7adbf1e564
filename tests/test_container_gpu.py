"""GPU parity for the container framing (container.rs) with CRCs from the HIP
path, through the C ABI, against the oracle and the golden fixtures."""
import json
import os
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _golden():
    with open(os.path.join(GOLDEN, "container_cases.json")) as f:
        return json.load(f)


def test_write_chunk_matches_golden_and_oracle(gpu, oracle):
    from sift_amd import container
    for c in _golden()["chunks"]:
        t, d, ser = c["type"].encode(), bytes.fromhex(c["data"]), bytes.fromhex(c["serialized"])
        out = bytearray()
        container.write_chunk(container.Chunk(t, d), out)
        assert bytes(out) == ser == oracle.write_chunk(t, d)
        chunk, pos = container.read_chunk(ser)
        assert (chunk.chunk_type, chunk.data, pos) == (t, d, len(ser))


def test_document_framing_round_trip(gpu):
    # document.rs:53-77: header, AMRG chunk, empty SEND chunk; read back with expect_type
    from sift_amd import container
    g = _golden()["document_framing"]
    payload = bytes.fromhex(g["amrg_data"])
    out = bytearray()
    container.write_header(out)
    container.write_chunk(container.Chunk(b"AMRG", payload), out)
    container.write_chunk(container.Chunk(b"SEND", b""), out)
    assert bytes(out) == bytes.fromhex(g["file"])
    pos = container.read_header(bytes(out))
    c1, pos = container.read_chunk(bytes(out), pos)
    c1.expect_type(b"AMRG")
    c2, pos = container.read_chunk(bytes(out), pos)
    c2.expect_type(b"SEND")
    assert c1.data == payload and c2.data == b"" and pos == len(out)
    with pytest.raises(container.Error) as e:
        c1.expect_type(b"SEND")
    assert e.value.kind == "Read"


def test_large_chunk_and_errors(gpu, oracle):
    from sift_amd import container
    data = np.random.default_rng(0x51F7).integers(0, 256, size=(48 << 20) + 5, dtype=np.uint8).tobytes()
    out = bytearray()
    container.write_chunk(container.Chunk(b"AMRG", data), out)
    ser = bytes(out)
    assert ser[-4:] == struct.pack(">I", oracle.chunk_crc(b"AMRG", data))
    chunk, pos = container.read_chunk(ser)
    assert chunk.data == data and pos == len(ser)
    # any flipped bit in type, data or stored crc -> ReadCrcMismatch (container.rs:60-63)
    for where in (4, 8, 8 + len(data) // 2, len(ser) - 5, len(ser) - 1):
        bad = bytearray(ser)
        bad[where] ^= 0x10
        with pytest.raises(container.Error) as e:
            container.read_chunk(bytes(bad))
        assert e.value.kind == "ReadCrcMismatch", where
    # flipped length byte: the verdict is whatever Chunk::read gives on the same bytes
    kinds = {1: "Read", 3: "ReadInvalidChunkSize", 5: "ReadCrcMismatch"}
    for where, bit in ((2, 0x01), (3, 0x01), (0, 0x40), (1, 0x80)):
        bad = bytearray(ser)
        bad[where] ^= bit
        want = oracle.read_chunk(bytes(bad))[0]
        assert want in kinds
        with pytest.raises(container.Error) as e:
            container.read_chunk(bytes(bad))
        assert e.value.kind == kinds[want], (where, bit)
    with pytest.raises(container.Error) as e:
        container.read_chunk(struct.pack(">I", (1 << 30) + 1) + b"AMRG")
    assert e.value.kind == "ReadInvalidChunkSize"
    with pytest.raises(container.Error) as e:
        container.read_chunk(ser[:-1])
    assert e.value.kind == "Read"
    with pytest.raises(container.Error):
        container.read_header(b"\x89SIFX\r\n\x1a\n")


def test_chunk_crc_device_resident(gpu, oracle):
    # a serialized chunk already in HBM: CRC over type|data in place (container.rs:79-84)
    import ctypes as C
    from sift_amd import _lib
    data = np.random.default_rng(2).integers(0, 256, size=3_000_001, dtype=np.uint8)
    ser = np.frombuffer(oracle.write_chunk(b"AMRG", data), dtype=np.uint8)
    d = torch.from_numpy(ser.copy()).cuda()
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.sift_chunk_crc_device(0, d.data_ptr() + 4, data.size, out.data_ptr(), None,
                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (int(out.item()) & 0xFFFFFFFF) == struct.unpack(">I", ser[-4:].tobytes())[0]


def test_write_and_read_document_through_the_c_abi(gpu, oracle):
    # document.rs:53-77: sift_write_document / sift_read_document against the oracle and the
    # golden file, then the same nine mutations the many-files kernel is tested with
    from oracle import py_oracle
    from sift_amd import container
    g = _golden()["document_framing"]
    payload, want = bytes.fromhex(g["amrg_data"]), bytes.fromhex(g["file"])
    out = bytearray()
    container.write_document(payload, out)
    assert bytes(out) == want == oracle.write_document(payload)
    assert container.read_document(want) == payload
    rng = np.random.default_rng(61)
    kinds = {1: "Read", 3: "ReadInvalidChunkSize", 5: "ReadCrcMismatch"}
    for n in (0, 1, 777, 100_000, (3 << 20) + 5):
        payload = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        out = bytearray()
        container.write_document(payload, out)
        good = bytes(out)
        assert good == oracle.write_document(payload)
        for mutate in range(10):
            f = bytearray(good)
            if mutate == 1:
                f[3] ^= 0x01
            elif mutate == 2 and n:
                f[17 + n // 2] ^= 0x80
            elif mutate == 3:
                f = bytearray(py_oracle.HEADER + oracle.write_chunk(b"SEND", payload) + oracle.write_chunk(b"SEND", b""))
            elif mutate == 4:
                f = bytearray(py_oracle.HEADER + oracle.write_chunk(b"AMRG", payload) + oracle.write_chunk(b"AMRG", b""))
            elif mutate == 5:
                f = f[:-2]
            elif mutate == 6:
                f[-1] ^= 0x04
            elif mutate == 7:
                f = f[:5]
            elif mutate == 8:
                f += b"trailing bytes are ignored by read_document"
            elif mutate == 9:
                f[9:13] = struct.pack(">I", (1 << 30) + 1)
            status, stage, off, ln, used = oracle.read_document(bytes(f))
            seen = []
            if status == 0:
                assert container.read_document(bytes(f), decode=lambda p: seen.append(p) or p) == payload
                assert seen == [payload]
            else:
                with pytest.raises(container.Error) as e:
                    container.read_document(bytes(f), decode=lambda p: seen.append(p) or p)
                assert e.value.kind == kinds[status], (n, mutate)
                # the decoder ran exactly when the reference would have reached decode_document
                assert (len(seen) == 1) == (stage >= 3), (n, mutate, stage)
    # the C entry point reports the same stage as the oracle
    import ctypes as C
    from sift_amd import _lib
    bad = bytearray(want)
    bad[-1] ^= 1
    a = np.frombuffer(bytes(bad), dtype=np.uint8)
    off, ln, stage = C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    rc = _lib.lib.sift_read_document(0, a.ctypes.data, a.size, C.byref(off), C.byref(ln), C.byref(stage))
    assert (-rc, stage.value, off.value, ln.value) == oracle.read_document(bytes(bad))[:4]
    # output too small: Error::Write from whichever write runs out of room
    small = np.empty(len(want) - 1, dtype=np.uint8)
    p = np.frombuffer(bytes.fromhex(g["amrg_data"]), dtype=np.uint8)
    assert _lib.lib.sift_write_document(0, p.ctypes.data, p.size, small.ctypes.data, small.size) == -_lib.ERR_WRITE
