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
