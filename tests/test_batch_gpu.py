"""GPU parity for the many-buffers-at-once entry points (one wave per buffer)."""
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batch_crc_vs_oracle(gpu, oracle):
    rng = np.random.default_rng(0x51F7)
    blob = rng.integers(0, 256, size=24 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    # every length class of the wave kernel, every alignment, plus a few long ones
    lens = list(range(0, 80)) + [127, 128, 1007, 1008, 1009, 1023, 1024, 1025, 1039, 1040, 1041,
                                 2047, 2048, 2049, 4096, 65535, 65536, 300001, (1 << 20),
                                 (1 << 20) + 1, (3 << 20) + 7]
    lens = lens + [int(x) for x in rng.integers(0, 5000, size=3000)]
    offs, pos = [], 0
    for i, n in enumerate(lens):
        pos += int(rng.integers(0, 19))  # ragged gaps -> every alignment
        if pos + n > blob.size:
            pos = int(rng.integers(0, 64))
        offs.append(pos)
        pos += n
    for state, xo in ((0xFFFFFFFF, 0xFFFFFFFF), (0, 0), (0x12345678, 0)):
        got = gpu.compute_batch(state, d, offs, lens, xor_out=xo).cpu().numpy().view(np.uint32)
        for i, (o, n) in enumerate(zip(offs, lens)):
            want = oracle.compute(state, blob[o:o + n]) ^ xo
            assert int(got[i]) == want, (i, o, n, hex(int(got[i])), hex(want))


def test_batch_empty_and_single(gpu, oracle):
    d = torch.arange(0, 200, dtype=torch.uint8, device="cuda")
    assert gpu.compute_batch(0xFFFFFFFF, d, [], []).numel() == 0
    got = gpu.compute_batch(0xFFFFFFFF, d, [3], [100]).cpu().numpy().view(np.uint32)
    assert int(got[0]) == oracle.compute(0xFFFFFFFF, d.cpu().numpy()[3:103])


def test_read_chunks_device(gpu, oracle):
    from sift_amd import container
    rng = np.random.default_rng(7)
    stream = bytearray()
    offs, want_status, want_len = [], [], []
    sizes = [0, 1, 3, 4, 11, 12, 13, 27, 28, 29, 500, 1012, 1013, 4096, 70000, 2_500_000] + \
            [int(x) for x in rng.integers(0, 3000, size=500)]
    for i, n in enumerate(sizes):
        data = rng.integers(0, 256, size=n, dtype=np.uint8)
        ser = bytearray(oracle.write_chunk(b"AMRG" if i % 3 else b"SEND", data))
        kind = i % 7
        if kind == 3 and n > 0:      # corrupt one data byte
            ser[8 + n // 2] ^= 0x40
            st = 5
        elif kind == 5:              # corrupt stored crc
            ser[-1] ^= 1
            st = 5
        elif kind == 6 and n > 2:    # type byte flipped
            ser[5] ^= 0x20
            st = 5
        else:
            st = 0
        stream += bytes(int(rng.integers(0, 5)))  # ragged padding between chunks
        offs.append(len(stream))
        stream += ser
        want_status.append(st)
        want_len.append(n if st == 0 else 0)
    # an oversize length field and a truncated chunk at the very end
    offs.append(len(stream)); stream += struct.pack(">I", (1 << 30) + 1) + b"AMRG"; want_status.append(3); want_len.append(0)
    offs.append(len(stream)); stream += oracle.write_chunk(b"AMRG", bytes(100))[:-3]; want_status.append(1); want_len.append(0)
    blob = np.frombuffer(bytes(stream), dtype=np.uint8)
    d = torch.from_numpy(blob.copy()).cuda()
    status, dlen = container.read_chunks_device(d, offs)
    status, dlen = status.cpu().tolist(), dlen.cpu().tolist()
    assert status == want_status
    assert dlen == want_len
    # same verdicts as the oracle's Chunk::read on the same windows
    for o, st in zip(offs, status):
        assert oracle.read_chunk(bytes(stream), o)[0] == st
    # windows clipped to the next chunk: a chunk cut short is Error::Read
    avail = [b - a for a, b in zip(offs, offs[1:] + [len(stream)])]
    avail[14] = 12 + sizes[14] - 1   # chunk 14 is intact; one byte short of its stored crc
    avail[15] = 12 + sizes[15]       # exactly enough
    s2, _ = container.read_chunks_device(d, offs, avail)
    s2 = s2.cpu().tolist()
    assert want_status[14] == 0 and s2[14] == 1
    assert s2[:14] == want_status[:14] and s2[15:] == want_status[15:]


def test_read_documents_device(gpu, oracle):
    # document.rs:53-77 framing on many files at once, against the oracle's
    # read_header / read_chunk / type checks applied in the reference's order
    from oracle import py_oracle
    from sift_amd import container
    rng = np.random.default_rng(11)

    def make(payload, mutate):
        f = bytearray(py_oracle.HEADER + oracle.write_chunk(b"AMRG", payload) + oracle.write_chunk(b"SEND", b""))
        if mutate == 1:
            f[3] ^= 0x01                       # bad magic
        elif mutate == 2 and len(payload):
            f[17 + len(payload) // 2] ^= 0x80  # payload bit flip
        elif mutate == 3:
            f = bytearray(py_oracle.HEADER + oracle.write_chunk(b"SEND", payload) + oracle.write_chunk(b"SEND", b""))
        elif mutate == 4:
            f = bytearray(py_oracle.HEADER + oracle.write_chunk(b"AMRG", payload) + oracle.write_chunk(b"AMRG", b""))
        elif mutate == 5:
            f = f[:-2]                         # truncated SEND chunk
        elif mutate == 6:
            f[-1] ^= 0x04                      # SEND crc damaged
        elif mutate == 7:
            f = f[:5]                          # shorter than the header
        elif mutate == 8:
            f += b"trailing bytes are ignored by read_document"
        return bytes(f)

    def expect(f):
        try:
            pos = py_oracle.read_header(f)
            t, d, pos = py_oracle.read_chunk(f, pos)
            if t != b"AMRG":
                raise py_oracle.ContainerError("Read")
            t2, _, pos = py_oracle.read_chunk(f, pos)
            if t2 != b"SEND":
                raise py_oracle.ContainerError("Read")
            return 0, len(d)
        except py_oracle.ContainerError as e:
            return {"Read": 1, "ReadInvalidChunkSize": 3, "ReadCrcMismatch": 5}[e.kind], 0

    files, offs, avail, want = bytearray(), [], [], []
    for i in range(400):
        n = int(rng.integers(0, 6000)) if i % 50 else 1_200_000
        f = make(rng.integers(0, 256, size=n, dtype=np.uint8).tobytes(), i % 9)
        files += bytes(int(rng.integers(0, 7)))
        offs.append(len(files))
        avail.append(len(f))
        files += f
        want.append(expect(f))
    d = torch.from_numpy(np.frombuffer(bytes(files), dtype=np.uint8).copy()).cuda()
    status, alen = container.read_documents_device(d, offs, avail)
    got = list(zip(status.cpu().tolist(), alen.cpu().tolist()))
    assert got == want
    assert {w[0] for w in want} == {0, 1, 5}


def test_write_chunk_crcs_device(gpu, oracle):
    # Chunk::write x n in place: lay out length | type | data | zeroed slot, let the GPU
    # fill the CRC slots, compare every byte with the oracle's Chunk::write.
    import struct as st
    from sift_amd import container
    rng = np.random.default_rng(13)
    blob, want, offs, avail = bytearray(), bytearray(), [], []
    for i, n in enumerate([0, 1, 5, 19, 20, 21, 1000, 65536, 1_500_000] + [int(x) for x in rng.integers(0, 4000, 300)]):
        t = b"AMRG" if i % 2 else b"SEND"
        d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        pad = bytes(int(rng.integers(0, 9)))
        blob += pad
        want += pad
        offs.append(len(blob))
        avail.append(12 + n)
        blob += st.pack(">I", n) + t + d + b"\0\0\0\0"
        want += oracle.write_chunk(t, d)
    d = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
    status = container.write_chunk_crcs_device(d, offs, avail)
    assert set(status.cpu().tolist()) == {0}
    assert bytes(d.cpu().numpy()) == bytes(want)
    # and what was written reads back clean
    s2, dl = container.read_chunks_device(d, offs, avail)
    assert set(s2.cpu().tolist()) == {0}
    # a slot that is too small is Error::Write and leaves the bytes alone
    avail[3] -= 1
    d2 = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
    s3 = container.write_chunk_crcs_device(d2, offs, avail).cpu().tolist()
    assert s3[3] == 2 and s3[:3] == [0, 0, 0]


def test_device_resident_descriptors(gpu, oracle):
    # offsets / lengths already in HBM: used in place, no host copy, and long
    # buffers stay on one wave each (the host never sees the lengths)
    from sift_amd import container
    rng = np.random.default_rng(17)
    blob = rng.integers(0, 256, size=8 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    lens = [0, 5, 31, 32, 33, 1000, 70000, (2 << 20) + 3] + [int(x) for x in rng.integers(0, 3000, 200)]
    offs = [int(x) for x in rng.integers(0, blob.size - max(lens), len(lens))]
    doff = torch.tensor(offs, dtype=torch.int64, device="cuda")
    dlen = torch.tensor(lens, dtype=torch.int64, device="cuda")
    got = gpu.compute_batch(0xFFFFFFFF, d, doff, dlen, xor_out=0xFFFFFFFF).cpu().numpy().view(np.uint32)
    for i, (o, n) in enumerate(zip(offs, lens)):
        assert int(got[i]) == oracle.crc32(blob[o:o + n]), (i, o, n)
    # chunk verification with device descriptors
    ser = bytearray()
    coffs = []
    for n in (0, 7, 500, 40000):
        coffs.append(len(ser))
        ser += oracle.write_chunk(b"AMRG", rng.integers(0, 256, size=n, dtype=np.uint8))
    ser[coffs[2] + 20] ^= 1
    dc = torch.from_numpy(np.frombuffer(bytes(ser), dtype=np.uint8).copy()).cuda()
    st, dl = container.read_chunks_device(dc, torch.tensor(coffs, dtype=torch.int64, device="cuda"))
    assert st.cpu().tolist() == [0, 0, 5, 0] and dl.cpu().tolist() == [0, 7, 0, 40000]


@pytest.mark.parametrize("top", [31, 64, 65, 128, 129, 200, 256, 257, 400, 1024, 1025, 2049, 4080, 4081])
def test_batch_short_items_every_group_width(gpu, top):
    # the group's shape follows the longest 32..4080-byte span of a turn (2 lanes up to 64
    # bytes, 4 x 2 vectors up to 128, 4 x 4 up to 256, 8 lanes beyond), its framing too
    # (from the spans' ends up to 1 KiB, memory's 16-byte blocks beyond).  Lengths 0..top at
    # every alignment, with counts that leave the last pass and the last turn ragged; one
    # span starts at the first byte of the allocation and one ends at its last.
    import zlib
    rng = np.random.default_rng(1000 + top)
    blob = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    M = 0xFFFFFFFF
    for n in (1, 3, 17, 64, 100, 4099, 20000):
        lens = [int(x) for x in rng.integers(0, top + 1, size=n)]
        if n >= 17:
            lens[5], lens[11] = top, max(top - 1, 0)
        offs = [int(rng.integers(0, blob.size - k + 1)) for k in lens]
        if n >= 3:
            offs[0], offs[1] = 0, blob.size - lens[1]
        for state, xo in ((M, M), (0x9E3779B9, 0)):
            got = gpu.compute_batch(state, d, offs, lens, xor_out=xo).cpu().numpy().view(np.uint32)
            for i, (o, k) in enumerate(zip(offs, lens)):
                want = (zlib.crc32(hb[o:o + k], state ^ M) ^ M) ^ xo
                assert int(got[i]) == want, (n, i, o, k, hex(int(got[i])), hex(want))


def test_batch_size_classes(gpu):
    # one wave / cut into 256 KiB pieces and stitched / whole-chip launch, in one batch
    import zlib
    rng = np.random.default_rng(23)
    blob = rng.integers(0, 256, size=(80 << 20) + 64, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    lens = [100, 262144, 262145, 524288, 524289, (5 << 20) + 3, (63 << 20) + 1, 64 << 20, (70 << 20) + 9,
            0, 31, 32, 300000, (1 << 20) - 1]
    offs = [int(rng.integers(0, blob.size - n + 1)) for n in lens]
    M = 0xFFFFFFFF
    for state, xo in ((M, M), (0x1234ABCD, 0)):
        got = gpu.compute_batch(state, d, offs, lens, xor_out=xo).cpu().numpy().view(np.uint32)
        for i, (o, n) in enumerate(zip(offs, lens)):
            want = (zlib.crc32(hb[o:o + n], state ^ M) ^ M) ^ xo
            assert int(got[i]) == want, (i, o, n, hex(int(got[i])), hex(want))


def test_batch_whole_chip_buffers_keep_their_slots(gpu):
    # buffers of 64 MiB and more leave the wave kernel's work list; the small ones
    # behind them must still land in their own result slots (no buffer in the
    # 256 KiB .. 64 MiB class here, so nothing is cut)
    import zlib
    rng = np.random.default_rng(29)
    blob = rng.integers(0, 256, size=(140 << 20) + 64, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    M = 0xFFFFFFFF
    for lens in ([64 << 20, 100], [100, 64 << 20, 31, (70 << 20) + 5, 5000], [(65 << 20) + 1]):
        offs, pos = [], 0
        for n in lens:
            offs.append(pos + 3)
            pos += n + 3
        for state, xo in ((M, M), (0x1234ABCD, 0)):
            out = torch.full((len(lens),), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            got = gpu.compute_batch(state, d, offs, lens, xor_out=xo, out=out).cpu().numpy().view(np.uint32)
            for i, (o, n) in enumerate(zip(offs, lens)):
                want = (zlib.crc32(hb[o:o + n], state ^ M) ^ M) ^ xo
                assert int(got[i]) == want, (lens, i, hex(int(got[i])), hex(want))


def test_batch_argument_errors_and_edges(gpu, oracle):
    import ctypes as C
    from sift_amd import _lib, container
    d = torch.arange(0, 256, dtype=torch.uint8, device="cuda").repeat(64)
    out = torch.zeros(8, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    host = np.array([0, 8], dtype=np.uint64)
    dev = torch.tensor([0, 8], dtype=torch.int64, device="cuda")
    # one array on the host, the other on the device: refused, not dereferenced
    rc = _lib.lib.sift_crc32_batch_device(0, 0, 0, d.data_ptr(), host.ctypes.data, dev.data_ptr(), 2, out.data_ptr(), s)
    assert rc == _lib.ERR_INVALID_ARG and b"both" in _lib.lib.sift_last_error()
    rc = _lib.lib.sift_crc32_batch_device(0, 0, 0, d.data_ptr(), dev.data_ptr(), host.ctypes.data, 2, out.data_ptr(), s)
    assert rc == _lib.ERR_INVALID_ARG
    rc = _lib.lib.sift_crc32_batch_device(0, 0, 0, d.data_ptr(), None, host.ctypes.data, 2, out.data_ptr(), s)
    assert rc == _lib.ERR_INVALID_ARG
    # n = 0 does nothing and succeeds
    assert _lib.lib.sift_crc32_batch_device(0, 0, 0, d.data_ptr(), None, None, 0, out.data_ptr(), s) == _lib.OK
    # pinned host descriptors are read in place
    po = torch.tensor([3, 100, 0], dtype=torch.int64).pin_memory()
    pl = torch.tensor([50, 5000, 16384], dtype=torch.int64).pin_memory()
    rc = _lib.lib.sift_crc32_batch_device(0, 0xFFFFFFFF, 0xFFFFFFFF, d.data_ptr(), po.data_ptr(), pl.data_ptr(), 3,
                                          out.data_ptr(), s)
    assert rc == _lib.OK
    torch.cuda.synchronize()
    h = d.cpu().numpy()
    assert [int(v) & 0xFFFFFFFF for v in out[:3].cpu().tolist()] == \
        [oracle.crc32(h[3:53]), oracle.crc32(h[100:5100]), oracle.crc32(h[:16384])]
    # the Python mirror refuses a window that reaches past `base` (no u64 wrap-around)
    with pytest.raises(ValueError):
        gpu.compute_batch(0, d, [d.numel() - 4], [5])
    with pytest.raises(ValueError):
        gpu.compute_batch(0, d, [(1 << 64) - 2], [3])
    # exactly 64 and 65 items: the self-planned form and the filed form meet here
    rng = np.random.default_rng(71)
    for n in (64, 65):
        lens = [int(x) for x in rng.integers(0, 200, size=n)]
        offs = [int(x) for x in rng.integers(0, d.numel() - 200, size=n)]
        got = gpu.compute_batch(0xFFFFFFFF, d, offs, lens, xor_out=0xFFFFFFFF).cpu().numpy().view(np.uint32)
        assert [int(g) for g in got] == [oracle.crc32(h[o:o + m]) for o, m in zip(offs, lens)]
    # back-to-back calls with host descriptors reuse the two pinned staging buffers
    outs = [gpu.compute_batch(0xFFFFFFFF, d, [k, k + 1], [1000 + k, 7 * k], xor_out=0) for k in range(12)]
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert [int(v) & 0xFFFFFFFF for v in o.cpu().tolist()] == \
            [oracle.compute(0xFFFFFFFF, h[k:k + 1000 + k]), oracle.compute(0xFFFFFFFF, h[k + 1:k + 1 + 7 * k])]


def test_descriptor_locations_and_batch_sizes(gpu, oracle):
    # The same items with descriptors in pageable host memory, in pinned host memory
    # and in HBM, for batch sizes on both sides of every path switch: up to 64 items
    # the host's descriptors travel in the kernel arguments; up to 4096 the host looks
    # at the lengths and, if none can be long, launches the wave kernel alone; with a
    # long item among them the piece kernel follows.  Same bits on every path.
    rng = np.random.default_rng(77)
    blob = rng.integers(0, 256, size=12 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    for n, with_long in ((1, False), (1, True), (7, True), (64, False), (64, True), (65, False), (65, True),
                         (300, False), (300, True), (4096, False), (4097, False)):
        lens = [int(x) for x in rng.integers(0, 6000, size=n)]
        if with_long:
            lens[n // 2] = (2 << 20) + 11
        offs, pos = [], 0
        for ln in lens:
            pos += int(rng.integers(0, 23))
            if pos + ln > blob.size:
                pos = int(rng.integers(0, 64))
            offs.append(pos)
            pos += ln
        want = np.array([oracle.compute(0x2468ACE0, blob[o:o + ln]) for o, ln in zip(offs, lens)], dtype=np.uint32)
        pin_o = torch.tensor(offs, dtype=torch.int64).pin_memory()
        pin_l = torch.tensor(lens, dtype=torch.int64).pin_memory()
        variants = {
            "pageable": (offs, lens),
            "pinned": (pin_o.numpy().view(np.uint64), pin_l.numpy().view(np.uint64)),
            "device": (torch.tensor(offs, dtype=torch.int64, device="cuda"),
                       torch.tensor(lens, dtype=torch.int64, device="cuda")),
        }
        for name, (o, ln) in variants.items():
            for rep in range(2):  # twice in a row: staging buffers and scratch are reused
                got = gpu.compute_batch(0x2468ACE0, d, o, ln).cpu().numpy().view(np.uint32)
                assert (got == want).all(), (n, with_long, name, rep, np.nonzero(got != want)[0][:5])


def test_small_chunk_batches_from_host_descriptors(gpu, oracle):
    # Chunk::read and Chunk::write's crc step on a handful of chunks described from the host
    # (kernel-argument descriptors), including a corrupted one and a truncated one.
    from sift_amd import container
    rng = np.random.default_rng(78)
    stream = bytearray()
    offs, avail, want = [], [], []
    for i, n in enumerate([0, 5, 100, 4096, 70000, 1 << 20, 33]):
        data = rng.integers(0, 256, size=n, dtype=np.uint8)
        ser = bytearray(oracle.write_chunk(b"AMRG", data))
        if i == 2:
            ser[10] ^= 1
        offs.append(len(stream))
        avail.append(len(ser) - (3 if i == 4 else 0))
        st, _, _, dl, _ = oracle.read_chunk(bytes(ser[:avail[-1]]))
        want.append((st, dl if st == 0 else 0))
        stream += ser
    d = torch.from_numpy(np.frombuffer(bytes(stream), dtype=np.uint8).copy()).cuda()
    status, dlen = container.read_chunks_device(d, offs, avail)
    assert list(zip(status.cpu().tolist(), dlen.cpu().tolist())) == want


def test_many_small_host_descriptor_calls_in_a_row(gpu, oracle):
    # Pageable descriptors of 65 .. 4096 items go through a ring of 16 small pinned
    # buffers whose reuse is guarded by one event per half: 50 calls with no host
    # sync in between, every result checked afterwards.
    rng = np.random.default_rng(79)
    blob = rng.integers(0, 256, size=4 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    calls = []
    for k in range(50):
        n = int(rng.integers(65, 400))
        lens = [int(x) for x in rng.integers(0, 3000, size=n)]
        offs = [int(x) for x in rng.integers(0, blob.size - 3000, size=n)]
        calls.append((offs, lens, gpu.compute_batch(0xFFFFFFFF, d, offs, lens)))
    torch.cuda.synchronize()
    for k, (offs, lens, out) in enumerate(calls):
        got = out.cpu().numpy().view(np.uint32)
        for i in (0, len(offs) // 2, len(offs) - 1):
            assert int(got[i]) == oracle.compute(0xFFFFFFFF, blob[offs[i]:offs[i] + lens[i]]), (k, i)
        if k % 7 == 0:
            want = np.array([oracle.compute(0xFFFFFFFF, blob[o:o + ln]) for o, ln in zip(offs, lens)], dtype=np.uint32)
            assert (got == want).all(), k


def test_randomised_rounds(gpu, oracle):
    # tests/fuzz_batch.py: random chunk / save-file streams with random corruptions, batch
    # sizes on both sides of every path switch, descriptors on the host or in HBM
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_batch", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_batch.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    for seed in range(7000, 7010):
        fuzz.one_round(seed)
