"""GPU parity for long items in the batch entry points: an item whose CRC span is
long is cut on the device into pieces that whole workgroups reduce; the piece
that completes an item gives it its verdict.  Every verdict is compared with the oracle's Chunk::read /
compute on the same bytes (container.rs:47-66, crc.rs:28-34), with the
descriptors on the host and in HBM, for corrupted, truncated and oversize chunks.
"""
import struct
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

M = 0xFFFFFFFF


def _dev(xs):
    return torch.tensor(xs, dtype=torch.int64, device="cuda")


@pytest.fixture()
def tune(gpu):
    from sift_amd import _lib as L
    yield lambda key, value: L.check(L.lib.sift_tune(key, value))
    L.check(L.lib.sift_tune(L.TUNE_BATCH_CUT, 1))
    L.check(L.lib.sift_tune(L.TUNE_BATCH_WAVE_MAX, 0))
    L.check(L.lib.sift_tune(L.TUNE_BATCH_PIECE_LOG2, 16))
    L.check(L.lib.sift_tune(L.TUNE_BATCH_PLAN, 1))


def test_cut_crc_mixed_lengths(gpu, tune):
    # crc mode: short items (one wave) and long ones (cut) in one batch, every
    # alignment of the first and last byte, lengths around every piece boundary
    from sift_amd import _lib as L
    rng = np.random.default_rng(0x51F7 + 1)
    blob = rng.integers(0, 256, size=(96 << 20) + 4096, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    P = 65536
    lens = [0, 5, 31, 32, 33, 1000, 32767, 32768, 32769, 32784, 40000, P - 1, P, P + 1, P + 15, P + 16, P + 17,
            2 * P - 1, 2 * P, 2 * P + 1, 3 * P + 7, 16 * P + 5, (1 << 20) + 1, (5 << 20) + 12345,
            (64 << 20) - 1, 64 << 20, (64 << 20) + 17, (70 << 20) + 3]
    lens += [int(x) for x in rng.integers(32769, 3 << 20, size=40)]
    offs = [int(rng.integers(0, blob.size - n + 1)) for n in lens]
    for i in range(16):                       # every start alignment for a cut item
        offs.append(4096 + i)
        lens.append(5 * P + 3 * i)
    for state, xo in ((M, M), (0, 0), (0x1234ABCD, 0)):
        want = [(zlib.crc32(hb[o:o + n], state ^ M) ^ M) ^ xo for o, n in zip(offs, lens)]
        got_h = gpu.compute_batch(state, d, offs, lens, xor_out=xo).cpu().numpy().view(np.uint32)
        got_d = gpu.compute_batch(state, d, _dev(offs), _dev(lens), xor_out=xo).cpu().numpy().view(np.uint32)
        for i, w in enumerate(want):
            assert int(got_h[i]) == w, ("host desc", i, offs[i], lens[i], hex(int(got_h[i])), hex(w))
            assert int(got_d[i]) == w, ("device desc", i, offs[i], lens[i], hex(int(got_d[i])), hex(w))
    # other piece sizes and thresholds give the same bits
    want = [zlib.crc32(hb[o:o + n]) for o, n in zip(offs, lens)]
    for wave_max, plog in ((4096, 14), (100000, 18), (1 << 20, 20)):
        tune(L.TUNE_BATCH_WAVE_MAX, wave_max)
        tune(L.TUNE_BATCH_PIECE_LOG2, plog)
        got = gpu.compute_batch(M, d, _dev(offs), _dev(lens), xor_out=M).cpu().numpy().view(np.uint32)
        assert [int(g) for g in got] == want, (wave_max, plog)
    # and so does the uncut form (one wave per item), on the shorter items
    tune(L.TUNE_BATCH_CUT, 0)
    short = [i for i, n in enumerate(lens) if n <= (5 << 20) + 12345]
    got = gpu.compute_batch(M, d, [offs[i] for i in short], [lens[i] for i in short], xor_out=M)
    assert [int(g) for g in got.cpu().numpy().view(np.uint32)] == [want[i] for i in short]


def test_few_items_planned_by_the_piece_kernel(gpu, tune):
    # up to 64 items skip the filing pass: every length class and alignment again,
    # in batches of 1, 2, 37 and 64 items, against the two-launch form and zlib
    from sift_amd import _lib as L
    rng = np.random.default_rng(0x51F7 + 5)
    blob = rng.integers(0, 256, size=(40 << 20) + 4096, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    P = 65536
    pool = [0, 1, 5, 31, 32, 33, 47, 48, 1000, 16383, 16384, 16385, 32768, 40000, P - 1, P, P + 1, P + 16,
            2 * P + 1, 5 * P + 3, (1 << 20) + 1, (3 << 20) + 12345, (30 << 20) + 7] + \
           [int(x) for x in rng.integers(0, 1 << 20, size=41)]
    for count in (1, 2, 37, 64):
        for rep in range(3):
            lens = [pool[int(k)] for k in rng.integers(0, len(pool), size=count)]
            if count == 1:
                lens = [[(30 << 20) + 7, 31, 5 * P + 3][rep]]
            offs = [int(rng.integers(0, blob.size - n + 1)) for n in lens]
            for state, xo in ((M, M), (0x1234ABCD, 0)):
                want = [(zlib.crc32(hb[o:o + n], state ^ M) ^ M) ^ xo for o, n in zip(offs, lens)]
                tune(L.TUNE_BATCH_PLAN, 1)
                got = gpu.compute_batch(state, d, _dev(offs), _dev(lens), xor_out=xo).cpu().numpy().view(np.uint32)
                assert [int(g) for g in got] == want, (count, rep, "planned")
                got = gpu.compute_batch(state, d, offs, lens, xor_out=xo).cpu().numpy().view(np.uint32)
                assert [int(g) for g in got] == want, (count, rep, "planned, host descriptors")
                tune(L.TUNE_BATCH_PLAN, 0)
                got = gpu.compute_batch(state, d, _dev(offs), _dev(lens), xor_out=xo).cpu().numpy().view(np.uint32)
                assert [int(g) for g in got] == want, (count, rep, "filed")


def test_cut_many_items_threshold(gpu):
    # n >= 4096 raises the one-wave threshold to 256 KiB: items on both sides of it
    rng = np.random.default_rng(3)
    blob = rng.integers(0, 256, size=24 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    lens = [int(x) for x in rng.integers(0, 3000, size=4200)]
    for k, n in ((7, 262144), (99, 262145), (1000, 300000), (2000, (2 << 20) + 9), (4199, 262143), (0, 1 << 20)):
        lens[k] = n
    offs = [int(rng.integers(0, blob.size - n + 1)) for n in lens]
    got = gpu.compute_batch(M, d, _dev(offs), _dev(lens), xor_out=M).cpu().numpy().view(np.uint32)
    for i, (o, n) in enumerate(zip(offs, lens)):
        assert int(got[i]) == zlib.crc32(hb[o:o + n]), (i, o, n)


def _chunks_case(oracle, rng, sizes):
    """Serialized chunks of `sizes` data bytes, every 5th..9th damaged in one of the
    ways the reference can notice; returns (stream, offsets, avail, mutation names)."""
    stream, offs, avail, kinds = bytearray(), [], [], []
    for i, n in enumerate(sizes):
        data = rng.integers(0, 256, size=n, dtype=np.uint8)
        ser = bytearray(oracle.write_chunk(b"AMRG" if i % 2 else b"SEND", data))
        kind = ("ok", "data", "crc", "type", "short", "ok", "len")[i % 7]
        if kind == "data" and n:
            ser[8 + (n * 2) // 3] ^= 0x04
        elif kind == "crc":
            ser[-2] ^= 0x80
        elif kind == "type":
            ser[6] ^= 0x01
        elif kind == "len" and n > 300:
            ser[3] ^= 0x01          # length field off by one: the stored crc is read from the wrong place
        stream += bytes(int(rng.integers(0, 17)))
        offs.append(len(stream))
        stream += ser
        avail.append(len(ser) - (1 + n // 3 if kind == "short" else 0))
        kinds.append(kind)
    stream += bytes(64)
    return bytes(stream), offs, avail, kinds


@pytest.mark.parametrize("plan", [1, 0])
def test_cut_read_chunks_vs_oracle(gpu, oracle, tune, plan):
    from sift_amd import container, _lib as L
    tune(L.TUNE_BATCH_PLAN, plan)      # 23 items: planned by the piece kernel, or filed by the wave kernel
    rng = np.random.default_rng(0x51F7 + 2)
    sizes = [0, 17, 32764, 32765, 40000, 65536, 65540, 100000, 131072 - 4, 131072, (1 << 20) + 3,
             (3 << 20) + 1, 5_000_000, (24 << 20) + 11, 70000, 300, 33000, (2 << 20) - 4, 65532, 99999, 50000]
    stream, offs, avail, kinds = _chunks_case(oracle, rng, sizes)
    # one oversize length field and one chunk cut inside its length field
    extra = len(stream)
    stream += struct.pack(">I", (1 << 30) + 1) + b"AMRG" + bytes(40)
    offs.append(extra); avail.append(48); kinds.append("oversize")
    offs.append(extra); avail.append(3); kinds.append("nolen")
    d = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    want = [oracle.read_chunk(stream[:o + a], o) for o, a in zip(offs, avail)]
    want_status = [w[0] for w in want]
    want_len = [w[3] if w[0] == 0 else 0 for w in want]
    assert {0, 1, 3, 5} <= set(want_status)
    for off_arg, av_arg in ((offs, avail), (_dev(offs), _dev(avail))):
        st, dl = container.read_chunks_device(d, off_arg, av_arg)
        assert st.cpu().tolist() == want_status, list(zip(kinds, st.cpu().tolist(), want_status))
        assert dl.cpu().tolist() == want_len


@pytest.mark.parametrize("plan", [1, 0])
def test_cut_write_then_read_large_chunks(gpu, oracle, tune, plan):
    from sift_amd import container, _lib as L
    tune(L.TUNE_BATCH_PLAN, plan)
    rng = np.random.default_rng(31)
    blob, want, offs, avail = bytearray(), bytearray(), [], []
    for i, n in enumerate([5, 40000, 65536, (1 << 20) + 7, (9 << 20) + 1, 32764, 32765, 100]):
        t = b"AMRG" if i % 2 else b"SEND"
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        pad = bytes(int(rng.integers(0, 16)))
        blob += pad
        want += pad
        offs.append(len(blob))
        avail.append(12 + n)
        blob += struct.pack(">I", n) + t + data + b"\xAA\xBB\xCC\xDD"
        want += oracle.write_chunk(t, data)
    d = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
    status = container.write_chunk_crcs_device(d, _dev(offs), _dev(avail))
    assert set(status.cpu().tolist()) == {0}
    assert bytes(d.cpu().numpy()) == bytes(want)
    st, dl = container.read_chunks_device(d, offs, avail)
    assert set(st.cpu().tolist()) == {0}
    assert dl.cpu().tolist() == [a - 12 for a in avail]


def test_cut_documents_with_large_payloads(gpu, oracle):
    # document.rs:62-77 framing where the AMRG chunk is long enough to be cut:
    # the verdict still follows the reference's order (AMRG crc, AMRG type, SEND ...)
    from oracle import py_oracle
    from sift_amd import container
    rng = np.random.default_rng(41)

    def make(payload, mutate):
        amrg = oracle.write_chunk(b"AMRG", payload)
        send = oracle.write_chunk(b"SEND", b"")
        f = bytearray(py_oracle.HEADER + amrg + send)
        if mutate == 1:
            f[17 + len(payload) // 2] ^= 0x80                      # payload bit flip -> 5
        elif mutate == 2:
            f = bytearray(py_oracle.HEADER + oracle.write_chunk(b"SEND", payload) + send)   # wrong first type -> 1
        elif mutate == 3:
            f = bytearray(py_oracle.HEADER + amrg + oracle.write_chunk(b"AMRG", b""))       # wrong second type -> 1
        elif mutate == 4:
            f = f[:-2]                                             # SEND cut short -> 1
        elif mutate == 5:
            f[-1] ^= 0x04                                          # SEND crc damaged -> 5
        elif mutate == 6:
            f[17 + len(payload) // 2] ^= 0x80                      # AMRG crc bad AND SEND damaged: AMRG's error wins
            f[-1] ^= 0x04
        elif mutate == 7:
            f = bytearray(py_oracle.HEADER + amrg + oracle.write_chunk(b"SEND", bytes(50000)))  # long SEND chunk, fine
        elif mutate == 8:
            f = f[: 9 + 8 + len(payload) + 2]                      # cut inside AMRG's stored crc -> 1
        return bytes(f)

    def expect(f):
        try:
            pos = py_oracle.read_header(f)
            t, dd, pos = py_oracle.read_chunk(f, pos)
            if t != b"AMRG":
                raise py_oracle.ContainerError("Read")
            t2, _, pos = py_oracle.read_chunk(f, pos)
            if t2 != b"SEND":
                raise py_oracle.ContainerError("Read")
            return 0, len(dd)
        except py_oracle.ContainerError as e:
            return {"Read": 1, "ReadInvalidChunkSize": 3, "ReadCrcMismatch": 5}[e.kind], 0

    files, offs, avail, want = bytearray(), [], [], []
    for i in range(27):
        n = (40000, 70000, (1 << 20) + 5)[i % 3]
        f = make(rng.integers(0, 256, size=n, dtype=np.uint8).tobytes(), i % 9)
        files += bytes(int(rng.integers(0, 16)))
        offs.append(len(files))
        avail.append(len(f))
        files += f
        want.append(expect(f))
    files += bytes(64)
    d = torch.from_numpy(np.frombuffer(bytes(files), dtype=np.uint8).copy()).cuda()
    for off_arg, av_arg in ((offs, avail), (_dev(offs), _dev(avail))):
        status, alen = container.read_documents_device(d, off_arg, av_arg)
        assert list(zip(status.cpu().tolist(), alen.cpu().tolist())) == want
    assert {w[0] for w in want} == {0, 1, 5}


def test_cut_overflow_falls_back_to_one_wave(gpu, tune):
    # Items that overlap can need more records / pieces than the capacities the host
    # derives from the extent of `base`; a span that finds the arrays full is walked
    # by its own wave.  Same bits either way.
    from sift_amd import _lib as L
    rng = np.random.default_rng(51)
    blob = rng.integers(0, 256, size=3 << 20, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    tune(L.TUNE_BATCH_WAVE_MAX, 4096)
    tune(L.TUNE_BATCH_PIECE_LOG2, 14)
    n = 6000
    lens = [int(x) for x in rng.integers(200_000, 1_000_000, size=n)]   # all overlapping, all long
    offs = [int(rng.integers(0, blob.size - m + 1)) for m in lens]
    got = gpu.compute_batch(M, d, _dev(offs), _dev(lens), xor_out=M).cpu().numpy().view(np.uint32)
    for i in range(0, n, 7):
        assert int(got[i]) == zlib.crc32(hb[offs[i]:offs[i] + lens[i]]), i
    # spot-check is on every 7th; the rest must at least agree with a second run
    again = gpu.compute_batch(M, d, offs, lens, xor_out=M).cpu().numpy().view(np.uint32)
    assert (got == again).all()


def test_cut_full_size_chunks(gpu):
    # the sizes VERDICT names: one 256 MiB and one 1 GiB chunk (the container.rs:49
    # cap) through sift_read_chunks_device; the stored crc comes from the whole-chip
    # compute entry point and is checked against zlib on the host copy.
    from sift_amd import container
    for n in (256 << 20, 1 << 30):
        g = torch.Generator(device="cuda").manual_seed(0x51F7)
        buf = torch.empty(n + 12 + 5, dtype=torch.uint8, device="cuda")
        buf[5 + 8:5 + 8 + n] = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
        buf[5:5 + 8] = torch.tensor(list(struct.pack(">I", n) + b"AMRG"), dtype=torch.uint8, device="cuda")
        crc = gpu.compute(M, buf[5 + 4:5 + 8 + n]) ^ M
        assert crc == zlib.crc32(buf[5 + 4:5 + 8 + n].cpu().numpy().tobytes())
        buf[5 + 8 + n:] = torch.tensor(list(struct.pack(">I", crc)), dtype=torch.uint8, device="cuda")
        st, dl = container.read_chunks_device(buf, _dev([5]), _dev([n + 12]))
        assert st.cpu().tolist() == [0] and (dl.cpu().tolist()[0] & M) == n
        buf[5 + 8 + n // 2] ^= 1                              # one flipped bit in 1 GiB
        st, _ = container.read_chunks_device(buf, [5], [n + 12])
        assert st.cpu().tolist() == [5]
        buf[5 + 8 + n // 2] ^= 1
        st, _ = container.read_chunks_device(buf, [5], [n + 11])   # one byte short
        assert st.cpu().tolist() == [1]
        # Chunk::write's crc step on the same chunk, slot zeroed first
        buf[5 + 8 + n:] = 0
        assert container.write_chunk_crcs_device(buf, [5], [n + 12]).cpu().tolist() == [0]
        assert bytes(buf[5 + 8 + n:].cpu().numpy()) == struct.pack(">I", crc)
        del buf


def _wide_lens(rng, count, long_share):
    P = 65536
    edge = [32768, 32769, 32784, 40000, P - 1, P, P + 1, P + 16, 2 * P - 1, 2 * P + 1, 3 * P + 7, 5 * P + 3,
            (1 << 20) + 1, (3 << 20) + 12345, (9 << 20) + 5]
    lens = []
    for _ in range(count):
        u = rng.random()
        if u < long_share * 0.3:
            lens.append(edge[int(rng.integers(0, len(edge)))])
        elif u < long_share:
            lens.append(int(rng.integers(32769, 600_000)))
        else:
            lens.append(int(rng.integers(0, 9000)))
    return lens


@pytest.mark.parametrize("count", [65, 300, 1500, 4096])
def test_wide_plan_crc_mixed(gpu, tune, count):
    # 65..4096 items: the batch kernel keeps the spans that fit a wave, the piece kernel plans
    # the longer ones for itself (no filing).  Short and long items mixed, every alignment;
    # against zlib and against the filed form; then a batch without long items (the piece
    # kernel's flag stays down) and the first batch again (the two flags take turns).
    from sift_amd import _lib as L
    rng = np.random.default_rng(0x51F7 + 70 + count)
    blob = rng.integers(0, 256, size=(48 << 20) + 4096, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    lens = _wide_lens(rng, count, 0.35 if count <= 300 else 0.08)
    offs = [int(rng.integers(0, blob.size - n + 1)) for n in lens]
    short = [min(n, 3000) for n in lens]
    for state, xo in ((M, M), (0x1234ABCD, 0)):
        want = [(zlib.crc32(hb[o:o + n], state ^ M) ^ M) ^ xo for o, n in zip(offs, lens)]
        want_short = [(zlib.crc32(hb[o:o + n], state ^ M) ^ M) ^ xo for o, n in zip(offs, short)]
        for off_arg, len_arg, short_arg in ((_dev(offs), _dev(lens), _dev(short)), (offs, lens, short)):
            for rep in range(3):
                got = gpu.compute_batch(state, d, off_arg, len_arg, xor_out=xo).cpu().numpy().view(np.uint32)
                bad = [i for i, w in enumerate(want) if int(got[i]) != w]
                assert not bad, (count, rep, bad[:5], [lens[i] for i in bad[:5]])
                got = gpu.compute_batch(state, d, off_arg, short_arg, xor_out=xo).cpu().numpy().view(np.uint32)
                assert [int(g) for g in got] == want_short, (count, rep, "no long item")
        tune(L.TUNE_BATCH_PLAN, 0)
        got = gpu.compute_batch(state, d, _dev(offs), _dev(lens), xor_out=xo).cpu().numpy().view(np.uint32)
        assert [int(g) for g in got] == want, (count, "filed")
        tune(L.TUNE_BATCH_PLAN, 1)
    # other thresholds and piece sizes give the same bits (more items are long, pieces smaller or larger)
    want = [zlib.crc32(hb[o:o + n]) for o, n in zip(offs, lens)]
    for wave_max, plog in ((4096, 14), (100000, 18)):
        tune(L.TUNE_BATCH_WAVE_MAX, wave_max)
        tune(L.TUNE_BATCH_PIECE_LOG2, plog)
        got = gpu.compute_batch(M, d, _dev(offs), _dev(lens), xor_out=M).cpu().numpy().view(np.uint32)
        assert [int(g) for g in got] == want, (wave_max, plog)


@pytest.mark.parametrize("count", [70, 900])
def test_wide_plan_chunks_vs_oracle(gpu, oracle, tune, count):
    # Chunk::read and Chunk::write's crc step on 70 / 900 serialized chunks, short and long,
    # some damaged in each way the reference notices: wide plan against the oracle
    from sift_amd import container, _lib as L
    rng = np.random.default_rng(0x51F7 + 90 + count)
    sizes = [int(x) for x in rng.integers(0, 5000, size=count)]
    for k in rng.choice(count, size=max(14, count // 12), replace=False):
        sizes[int(k)] = int(rng.choice([32764, 32765, 40000, 65532, 65536, 65540, 131072, 200001, (1 << 20) + 3,
                                        (2 << 20) - 4, 3_000_000]))
    stream, offs, avail, kinds = _chunks_case(oracle, rng, sizes)
    extra = len(stream)
    stream += struct.pack(">I", (1 << 30) + 1) + b"AMRG" + bytes(40)
    offs.append(extra); avail.append(48); kinds.append("oversize")
    offs.append(extra); avail.append(3); kinds.append("nolen")
    d = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    want = [oracle.read_chunk(stream[:o + a], o) for o, a in zip(offs, avail)]
    want_status = [w[0] for w in want]
    want_len = [w[3] if w[0] == 0 else 0 for w in want]
    assert {0, 1, 3, 5} <= set(want_status)
    for plan in (1, 0):
        tune(L.TUNE_BATCH_PLAN, plan)
        for off_arg, av_arg in ((offs, avail), (_dev(offs), _dev(avail))):
            st, dl = container.read_chunks_device(d, off_arg, av_arg)
            got = st.cpu().tolist()
            bad = [(i, kinds[i], sizes[i] if i < len(sizes) else -1, got[i], want_status[i])
                   for i in range(len(got)) if got[i] != want_status[i]]
            assert not bad, (plan, bad[:6])
            assert dl.cpu().tolist() == want_len
    tune(L.TUNE_BATCH_PLAN, 1)
    # Chunk::write: zero every crc slot of the intact chunks, have the device fill them, read back
    ok = [i for i, k in enumerate(kinds) if k == "ok"]
    d2 = d.clone()
    for i in ok:
        d2[offs[i] + 8 + sizes[i]:offs[i] + 12 + sizes[i]] = 0
    st = container.write_chunk_crcs_device(d2, _dev([offs[i] for i in ok]), _dev([avail[i] for i in ok]))
    assert set(st.cpu().tolist()) == {0}
    assert bytes(d2.cpu().numpy()) == bytes(d.cpu().numpy())


def test_wide_plan_one_huge_item_among_small(gpu):
    # one item of 1.2 GiB (pieces of 2 MiB, runs capped at 1 GiB, many workgroups on one record)
    # among 99 short ones and a few of 100..900 KiB: the wide plan's accumulators and its
    # direct verdicts in one call; twice (the accumulators must be back at zero)
    n_big = (1200 << 20) + 12345
    g = torch.Generator(device="cuda").manual_seed(0x51F7 + 9)
    d = torch.randint(0, 256, (n_big + (8 << 20),), dtype=torch.uint8, device="cuda", generator=g)
    hb = d.cpu().numpy()
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.integers(0, 5000, size=99)] + [100_000, 333_333, 900_001, 65_537]
    offs = [int(rng.integers(0, (8 << 20) - 1_000_000)) + n_big for _ in lens]
    lens.insert(40, n_big)
    offs.insert(40, 7)
    want = [zlib.crc32(hb[o:o + n]) for o, n in zip(offs, lens)]
    for rep in range(2):
        got = gpu.compute_batch(M, d, _dev(offs), _dev(lens), xor_out=M).cpu().numpy().view(np.uint32)
        assert [int(x) for x in got] == want, rep


@pytest.mark.parametrize("count,size", [(2048, 65536 + 17), (300, 65536 + 17), (2600, 101_000), (5000, 40_000)])
def test_wave_threshold_from_the_lengths(gpu, count, size):
    # The span length above which an item is cut comes from the lengths themselves (a sample
    # every wave of both kernels counts alike; on the host when it can read them): thousands
    # of like items stay on waves, few are cut, and a lone long item among them is cut
    # whatever the count.  Same bits every way, descriptors in HBM and on the host.
    rng = np.random.default_rng(count)
    blob = rng.integers(0, 256, size=(160 << 20) + 4096, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    hb = blob.tobytes()
    lens = [size - int(x) for x in rng.integers(0, 9, size=count)]
    lens[count // 3] = 250_000          # a lone longer one
    lens[count // 2] = (3 << 20) + 5    # and a much longer one
    lens[5] = 40
    offs = [int(rng.integers(0, blob.size - n + 1)) for n in lens]
    want = [zlib.crc32(hb[o:o + n]) for o, n in zip(offs, lens)]
    for off_arg, len_arg in ((_dev(offs), _dev(lens)), (offs, lens)):
        got = gpu.compute_batch(M, d, off_arg, len_arg, xor_out=M).cpu().numpy().view(np.uint32)
        bad = [i for i, w in enumerate(want) if int(got[i]) != w]
        assert not bad, (count, size, bad[:5])
