"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact.

Mirrors the reference's own tests (sift-persist/src/crc.rs:66-131) and widens
them to every size class and alignment the kernels distinguish.
"""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SEED = 0x51F7


def _rand(n, seed=SEED):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)


def test_reference_known_answers(gpu):
    # crc.rs:71-96, :121-122, :127-128
    with open(os.path.join(GOLDEN, "crc_kat.json")) as f:
        kat = json.load(f)
    for case in kat["reference_kat"]:
        data = case["data"].encode()
        want = int(case["crc"], 16)
        assert gpu.crc32(data) == want, case
        assert gpu.crc32(torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8)[: len(data)].cuda()) == want


def test_test_check_format(gpu):
    # crc.rs:119-123
    assert "0x{:08x}".format(gpu.crc32(b"123456789")) == "0xcbf43926"


def test_golden_random(gpu):
    with open(os.path.join(GOLDEN, "crc_random.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        buf = _rand(case["len"], case["seed"])
        t = torch.from_numpy(buf).cuda()
        assert gpu.compute(int(case["state"], 16), t) == int(case["compute"], 16), case


# Size classes: tiny (<16 B, one lane), one partial tile, one full tile
# (16 KiB), several one-tile groups, first multi-tile case (y table), ragged
# multi-tile, and sizes straddling every boundary by +-1.
SIZES = sorted(set(
    list(range(0, 70)) +
    [127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097,
     16383, 16384, 16385, 16400, 32767, 32768, 32769, 65536, 100000,
     (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 3 * (1 << 20) + 12345,
     256 * 16384 - 1, 256 * 16384, 256 * 16384 + 1, 257 * 16384 + 5,
     512 * 16384, 512 * 16384 + 16, 5 * 256 * 16384 + 777, (64 << 20) + 3]))


def test_sizes_vs_oracle(gpu, oracle):
    big = _rand(max(SIZES) + 64)
    dbig = torch.from_numpy(big).cuda()
    for n in SIZES:
        for state in (0xFFFFFFFF, 0, 0x12345678):
            want = oracle.compute(state, big[:n])
            got = gpu.compute(state, dbig[:n])
            assert got == want, (n, hex(state), hex(got), hex(want))


def test_alignments_vs_oracle(gpu, oracle):
    big = _rand(3 * (1 << 20) + 256, seed=7)
    dbig = torch.from_numpy(big).cuda()
    for off in list(range(0, 33)) + [63, 64, 65, 255]:
        for n in (0, 1, 5, 15, 16, 17, 31, 32, 33, 100, 16384, 16385, 70001, 2 * (1 << 20) + 3,
                  3 * (1 << 20)):
            want = oracle.compute(0xFFFFFFFF, big[off:off + n])
            got = gpu.compute(0xFFFFFFFF, dbig[off:off + n])
            assert got == want, (off, n, hex(got), hex(want))


def test_host_buffer_path(gpu, oracle):
    # pageable input: up to 512 KiB the calling thread copies it into a pinned slot, from
    # there on host threads fill a ring of 8 MiB pinned slots (pieces of 1, 2, 4, 8 MiB)
    for n in (0, 1, 9, 31, 32, 4096, 100000, (512 << 10) - 1, 512 << 10, (512 << 10) + 1, (1 << 20) + 3,
              (3 << 20) - 7, (4 << 20) - 1, 4 << 20, (8 << 20) - 1, 8 << 20, (8 << 20) + 1, (16 << 20) + 5,
              (32 << 20) - 1, (32 << 20), (32 << 20) + 1, (70 << 20) + 13):
        buf = _rand(n, seed=n + 1)
        assert gpu.compute(0xFFFFFFFF, buf) == oracle.compute(0xFFFFFFFF, buf), n
        assert gpu.compute(0xFFFFFFFF, buf.tobytes()) == zlib.crc32(buf.tobytes()) ^ 0xFFFFFFFF
        assert gpu.compute(0x13579BDF, buf[1:]) == oracle.compute(0x13579BDF, buf[1:]), n  # odd start, other register
    # the staged-copy path that remains for 0 copy threads is chosen per device at first use:
    # covered by tools/host_input_sweep.py 0 (a process of its own), not here


def test_host_input_random_sizes_vs_zlib(gpu):
    # both sides of the 512 KiB switch between the caller-copied pinned slot and the
    # threaded ring, odd starts, chained registers
    rng = np.random.default_rng(4242)
    big = rng.integers(0, 256, size=3 << 20, dtype=np.uint8)
    for k in range(200):
        n = int(rng.integers(0, 1 << int(rng.integers(1, 22))))
        off = int(rng.integers(0, big.size - n))
        piece = big[off:off + n]
        cut = int(rng.integers(0, n + 1))
        state = gpu.compute(0xFFFFFFFF, piece[:cut])
        state = gpu.compute(state, piece[cut:])
        assert state == zlib.crc32(piece.tobytes()) ^ 0xFFFFFFFF, (k, off, n, cut)


def test_synchronous_calls_from_many_threads(gpu, oracle):
    # The synchronous entry points poll a pinned 8-byte slot (tag | result) that belongs to the
    # calling thread: 16 threads, device-resident and host-resident inputs, 100 calls each.
    import threading
    bufs = [_rand(3000 + 977 * i, seed=200 + i) for i in range(16)]
    want = [oracle.compute(0xFFFFFFFF, b) for b in bufs]
    dbufs = [torch.from_numpy(b).cuda() for b in bufs]
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            for k in range(100):
                got = gpu.compute(0xFFFFFFFF, dbufs[i] if k & 1 else bufs[i])
                if got != want[i]:
                    errors.append((i, k, hex(got), hex(want[i])))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_streaming_equals_one_shot(gpu, oracle):
    # Digest::update chaining (crc.rs:45-47; container.rs:81-82 feeds type then data)
    buf = _rand(1 << 20, seed=3)
    d = gpu.Digest()
    cuts = [0, 4, 4 + 1000, 4 + 1000 + 16384, 500000, 1 << 20]
    for a, b in zip(cuts, cuts[1:]):
        d.update(buf[a:b])
    assert d.finalize() == oracle.crc32(buf)
    d2 = gpu.Digest()
    t = torch.from_numpy(buf).cuda()
    for a, b in zip(cuts, cuts[1:]):
        d2.update(t[a:b])
    assert d2.finalize() == oracle.crc32(buf)


def test_async_chain_on_stream(gpu, oracle):
    buf = _rand((5 << 20) + 77, seed=11)
    t = torch.from_numpy(buf).cuda()
    state = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.zeros(gpu.workspace_bytes(), dtype=torch.uint8, device="cuda")
    cuts = [0, 13, 1 << 20, (3 << 20) + 5, buf.size]
    gpu.compute_async(0xFFFFFFFF, t[cuts[0]:cuts[1]], state, ws)
    for a, b in zip(cuts[1:], cuts[2:]):
        gpu.compute_async(state, t[a:b], state, ws)
    torch.cuda.synchronize()
    assert (int(state.item()) & 0xFFFFFFFF) == oracle.compute(0xFFFFFFFF, buf)


def test_linearity_property(gpu):
    # size-independent property at a large size: crc is affine over GF(2):
    # compute(0, a^b) == compute(0, a) ^ compute(0, b)
    n = (200 << 20) + 5
    g = torch.Generator(device="cuda").manual_seed(SEED)
    a = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    assert gpu.compute(0, a ^ b) == gpu.compute(0, a) ^ gpu.compute(0, b)


def test_split_concat_property_full_size(gpu):
    # BASELINE size (1 GiB, the container.rs:49 cap): whole == stitched halves
    # at arbitrary cut points, and equals zlib on the host copy.
    n = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(SEED)
    t = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    whole = gpu.compute(0xFFFFFFFF, t)
    for cut in (1, 12345, (1 << 29) + 7, n - 3):
        sa = gpu.compute(0xFFFFFFFF, t[:cut])
        sb = gpu.compute(0, t[cut:])
        assert gpu.concat(sa, sb, n - cut) == whole, cut
        assert gpu.compute(sa, t[cut:]) == whole, cut
    assert whole ^ 0xFFFFFFFF == zlib.crc32(t.cpu().numpy().tobytes())


def test_two_launch_form_matches(gpu, oracle):
    # The A/B measurement path (separate combine kernel) must give the same bits.
    from sift_amd import _lib
    big = _rand((40 << 20) + 99, seed=5)
    d = torch.from_numpy(big).cuda()
    try:
        _lib.check(_lib.lib.sift_tune(_lib.TUNE_SINGLE_LAUNCH, 0))
        for n in (16385, 300000, (8 << 20) + 1, big.size):
            assert gpu.compute(0xFFFFFFFF, d[3:3 + n - 3]) == oracle.compute(0xFFFFFFFF, big[3:n]), n
    finally:
        _lib.check(_lib.lib.sift_tune(_lib.TUNE_SINGLE_LAUNCH, 1))


def test_back_to_back_calls_share_workspace(gpu, oracle):
    # Single-launch finish: residues and the arrival counter live in one
    # workspace reused by consecutive calls on a stream.  Different buffers
    # and sizes back to back, no host sync in between, every result checked
    # (a stale residue or a counter left non-zero would show here).
    rng = np.random.default_rng(99)
    bufs = [rng.integers(0, 256, size=int(n), dtype=np.uint8)
            for n in rng.integers(20000, 6 << 20, size=24)]
    dbufs = [torch.from_numpy(b).cuda() for b in bufs]
    ws = torch.zeros(gpu.workspace_bytes(), dtype=torch.uint8, device="cuda")
    outs = torch.zeros(len(bufs) * 5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for rep in range(5):
        for i, d in enumerate(dbufs):
            gpu.compute_async(0xFFFFFFFF, d, outs[rep * len(bufs) + i: rep * len(bufs) + i + 1], ws)
    torch.cuda.synchronize()
    got = [int(v) & 0xFFFFFFFF for v in outs.cpu().tolist()]
    want = [oracle.compute(0xFFFFFFFF, b) for b in bufs]
    assert got == want * 5


def test_beyond_4gib_offsets(gpu):
    # 64-bit lengths: 5 GiB + 3 bytes, cut on both sides of the 4 GiB mark
    n = (5 << 30) + 3
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(SEED)
    for a in range(0, n, 1 << 30):
        b = min(n, a + (1 << 30))
        t[a:b] = torch.randint(0, 256, (b - a,), dtype=torch.uint8, device="cuda", generator=g)
    whole = gpu.compute(0xFFFFFFFF, t)
    for cut in ((4 << 30) - 1, (4 << 30) + 17):
        sa = gpu.compute(0xFFFFFFFF, t[:cut])
        assert gpu.compute(sa, t[cut:]) == whole, cut
        assert gpu.concat(sa, gpu.compute(0, t[cut:]), n - cut) == whole, cut
    h = t.cpu().numpy()
    del t
    crc = 0
    for a in range(0, n, 1 << 30):
        crc = zlib.crc32(h[a:a + (1 << 30)], crc)
    assert whole ^ 0xFFFFFFFF == crc


def test_graph_capture_and_replay(gpu, oracle):
    # The device entry point allocates nothing and never synchronises, so it can be
    # captured into a HIP graph and replayed on new data in the same buffers.
    buf = _rand((6 << 20) + 123, seed=21)
    d = torch.from_numpy(buf).cuda()
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = torch.zeros(gpu.workspace_bytes(), dtype=torch.uint8, device="cuda")
    gpu.compute_async(0xFFFFFFFF, d[7:], out[0:1], ws)  # first-use initialisation outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gpu.compute_async(0xFFFFFFFF, d[7:], out[0:1], ws)
        gpu.compute_async(out[0:1], d[:1000], out[1:2], ws)
    for seed in (22, 23, 24):
        nb = _rand(buf.size, seed=seed)
        d.copy_(torch.from_numpy(nb).cuda())
        graph.replay()
        torch.cuda.synchronize()
        got = [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()]
        s0 = oracle.compute(0xFFFFFFFF, nb[7:])
        assert got == [s0, oracle.compute(s0, nb[:1000])], seed


def test_pinned_host_input_is_streamed_in_place(gpu, oracle):
    # hipHostMalloc'ed input: no staging copy, the kernel reads host memory directly
    n = (40 << 20) + 7
    t = torch.empty(n, dtype=torch.uint8).pin_memory()
    buf = _rand(n, seed=31)
    t.numpy()[:] = buf
    for a, b in ((0, n), (3, n - 5), (4097, 4097 + 5000)):
        assert gpu.compute(0xFFFFFFFF, t.numpy()[a:b]) == oracle.compute(0xFFFFFFFF, buf[a:b])


def test_two_threads_two_streams(gpu, oracle):
    # Calls in flight at the same time on one device: each thread has its own
    # stream and its own zeroed workspace, as include/sift_amd.h requires.
    import threading
    bufs = [_rand((9 << 20) + 11 * i, seed=40 + i) for i in range(2)]
    want = [oracle.compute(0xFFFFFFFF, b) for b in bufs]
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d = torch.from_numpy(bufs[i]).cuda()
                ws = torch.zeros(gpu.workspace_bytes(), dtype=torch.uint8, device="cuda")
                out = torch.zeros(200, dtype=torch.int32, device="cuda")
                stream.synchronize()
                for k in range(200):
                    gpu.compute_async(0xFFFFFFFF, d, out[k:k + 1], ws)
                stream.synchronize()
                got = {int(v) & 0xFFFFFFFF for v in out.cpu().tolist()}
            if got != {want[i]}:
                errors.append((i, [hex(g) for g in got], hex(want[i])))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_null_workspace_many_streams_in_flight(gpu, oracle):
    # workspace=NULL: the library keeps one workspace per stream, so calls in flight
    # on different streams never share an accumulator, however many there are.
    # 8 streams x 200 calls, nothing synchronised until the end.
    nstreams, ncalls = 8, 200
    bufs = [_rand((3 << 20) + 4099 * i + 1, seed=60 + i) for i in range(nstreams)]
    want = [oracle.compute(0xFFFFFFFF, b) for b in bufs]
    dbufs = [torch.from_numpy(b).cuda() for b in bufs]
    outs = [torch.zeros(ncalls, dtype=torch.int32, device="cuda") for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize()
    for k in range(ncalls):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                gpu.compute_async(0xFFFFFFFF, dbufs[i], outs[i][k:k + 1])   # no workspace
    torch.cuda.synchronize()
    for i in range(nstreams):
        got = {int(v) & 0xFFFFFFFF for v in outs[i].cpu().tolist()}
        assert got == {want[i]}, (i, [hex(g) for g in got], hex(want[i]))


def test_null_workspace_is_refused_under_capture_and_reset_works(gpu, oracle):
    from sift_amd import _lib
    buf = _rand((2 << 20) + 5, seed=77)
    d = torch.from_numpy(buf).cuda()
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu.compute_async(0xFFFFFFFF, d, out)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        with pytest.raises(_lib.SiftError) as e:
            gpu.compute_async(0xFFFFFFFF, d, out)          # library scratch cannot go into a graph
        ws = None
    assert e.value.status == _lib.ERR_INVALID_ARG and "captured" in str(e.value)
    del graph
    _lib.check(_lib.lib.sift_reset_workspace(0))
    assert gpu.compute(0xFFFFFFFF, d) == oracle.compute(0xFFFFFFFF, buf)


def test_random_differential(gpu):
    # 400 random (offset, length, state) triples across every geometry the host
    # picks (1..256 groups, 1..10 tiles per group, ragged first tiles, tails),
    # against zlib.crc32 -- an implementation independent of both oracle and product.
    rng = np.random.default_rng(0x51F7 + 99)
    big = rng.integers(0, 256, size=(48 << 20) + 4096, dtype=np.uint8)
    d = torch.from_numpy(big).cuda()
    hb = big.tobytes()
    M = 0xFFFFFFFF
    for i in range(400):
        kind = i % 4
        if kind == 0:
            n = int(rng.integers(32, 70000))
        elif kind == 1:
            n = int(rng.integers(70000, 5 << 20))
        elif kind == 2:
            n = int(rng.integers(1, 2600)) * 16384 + int(rng.integers(-40, 41))  # near tile multiples
        else:
            n = int(rng.integers(5 << 20, 44 << 20))
        off = int(rng.integers(0, 4096))
        state = int(rng.integers(0, 1 << 32))
        want = zlib.crc32(hb[off:off + n], state ^ M) ^ M
        got = gpu.compute(state, d[off:off + n])
        assert got == want, (i, off, n, hex(state), hex(got), hex(want))
