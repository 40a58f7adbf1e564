"""GPU parity for sift_crc32_compute_device_static, the entry point that
allocates nothing (its workspace is a static device array of the library).

Checked through the C ABI against the existing entry point it is a variant of
(sift_crc32_compute_device with its allocated workspace ring), bit for bit, and
against the oracle, on calls in a row over the SAME scratch: big, then smaller,
then tiny -- what the last call left in the array must not leak into the next.
Collected last (file name) so that a failure here hides no other test.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _static(L, state, state_t, buf, out):
    dev = buf.device.index
    L.check(L.lib.sift_crc32_compute_device_static(
        dev, state & 0xFFFFFFFF, state_t.data_ptr() if state_t is not None else None,
        buf.data_ptr(), buf.numel(), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))


def _existing(L, state, state_t, buf, out):
    dev = buf.device.index
    L.check(L.lib.sift_crc32_compute_device(
        dev, state & 0xFFFFFFFF, state_t.data_ptr() if state_t is not None else None,
        buf.data_ptr(), buf.numel(), out.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream))


def test_static_scratch_matches_existing_entry_point(gpu, oracle):
    from sift_amd import _lib as L
    rng = np.random.default_rng(0x51F7 + 7)
    big = rng.integers(0, 256, size=(40 << 20) + 3, dtype=np.uint8)
    d = torch.from_numpy(big).cuda()
    # two calls in a row, the second with a smaller input; then ever smaller ones,
    # down to the one-tile, one-group and serial (< 32 B) paths, and back up
    sizes = [big.size, (1 << 20) + 5, 300001, 16385, 16384, 100, 31, 0, (7 << 20) + 1, 4096]
    got_s = torch.zeros(len(sizes), dtype=torch.int32, device="cuda")
    got_e = torch.zeros(len(sizes), dtype=torch.int32, device="cuda")
    for i, n in enumerate(sizes):           # back to back on one stream, no sync between
        _static(L, 0xFFFFFFFF, None, d[3:3 + n] if n else d[:0], got_s[i:i + 1])
    for i, n in enumerate(sizes):
        _existing(L, 0xFFFFFFFF, None, d[3:3 + n] if n else d[:0], got_e[i:i + 1])
    torch.cuda.synchronize()
    s = [int(v) & 0xFFFFFFFF for v in got_s.cpu().tolist()]
    e = [int(v) & 0xFFFFFFFF for v in got_e.cpu().tolist()]
    want = [oracle.compute(0xFFFFFFFF, big[3:3 + n]) for n in sizes]
    assert s == e, [(n, hex(a), hex(b)) for n, a, b in zip(sizes, s, e) if a != b]
    assert s == want

    # state chained through device memory (Digest::update on a stream), same scratch
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    cuts = [0, 13, 9 << 20, (9 << 20) + 70000, big.size]
    _static(L, 0xFFFFFFFF, None, d[cuts[0]:cuts[1]], st)
    for a, b in zip(cuts[1:], cuts[2:]):
        _static(L, 0, st, d[a:b], st)
    torch.cuda.synchronize()
    assert (int(st.item()) & 0xFFFFFFFF) == oracle.compute(0xFFFFFFFF, big)


def test_static_scratch_many_geometries_in_a_row(gpu, oracle):
    # 40 calls of different geometry back to back over the one array: a single
    # leaked residue or arrival ticket would make every later result wrong.
    from sift_amd import _lib as L
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, size=(12 << 20) + 99, dtype=np.uint8)
    d = torch.from_numpy(big).cuda()
    sizes = [int(x) for x in rng.integers(20000, big.size, size=40)]
    out = torch.zeros(len(sizes), dtype=torch.int32, device="cuda")
    for i, n in enumerate(sizes):
        _static(L, 0x12345678, None, d[:n], out[i:i + 1])
    torch.cuda.synchronize()
    got = [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()]
    assert got == [oracle.compute(0x12345678, big[:n]) for n in sizes]


def test_static_rejects_what_it_cannot_do(gpu):
    from sift_amd import _lib as L
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    # beyond the stated maximum: refused before anything is launched
    rc = L.lib.sift_crc32_compute_device_static(0, 0, None, d.data_ptr(), L.STATIC_MAX_LEN + 1,
                                                out.data_ptr(), s)
    assert rc == L.ERR_INVALID_ARG and b"SIFT_STATIC_MAX_LEN" in L.lib.sift_last_error()
    # a device that was never initialised: refused (setting it up would allocate)
    rc = L.lib.sift_crc32_compute_device_static(torch.cuda.device_count() + 3, 0, None, d.data_ptr(), 64,
                                                out.data_ptr(), s)
    assert rc == L.ERR_INVALID_ARG and b"sift_init" in L.lib.sift_last_error()


def test_static_under_graph_capture(gpu, oracle):
    # the caller this entry point is for: nothing in it allocates or synchronises
    buf = np.random.default_rng(21).integers(0, 256, size=(6 << 20) + 123, dtype=np.uint8)
    from sift_amd import _lib as L
    d = torch.from_numpy(buf).cuda()
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _static(L, 0xFFFFFFFF, None, d[7:], out[0:1])
        _static(L, 0, out[0:1], d[:1000], out[1:2])
    for seed in (22, 23):
        nb = np.random.default_rng(seed).integers(0, 256, size=buf.size, dtype=np.uint8)
        d.copy_(torch.from_numpy(nb).cuda())
        graph.replay()
        torch.cuda.synchronize()
        got = [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()]
        s0 = oracle.compute(0xFFFFFFFF, nb[7:])
        assert got == [s0, oracle.compute(s0, nb[:1000])], seed
