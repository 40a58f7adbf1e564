"""Stream identity and scratch lifetime of the library's own scratch (workspace = NULL
and the batch entry points), through the C ABI.  No reference counterpart: the
reference never leaves the CPU; results are checked against the oracle / zlib."""
import ctypes as C
import threading
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HIP_STREAM_PER_THREAD = 2  # hipStreamPerThread: one handle value, a different stream in every thread
HIP_STREAM_LEGACY = 1


def _hip():
    """The HIP runtime this process has loaded (torch's)."""
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert paths, "no HIP runtime mapped"
    h = C.CDLL(sorted(paths)[0])
    h.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    h.hipStreamDestroy.argtypes = [C.c_void_p]
    h.hipStreamSynchronize.argtypes = [C.c_void_p]
    h.hipSetDevice.argtypes = [C.c_int]
    h.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    h.hipHostFree.argtypes = [C.c_void_p]
    return h


def test_two_threads_on_hipStreamPerThread_do_not_share_scratch(gpu, oracle):
    """scratch keyed by the raw handle made two threads on hipStreamPerThread
    run on one exchange block.  200 overlapping calls per thread, NULL workspace."""
    from sift_amd import _lib as L
    hip = _hip()
    rng = np.random.default_rng(0x51F7 + 31)
    host = rng.integers(0, 256, size=(24 << 20) + 77, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    # sizes that launch many workgroups (the exchange words are what would be shared)
    sizes = [(5 << 20) + 3, (1 << 20) + 1, (9 << 20) + 17, 300001]
    want = {n: oracle.compute(0xFFFFFFFF, host[7:7 + n]) for n in sizes}
    calls = 200
    outs = [torch.zeros(calls, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(2)

    def worker(t):
        try:
            assert hip.hipSetDevice(0) == 0
            start.wait()
            for i in range(calls):
                n = sizes[(i + t) % len(sizes)]
                L.check(L.lib.sift_crc32_compute_device(
                    0, 0xFFFFFFFF, None, d.data_ptr() + 7, n, outs[t].data_ptr() + 4 * i, None,
                    HIP_STREAM_PER_THREAD))
            assert hip.hipStreamSynchronize(HIP_STREAM_PER_THREAD) == 0
            L.check(L.lib.sift_release_stream(0, HIP_STREAM_PER_THREAD))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for t in range(2):
        got = [int(v) & 0xFFFFFFFF for v in outs[t].cpu().tolist()]
        exp = [want[sizes[(i + t) % len(sizes)]] for i in range(calls)]
        assert got == exp, [(i, hex(a), hex(b)) for i, (a, b) in enumerate(zip(got, exp)) if a != b][:5]


def test_null_and_legacy_stream_are_one_stream(gpu):
    from sift_amd import _lib as L
    host = np.random.default_rng(5).integers(0, 256, size=(3 << 20) + 5, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib.sift_crc32_compute_device(0, 0xFFFFFFFF, None, d.data_ptr(), d.numel(), out.data_ptr(), None, None))
    L.check(L.lib.sift_crc32_compute_device(0, 0xFFFFFFFF, None, d.data_ptr(), d.numel(), out.data_ptr() + 4, None,
                                            HIP_STREAM_LEGACY))
    torch.cuda.synchronize()
    want = zlib.crc32(host.tobytes()) ^ 0xFFFFFFFF
    assert [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()] == [want, want]


def test_release_stream_keeps_device_memory_flat(gpu):
    """1,000 streams created, used (compute with NULL workspace and a batch call that
    allocates the stream's batch scratch and pinned descriptor buffers), released and
    destroyed: device memory in use must not grow with the number of streams."""
    from sift_amd import _lib as L
    hip = _hip()
    host = np.random.default_rng(9).integers(0, 256, size=1 << 20, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    out = torch.zeros(128, dtype=torch.int32, device="cuda")
    offs = (np.arange(100, dtype=np.uint64) * 8192).astype(np.uint64)
    lens = np.full(100, 5000, dtype=np.uint64)
    want_whole = zlib.crc32(host.tobytes()) ^ 0xFFFFFFFF
    want_items = [zlib.crc32(host[int(o):int(o) + 5000].tobytes()) for o in offs]

    def one(check):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0  # hipStreamNonBlocking
        L.check(L.lib.sift_crc32_compute_device(0, 0xFFFFFFFF, None, d.data_ptr(), d.numel(), out.data_ptr(), None, s))
        L.check(L.lib.sift_crc32_batch_device(0, 0xFFFFFFFF, 0xFFFFFFFF, d.data_ptr(), offs.ctypes.data,
                                              lens.ctypes.data, 100, out.data_ptr() + 4, s))
        L.check(L.lib.sift_release_stream(0, s))  # waits for the stream
        if check:
            got = [int(v) & 0xFFFFFFFF for v in out[:101].cpu().tolist()]
            assert got[0] == want_whole and got[1:] == want_items
        assert hip.hipStreamDestroy(s) == 0

    for _ in range(20):
        one(True)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(1000):
        one(i % 100 == 0)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # without the release each stream would keep > 0.5 MiB of device memory (0.5 GiB in all)
    assert free0 - free1 < (64 << 20), (free0, free1)
    # a stream the library holds nothing for is not an error
    L.check(L.lib.sift_release_stream(0, None))
    L.check(L.lib.sift_release_stream(0, None))


def test_misaligned_workspace_is_refused(gpu):
    from sift_amd import _lib as L
    import sift_amd
    d = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.zeros(sift_amd.workspace_bytes() + 512, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert ws.data_ptr() % 256 == 0
    for skew in (4, 8, 64, 128):
        rc = L.lib.sift_crc32_compute_device(0, 0xFFFFFFFF, None, d.data_ptr(), d.numel(), out.data_ptr(),
                                             ws.data_ptr() + skew, s)
        assert rc == L.ERR_INVALID_ARG
        rc = L.lib.sift_chunk_crc_device(0, d.data_ptr(), 1000, out.data_ptr(), ws.data_ptr() + skew, s)
        assert rc == L.ERR_INVALID_ARG
    L.check(L.lib.sift_crc32_compute_device(0, 0xFFFFFFFF, None, d.data_ptr(), d.numel(), out.data_ptr(),
                                            ws.data_ptr() + 256, s))
    torch.cuda.synchronize()
    assert int(out.item()) & 0xFFFFFFFF == zlib.crc32(bytes(1 << 20)) ^ 0xFFFFFFFF


def test_pinned_descriptors_may_be_reused_when_the_call_returns(gpu):
    """pinned descriptor arrays were read in place, asynchronously.  Up to 1 MiB
    of host descriptors is now copied before the call returns, pinned or not."""
    from sift_amd import _lib as L
    hip = _hip()
    n = 3000
    host = np.random.default_rng(11).integers(0, 256, size=n * 700 + 64, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), 16 * n, 0) == 0
    try:
        desc = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(2 * n,))
        offs = np.arange(n, dtype=np.uint64) * 700 + 3
        lens = 100 + (np.arange(n, dtype=np.uint64) * 7) % 590
        # a long kernel in front, so that the batch kernel has certainly not started
        # reading when the arrays are overwritten
        big = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(4):
            L.check(L.lib.sift_crc32_compute_device(0, 0, None, big.data_ptr(), big.numel(), out.data_ptr(), None, s))
        desc[:n] = offs
        desc[n:] = lens
        L.check(L.lib.sift_crc32_batch_device(0, 0xFFFFFFFF, 0xFFFFFFFF, d.data_ptr(), p.value, p.value + 8 * n, n,
                                              out.data_ptr(), s))
        desc[:] = 0xFFFFFFFFFFFFFF00  # the caller reuses its arrays at once
        torch.cuda.synchronize()
        got = [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()]
        want = [zlib.crc32(host[int(o):int(o + l)].tobytes()) for o, l in zip(offs, lens)]
        assert got == want
    finally:
        hip.hipHostFree(p)
