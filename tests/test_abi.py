"""CPU: the C-ABI library loads, exports every symbol include/sift_amd.h
declares, and its host-only entry points (no GPU needed) behave like the
reference.  No compute call is made here."""
import ctypes as C
import os
import re
import struct
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from sift_amd import _lib
    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "sift_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sift_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(L):
    declared = _declared()
    assert declared, "header parse found nothing"
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True,
                        check=True).stdout
    exported = set(re.findall(r" T (sift_[a-z0-9_]+)", nm))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(L.EXPORTS), (sorted(set(declared) ^ set(L.EXPORTS)))
    for name in declared:
        getattr(L.lib, name)


def test_no_cpu_fallback_without_device(L):
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    out = C.c_uint32(0)
    buf = b"123456789"
    assert L.lib.sift_crc32_compute_host(0, 0xFFFFFFFF, buf, len(buf), C.byref(out)) == L.ERR_NO_DEVICE
    assert L.lib.sift_init(0) == L.ERR_NO_DEVICE
    assert b"device" in L.lib.sift_strerror(L.ERR_NO_DEVICE)
    with pytest.raises(L.SiftError):
        L.check(L.lib.sift_init(0))


def test_product_does_not_import_oracle():
    # oracle/ is test infrastructure; nothing under sift_amd/, include/ or the measurement
    # tools may reference it (checkers that need it live under tests/)
    for base in ("sift_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".sh", "Makefile")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    hit = re.search(r"import\s+oracle|from\s+oracle|oracle/|sift_oracle|liboracle|c_oracle|py_oracle",
                                    text)
                    assert hit is None, (os.path.join(dirpath, fn), hit.group(0))


def test_concat_matches_oracle(L, oracle):
    import numpy as np
    rng = np.random.default_rng(3)
    for _ in range(100):
        a = rng.integers(0, 256, size=int(rng.integers(0, 400)), dtype=np.uint8).tobytes()
        b = rng.integers(0, 256, size=int(rng.integers(0, 70000)), dtype=np.uint8).tobytes()
        s = int(rng.integers(0, 1 << 32))
        sa, sb = oracle.compute(s, a), oracle.compute(0, b)
        assert L.lib.sift_crc32_concat(sa, sb, len(b)) == oracle.compute(s, a + b)


def test_gf2_field_facts(L):
    one = 0x80000000  # the polynomial 1 in the reflected representation (gf2.h)
    # x^(8n) for n = 0 is 1; P is primitive so x^(2^32-1) = 1, i.e. 8n = 8(2^32-1) bits -> 1
    assert L.lib.sift_crc32_concat(one, 0, 0) == one
    assert L.lib.sift_crc32_concat(one, 0, (1 << 32) - 1) == one
    # one zero byte advances the register by x^8: matches the table step on state 1
    assert L.lib.sift_crc32_concat(one, 0, 1) == 0x00800000
    # linear in both arguments
    assert L.lib.sift_crc32_concat(0x1234, 0x5678, 77) ^ L.lib.sift_crc32_concat(0x1234, 0, 77) == 0x5678


def test_header_entry_points(L):
    buf = (C.c_uint8 * 16)()
    assert L.lib.sift_write_header(buf, 16) == 9
    assert bytes(buf[:9]) == b"\x89SIFT\r\n\x1a\n"            # container.rs:99,104
    assert L.lib.sift_write_header(buf, 8) == -L.ERR_WRITE
    assert L.lib.sift_read_header(buf, 16) == 9
    assert L.lib.sift_read_header(buf, 8) == -L.ERR_READ        # short read
    buf[4] ^= 1
    assert L.lib.sift_read_header(buf, 16) == -L.ERR_READ       # InvalidData
    assert L.lib.sift_expect_type(b"AMRG", b"AMRG") == L.OK     # container.rs:86-95
    assert L.lib.sift_expect_type(b"AMRG", b"SEND") == L.ERR_READ


def test_chunk_checks_that_precede_the_device(L):
    # Chunk::read rejects an oversize length before reading type or data (container.rs:48-52);
    # short inputs are Error::Read; Chunk::write rejects len > u32::MAX (container.rs:69-71).
    t = (C.c_uint8 * 4)()
    off, ln = C.c_uint64(0), C.c_uint32(0)

    def read(b):
        arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0")
        return L.lib.sift_read_chunk(0, arr, len(b), t, C.byref(off), C.byref(ln))

    assert read(struct.pack(">I", (1 << 30) + 1)) == -L.ERR_READ_INVALID_CHUNK_SIZE
    assert read(struct.pack(">I", 0xFFFFFFFF) + b"AMRG") == -L.ERR_READ_INVALID_CHUNK_SIZE
    assert read(b"") == -L.ERR_READ
    assert read(b"\0\0\0") == -L.ERR_READ
    assert read(struct.pack(">I", 5) + b"AM") == -L.ERR_READ
    assert read(struct.pack(">I", 5) + b"AMRGabc") == -L.ERR_READ
    assert read(struct.pack(">I", 5) + b"AMRGabcde\0\0\0") == -L.ERR_READ  # crc cut short
    out = (C.c_uint8 * 32)()
    assert L.lib.sift_write_chunk(0, b"AMRG", out, 1 << 32, out, 32) == -L.ERR_WRITE_INVALID_CHUNK_SIZE
    assert L.lib.sift_write_chunk(0, b"AMRG", out, 40, out, 32) == -L.ERR_WRITE


def test_python_mirror_shape():
    # the host-side mirror keeps the reference's names (crc.rs:36-53, container.rs:35-136)
    import sift_amd
    from sift_amd import container
    for name in ("Digest", "compute"):
        assert hasattr(sift_amd, name)
    for name in ("Chunk", "Error", "read_header", "write_header", "read_chunk", "write_chunk", "HEADER"):
        assert hasattr(container, name)
    assert {"new", "update", "finalize"} - {"new"} <= set(dir(sift_amd.Digest))


def test_header_is_plain_c_and_links(L, tmp_path):
    exe = tmp_path / "abi_link"
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wno-pedantic",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_link.c"),
                    "-L", libdir, "-lsift_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    env = dict(os.environ)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + torch_lib + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_link ok" in r.stdout
    # the C file lists every declared symbol
    src = open(os.path.join(ROOT, "tests", "abi_link.c")).read()
    for name in _declared():
        assert name in src, name
