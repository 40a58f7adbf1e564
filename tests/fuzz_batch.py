#!/usr/bin/env python3
"""Randomised differential run of the batch entry points against the C oracle
(test infrastructure; this script is a checker, not product): random streams of
serialized chunks and save files with random corruptions, batch sizes on both
sides of every path switch, descriptors on the host or in HBM.
usage: python tests/fuzz_batch.py [rounds] [first seed]; tests/test_batch_gpu.py runs a few rounds"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import sift_amd  # noqa: F401
from oracle import c_oracle as oracle
from sift_amd import container, crc

SIZES = [0, 1, 3, 20, 27, 28, 29, 60, 500, 4070, 4076, 4080, 4090, 4096, 5000, 32760, 32764, 32768, 40000,
         262130, 262140, 262144, 300000, 1 << 20, (2 << 20) + 5]


def pick_size(rng):
    r = rng.random()
    if r < 0.6:
        return int(rng.integers(0, 3000))
    if r < 0.9:
        return int(SIZES[int(rng.integers(0, len(SIZES)))])
    return int(rng.integers(0, 600000))


def one_round(seed):
    rng = np.random.default_rng(seed)
    counts = [1, 2, 7, 63, 64, 65, 200, 1000, 1024, 1025, 1536, 1537, 3072, 3073, 4096, 4097, 6000]
    n = int(counts[int(rng.integers(0, len(counts)))])
    budget = 48 << 20
    stream = bytearray()
    offs, avail, want = [], [], []
    docs = bool(rng.integers(0, 2))
    for i in range(n):
        size = pick_size(rng) if len(stream) < budget else int(rng.integers(0, 200))
        data = rng.integers(0, 256, size=size, dtype=np.uint8)
        ser = bytearray(oracle.write_document(data) if docs else oracle.write_chunk(b"AMRG", data))
        kind = int(rng.integers(0, 10))
        if kind == 0 and len(ser) > 0:
            ser[int(rng.integers(0, len(ser)))] ^= 1 << int(rng.integers(0, 8))
        cut = int(rng.integers(0, min(len(ser), 40) + 1)) if kind == 1 else 0
        pad = int(rng.integers(0, 17))
        stream += bytes(pad)
        offs.append(len(stream))
        avail.append(len(ser) - cut + (int(rng.integers(0, 9)) if kind == 2 else 0))
        stream += ser + bytes(8)
        view = bytes(stream[offs[-1]:offs[-1] + avail[-1]])
        if docs:
            st, _, _, ln, _ = oracle.read_document(view)
            want.append((st, ln if st == 0 else None))
        else:
            st, _, _, ln, _ = oracle.read_chunk(view)
            want.append((st, ln if st == 0 else 0))
    d = torch.from_numpy(np.frombuffer(bytes(stream), dtype=np.uint8).copy()).cuda()
    where = int(rng.integers(0, 2))
    o = torch.tensor(offs, dtype=torch.int64, device="cuda") if where else offs
    a = torch.tensor(avail, dtype=torch.int64, device="cuda") if where else avail
    fn = container.read_documents_device if docs else container.read_chunks_device
    status, dlen = fn(d, o, a)
    status, dlen = status.cpu().tolist(), dlen.cpu().tolist()
    for i in range(n):
        assert status[i] == want[i][0], (seed, "docs" if docs else "chunks", n, where, i, status[i], want[i], avail[i])
        if want[i][1] is not None:
            assert dlen[i] == want[i][1], (seed, n, where, i, dlen[i], want[i])
    # Chunk::write's crc step (container.rs:75) on the chunks that read clean: their crc
    # fields zeroed, filled in again by the device, the stream must be what it was
    if not docs:
        clean = [i for i in range(n) if status[i] == 0]
        if clean:
            d2 = d.clone()
            for i in clean:
                at = offs[i] + 8 + want[i][1]
                d2[at:at + 4] = 0
            co = [offs[i] for i in clean]
            ca = [avail[i] for i in clean]
            st3 = container.write_chunk_crcs_device(
                d2, torch.tensor(co, dtype=torch.int64, device="cuda") if where else co,
                torch.tensor(ca, dtype=torch.int64, device="cuda") if where else ca)
            assert set(st3.cpu().tolist()) == {0}, (seed, "write status", n, where)
            assert bool((d2 == d).all().item()), (seed, "write bytes", n, where)
    # plain buffers over the same bytes
    lens = [min(av, int(rng.integers(0, av + 1)) if rng.random() < 0.5 else av) for av in avail]
    got = crc.compute_batch(0xFFFFFFFF, d, o if where else offs,
                            torch.tensor(lens, dtype=torch.int64, device="cuda") if where else lens).cpu().numpy().view(np.uint32)
    blob = np.frombuffer(bytes(stream), dtype=np.uint8)
    for i in rng.integers(0, n, size=min(n, 40)):
        i = int(i)
        assert int(got[i]) == oracle.compute(0xFFFFFFFF, blob[offs[i]:offs[i] + lens[i]]), (seed, "crc", n, where, i)
    return n, docs, where


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    for s in range(first, first + rounds):
        n, docs, where = one_round(s)
        print(f"seed {s}: {n} {'documents' if docs else 'chunks'}, descriptors {'in HBM' if where else 'on the host'}: ok", flush=True)
    print("fuzz ok")


if __name__ == "__main__":
    main()
