"""world_size 2 over gloo: the N>1 path of bench.py — sacapart chunk ownership per rank (no data-path
collective), position-addressable generator, max-over-ranks timing reduce.  Without a GPU (this
container) the ranks compute their chunk's SA with the ORACLE, so only the host logic is under test; the
`-m gpu` variant runs the same two processes with the PRODUCT (libdc3hip through the C ABI, both ranks on
GPU 0) and the parent compares every chunk with the oracle."""
import os
import socket
import sys

import multiprocessing as mp

import numpy as np
import pytest

from conftest import ROOT

# torch is imported by the spawned rank processes only (each a fresh process: torch first, then the library); the pytest
# process itself stays without it on a GPU box (conftest.py: one HIP runtime per process)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total_len, seed, q, use_product=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import Oracle
    from stringsearch_amd.partition import rank_chunk
    o = Oracle()
    off, n = rank_chunk(total_len, world, rank)
    # each rank generates ONLY its chunk of the global stream (offset-addressable generator)
    full = o.gen(total_len, seed, 0)
    chunk = full[off:off + n]
    if use_product:
        import stringsearch_amd as ss
        assert ss.device_count() >= 1, "product variant needs a device (no CPU fallback)"
        sa = ss.sort(chunk).into_parts()[1]
    else:
        sa = o.sufsort(chunk)
    # whole-job time = max over ranks
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64))
    dist.barrier()
    q.put((rank, off, n, sa.tolist(), float(t.item()), [int(s.item()) for s in sizes]))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_rank_sacapart_gloo_product(oracle):
    _two_rank(oracle, True)


@pytest.mark.timeout(120)
def test_two_rank_sacapart_gloo(oracle):
    _two_rank(oracle, False)


def _two_rank(oracle, use_product):
    import stringsearch_amd as ss
    from stringsearch_amd.partition import chunk_bounds
    world, total_len, seed = 2, 20001, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_len, seed, q, use_product)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=250 if use_product else 100) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    text = oracle.gen(total_len, seed, 0)
    bounds = chunk_bounds(total_len, world)
    assert [(r[1], r[2]) for r in res] == bounds == [(0, 10001), (10001, 10000)]
    assert all(abs(r[4] - 0.002) < 1e-12 for r in res)                 # MAX over ranks
    assert all(r[5] == [10001, 10000] for r in res)
    # stitched result == PartitionedSuffixArray built in one process
    part = ss.PartitionedSuffixArray(text, world, lambda c: ss.SuffixArray(c, oracle.sufsort(c)))
    for r, sa in zip(res, part.sas):
        assert r[3] == sa.into_parts()[1].tolist()


def _transport_worker(rank, world, port, q):
    import torch  # noqa: F401
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    from stringsearch_amd.global_sa import torch_host_callbacks
    a2a, ag = torch_host_callbacks(dist, rank, world)
    U = ctypes.c_uint64 * world
    # all_to_all_v: rank r sends (r+1)*(d+1) bytes of value 16*r+d to rank d
    sb = [(rank + 1) * (d + 1) for d in range(world)]
    so = [sum(sb[:d]) for d in range(world)]
    send = np.concatenate([np.full(sb[d], 16 * rank + d, dtype=np.uint8) for d in range(world)])
    rb = [(r + 1) * (rank + 1) for r in range(world)]
    ro = [sum(rb[:r]) for r in range(world)]
    recv = np.zeros(sum(rb), dtype=np.uint8)
    assert a2a(None, send.ctypes.data, U(*so), U(*sb), recv.ctypes.data, U(*ro), U(*rb)) == 0
    want = np.concatenate([np.full(rb[r], 16 * r + rank, dtype=np.uint8) for r in range(world)])
    ok1 = bool(np.array_equal(recv, want))
    # all_gather_v with ragged blocks (rank r contributes 3r+1 bytes of value r+1), one of them empty-capable
    gb = [3 * r + 1 for r in range(world)]
    go = [sum(gb[:r]) for r in range(world)]
    mine = np.full(gb[rank], rank + 1, dtype=np.uint8)
    out = np.zeros(sum(gb), dtype=np.uint8)
    assert ag(None, mine.ctypes.data, gb[rank], out.ctypes.data, U(*go), U(*gb)) == 0
    ok2 = bool(np.array_equal(out, np.concatenate([np.full(gb[r], r + 1, dtype=np.uint8) for r in range(world)])))
    q.put((rank, ok1, ok2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_global_mode_host_transport_callbacks_gloo():
    """The host-staged transport of the global mode (dc3hip_host_transport callbacks over torch.distributed) on CPU,
    world size 2: ragged all_to_all_v and all_gather_v deliver every byte to the right place."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transport_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=100) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res == [(0, True, True), (1, True, True)]


def test_global_mode_block_arithmetic():
    """Text blocks of the global mode = sacapart's chunking (lib.rs:43-46): they tile [0, n), never overlap, and agree
    with chunk_bounds wherever par_chunks produces a chunk."""
    from stringsearch_amd.global_sa import block_of
    from stringsearch_amd.partition import chunk_bounds
    for n in (0, 1, 2, 5, 7, 100, 20001, 2**30, 2**32 - 2**24):
        for P in (1, 2, 3, 4, 8, 16):
            blocks = [block_of(n, P, r) for r in range(P)]
            assert sum(l for _, l in blocks) == n
            nxt = 0
            for off, ln in blocks:
                assert off == nxt or ln == 0
                nxt += ln
            assert [b for b in blocks if b[1] > 0] == chunk_bounds(n, P)
