"""world_size 2 over gloo: the N>1 path of bench.py — sacapart chunk ownership per rank (no data-path
collective), position-addressable generator, max-over-ranks timing reduce.  Without a GPU (this
container) the ranks compute their chunk's SA with the ORACLE, so only the host logic is under test; the
`-m gpu` variant runs the same two processes with the PRODUCT (libdc3hip through the C ABI, both ranks on
GPU 0) and the parent compares every chunk with the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total_len, seed, q, use_product=False):
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
