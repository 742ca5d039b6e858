"""bench.py --mode global: ONE suffix array of the whole N x SIZE text over N ranks (one per GPU).

A step = one collective dc3hip_global_build: all-gather of the text blocks, key-range split of every level, rank
exchange (all-to-all of pairs + all-gather of 4-byte blocks) over the transport.  value = total bytes x steps / (max
over ranks of the wall time between two barriers).  The text is generated per rank (each rank only its own block)."""
import os
import time

from .benchlib import HBM_PEAK_GBS, PATH_NAMES, KernelAcc, kernel_rooflines, path_roofline

MAX_N = 4278190080            # DC3HIP_MAX_N: positions are unsigned 32-bit on the device
XGMI_LINK_GBS = 153.0         # per direction and link (prompt / MI355X guide: 7 links x ~153 GB/s per GPU)


def global_total(total, kind):
    """(bytes of the one text, wide?, clipped?): beyond DC3HIP_MAX_N the library switches to 64-bit positions (wide mode) —
    high-entropy inputs only, so the low-entropy text generator is clipped to the 32-bit limit instead."""
    forced = any(k == "global_force_wide" and v != "0"            # "name" and "name=1" alike (dbg_on in the library)
                 for k, _, v in (tok.strip().partition("=") for tok in os.environ.get("DC3HIP_DEBUG", "").split(",")))
    wide = (total > MAX_N or forced) and kind != 2
    clipped = total > MAX_N and not wide
    return (MAX_N if clipped else total), wide, clipped


def make_rank(ss, dist, backend, world, rank, local_rank, max_total):
    """This process's rank of the group: RCCL (backend nccl: one rank per GPU) or the host-staged transport over the gloo
    group (DC3HIP_BENCH_BACKEND=gloo: several ranks on one GPU — the plumbing test on 1-GPU boxes, which says so in
    `interconnect.transport`).  With backend nccl there is NO fallback: if the library's RCCL communicator cannot be
    created on every rank, every rank raises (bench.py then reports the sacapart leg alone and says why there is no
    global number) — unless DC3HIP_BENCH_ALLOW_HOST_FALLBACK=1 explicitly asks for the host-staged transport, and then
    the line names it."""
    import sys
    if backend == "nccl":
        import torch
        g = None
        err = ""
        try:
            uid = [ss.GlobalRank.rccl_unique_id() if rank == 0 else None]
        except Exception as e:          # rank 0 could not even load RCCL: the others must not wait for an id
            uid, err = [None], repr(e)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                g = ss.GlobalRank.rccl(uid[0], rank, world, local_rank, max_total)
            except Exception as e:
                err = repr(e)
        ok = torch.tensor([1 if g is not None else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            return g
        if g is not None:
            g.close()
        if os.environ.get("DC3HIP_BENCH_ALLOW_HOST_FALLBACK") != "1":
            msg = (f"the library's RCCL communicator could not be created ({err or 'on another rank'}); not falling back to a "
                   "host-staged transport (DC3HIP_BENCH_ALLOW_HOST_FALLBACK=1 would)")
            print(f"bench_global: rank {rank}: {msg}", file=sys.stderr, flush=True)
            raise RuntimeError(msg)          # (every rank raises: the decision was all-reduced)
        if rank == 0:
            print(f"bench_global: RCCL transport unavailable ({err or 'on another rank'}); DC3HIP_BENCH_ALLOW_HOST_FALLBACK=1: "
                  "using the host-staged transport over gloo", flush=True)
        grp = dist.new_group(backend="gloo")

        class _GroupDist:               # the subset of torch.distributed the callbacks use, bound to the gloo group
            P2POp = dist.P2POp
            isend = staticmethod(dist.isend); irecv = staticmethod(dist.irecv)

            @staticmethod
            def batch_isend_irecv(ops):
                return dist.batch_isend_irecv([dist.P2POp(o.op, o.tensor, o.peer, grp) for o in ops])

            @staticmethod
            def all_gather(outs, t):
                return dist.all_gather(outs, t, group=grp)
        return ss.GlobalRank.torch_host(_GroupDist, rank, world, local_rank, max_total)
    return ss.GlobalRank.torch_host(dist, rank, world, local_rank, max_total)


def run_global(args, ss, dist, backend, world, rank, local_rank, per_gpu, kind, barrier, G=None):
    import numpy as np
    import torch

    total, wide, clipped = global_total(per_gpu * world, kind)
    if G is None:
        G = make_rank(ss, dist, backend, world, rank, local_rank, total)
    G.generate(total, args.seed, kind)
    for _ in range(max(args.warmup, 1)):
        G.build()
    chk0 = G.shard_checksum()

    barrier()
    t0 = time.perf_counter()
    acc = None
    kacc = KernelAcc()
    comm_ms = 0.0; comm_in = 0; comm_out = 0
    for _ in range(args.steps):
        G.build()
        st = G.stats()
        kacc.add(st["ctx"])
        comm_ms += st["comm_ms"]; comm_in += st["comm_bytes_in"]; comm_out += st["comm_bytes_out"]
        acc = st
    kernel_ms = kacc.build_ms
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    verify = {}
    assert G.shard_checksum() == chk0, "shard changed between identical builds"
    verify["idempotent_checksum"] = True
    if wide and not args.no_verify:
        # no single-device array exists at this size: the library's collective verifier (range + strict suffix order
        # over the whole array, across rank boundaries)
        verify["global_sufcheck"] = G.sufcheck()
        assert verify["global_sufcheck"] == 0, "global SA rejected by the collective verifier"
    # the shards tile [0, n) and their checksums add up to the checksum of a single-device build of the same text
    info = [None] * world
    dist.all_gather_object(info, {"rank": rank, "first": acc["shard_first"], "count": acc["shard_count"], "chk": chk0,
                                  "comm_ms": comm_ms / args.steps, "in": comm_in // args.steps, "out": comm_out // args.steps,
                                  "device_ms": kernel_ms / args.steps})
    dump = os.environ.get("DC3HIP_BENCH_DUMP_SA")
    if dump:
        first, sa = G.shard_sa(np.int64)
        np.save(os.path.join(dump, f"gshard_{rank}.npy"), sa)
    transport = G.transport()
    ctx_stats = acc["ctx"]
    G.close()

    out = None
    if rank == 0:
        info.sort(key=lambda d: d["rank"])
        nxt = 0
        for d in info:
            assert d["first"] == nxt, f"shards do not tile the suffix array: rank {d['rank']} starts at {d['first']}, expected {nxt}"
            nxt += d["count"]
        assert nxt == total
        verify["shards_tile_0_n"] = True
        gsum = sum(d["chk"] for d in info) & (2**64 - 1)
        if not args.no_verify and not wide:
            # single-device build of the whole text on this rank's GPU (untimed): sufcheck + checksum equality
            try:
                with ss.Context(total, device=local_rank) as c1:
                    c1.generate(total, args.seed, kind)
                    c1.build()
                    verify["single_device_sufcheck"] = c1.sufcheck()
                    verify["equal_single_device_checksum"] = bool(c1.checksum() == gsum)
                    assert verify["single_device_sufcheck"] == 0 and verify["equal_single_device_checksum"], "global SA differs from the single-device SA"
            except ss.Dc3HipError as e:
                verify["single_device_reference"] = f"skipped: {e}"
        value = total * args.steps / dt / 1e6
        roof, _ = kernel_rooflines(kacc, args.steps, ctx_stats, max(kernel_ms, 1e-9))
        worst = max(info, key=lambda d: d["comm_ms"])
        out = {
            "metric": "MB/s of input indexed (SA build), 1 GiB bytes, 1/2/4/8 GPUs",
            "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if not clipped else "weak up to DC3HIP_MAX_N, then strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{total / 2**30:g} GiB {args.kind} bytes over {world} GPUs (splitmix64 seed {args.seed}), ONE suffix array, "
                                   f"{'64-bit' if wide else 'u32'} positions, text blocks and SA shards resident in HBM",
                       "total_bytes": total, "bytes_per_gpu": total // world,
                       "partitioning": f"global SA, {transport}: text blocks (len/{world}+1 bytes) all-gathered, every level split by key "
                                       "range, rank exchange = all-to-all of (destination, rank) pairs + all-gather of 4-byte blocks; "
                                       "result sharded by suffix rank",
                       "clipped_to_DC3HIP_MAX_N": clipped},
            "path": {"text_order": acc["text_order"], "taken": PATH_NAMES.get(ctx_stats.get("text_sort_state", 0), "?"),
                     "levels": acc["levels"], "local_from_level": acc["local_from_level"], "exchanges_per_step": acc["exchanges"]},
            "value_MiBps": total * args.steps / dt / 2**20,
            "roofline": roof, "roofline_path": path_roofline(ctx_stats, kernel_ms / args.steps),
            "interconnect": {"transport": transport,
                             "bytes_in_per_rank_per_step": [d["in"] for d in info], "bytes_out_per_rank_per_step": [d["out"] for d in info],
                             "comm_ms_per_step": [round(d["comm_ms"], 3) for d in info],
                             "comm_ms": round(worst["comm_ms"], 3),
                             "device_ms_per_step": [round(d["device_ms"], 3) for d in info],
                             "achieved_GBps_in_slowest_rank": worst["in"] / max(worst["comm_ms"], 1e-9) / 1e6,
                             "peak_GBps_in": XGMI_LINK_GBS * min(world - 1, 7),
                             "note": "comm_ms is host wall time inside the (synchronous) collectives and includes waiting for the slowest rank"},
            "shards": [{"rank": d["rank"], "first": d["first"], "count": d["count"]} for d in info],
            "verify": verify, "hbm_peak_GBps": HBM_PEAK_GBS,
        }
    return out
