#!/usr/bin/env python3
"""bench.py — MB/s of input indexed (suffix-array build) on MI355X, the metric of BASELINE.json.

A "step" = one device-resident suffix-array build of this rank's share of the text (text already in HBM,
SA left in HBM).  Two multi-GPU semantics:

  sacapart   crates/sacapart/src/lib.rs:39-58: the N x SIZE byte text is cut into chunks of len/N + 1 bytes, rank c
             builds the independent local SA of chunk c — no data-path collective, weak scaling.
  global     ONE suffix array of the whole N x SIZE text, sharded over the ranks by suffix rank; text blocks are
             all-gathered and sample ranks exchanged over RCCL (xGMI); see DESIGN.md §6.

--mode auto (default): N = 1 is the single-GPU build.  N > 1 runs BOTH legs, K timed steps each, and prints ONE line:
`value` = the global mode (what north_star asks for: one true SA, RCCL rank exchange; `interconnect` has its bytes and
time), `sacapart` = the reference's own partitioned semantics beside it, `cpu_baseline` = sacapart's parallel story on
the host (P chunks of len/P + 1 bytes on P pinned threads, one divsufsort() each).  An N > 1 run starts with a transport
self-test (ragged all-to-all / all-gather of known bytes through the library's RCCL communicator, every byte checked,
ncclCommCount compared with N) and exits non-zero if it fails — a run never reports a host-staged number as xGMI.

Rank 0 prints ONE JSON line — compact (< 4 KB, strict JSON: metric, value, roofline, cpu_baseline, verify, the recursion-only
and per-config figures as a few numbers each); every block in full (all kernel families, the loopback table, per-level
traces) goes to the side file named in its `detail` field (gpurun_out/bench_detail.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size BYTES] [--kind random|dna|text] [--mode auto|sacapart|global]
                    [--cpu-sample-mib M] [--no-cpu] [--no-verify] [--no-extras]
    N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
             --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stringsearch_amd.benchlib import (HBM_PEAK_GBS, KINDS, PATH_NAMES, KernelAcc, build_block, compact_line, kernel_rooflines,  # noqa: E402
                                        parse_size, path_roofline, write_detail)


def host_cpu_model():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return "unknown"


def cpu_baseline(text_u8, sample_bytes):
    """The reference's CPU path (libdivsufsort built from /root/reference into oracle/_ref) or, if that
    is absent, our C restatement of crates/dc3 — timed on one host core like divsuftest's measure()
    (crates/divsuftest/src/main.rs:145-151: wall clock around the call incl. the SA allocation)."""
    import numpy as np
    n = len(text_u8)
    sample = text_u8 if sample_bytes >= n else np.ascontiguousarray(text_u8[:sample_bytes])
    ref = os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so")
    port = os.path.join(ROOT, "oracle", "liboracle_dc3.so")
    if os.path.exists(ref):
        L = ctypes.CDLL(ref); f = L.divsufsort; kind = "reference"
    elif os.path.exists(port):
        L = ctypes.CDLL(port); f = L.dc3_oracle_sufsort_i32; kind = "port"
    else:
        return None
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]; f.restype = ctypes.c_int32
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(old_aff)[0]})
    except Exception:
        pass
    t0 = time.perf_counter()
    sa = np.zeros(len(sample), dtype=np.int32)
    rc = f(sample.ctypes.data, sa.ctypes.data, len(sample))
    dt = time.perf_counter() - t0
    if old_aff:
        try:
            os.sched_setaffinity(0, old_aff)
        except Exception:
            pass
    assert rc == 0
    what = "the whole buffer" if len(sample) == n else f"first {len(sample) / 2**20:.0f} MiB of the same buffer"
    return {"value": len(sample) / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": f"{what}, one divsufsort() call, wall clock incl. SA allocation ({dt:.2f} s)",
            "seconds": dt, "host_cpu": host_cpu_model(), "host_cores_available": os.cpu_count()}, sa


def cpu_baseline_partitions(ss, P, total_len, sample_bytes, seed, kind, device):
    """sacapart's parallel story on the host (crates/sacapart/src/lib.rs:12-14,43-49): the P chunks of len/P + 1 bytes,
    one divsufsort() per chunk, P host threads pinned to P different cores — on a bounded sample (the first
    `sample_bytes` of every chunk) so that the run stays within minutes.  The chunk texts come from the same device
    generator as the GPU legs."""
    import threading
    import numpy as np
    from stringsearch_amd.partition import rank_chunk
    ref = os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so")
    port = os.path.join(ROOT, "oracle", "liboracle_dc3.so")
    if os.path.exists(ref):
        L = ctypes.CDLL(ref); f = L.divsufsort; kindname = "reference"
    elif os.path.exists(port):
        L = ctypes.CDLL(port); f = L.dc3_oracle_sufsort_i32; kindname = "port"
    else:
        return None
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]; f.restype = ctypes.c_int32
    texts = []
    for c in range(P):
        off, ln = rank_chunk(total_len, P, c)
        m = min(ln, sample_bytes)
        with ss.Context(m, device=device) as cx:
            cx.generate(m, seed, kind, offset=off)
            texts.append(cx.text())
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        cores = list(range(os.cpu_count() or 1))
    secs = [0.0] * P
    rcs = [0] * P

    def work(c):
        try:
            os.sched_setaffinity(0, {cores[c % len(cores)]})       # (Linux: pid 0 = the calling thread)
        except Exception:
            pass
        t = time.perf_counter()
        sa = np.zeros(len(texts[c]), dtype=np.int32)              # incl. the SA allocation, as divsuftest's measure()
        rcs[c] = f(texts[c].ctypes.data, sa.ctypes.data, len(texts[c]))
        secs[c] = time.perf_counter() - t
    th = [threading.Thread(target=work, args=(c,)) for c in range(P)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    assert all(r == 0 for r in rcs), rcs
    nbytes = sum(len(t) for t in texts)
    whole = all(len(texts[c]) == rank_chunk(total_len, P, c)[1] for c in range(P))
    return {"value": nbytes / wall / 1e6, "unit": "MB/s", "cores": min(P, len(cores)), "kind": kindname,
            "sample": (f"sacapart on the host: {P} chunks of len/{P}+1 bytes, " + ("whole chunks" if whole else f"first {sample_bytes >> 20} MiB of every chunk")
                       + f", one divsufsort() per chunk on {min(P, len(cores))} pinned threads, wall clock incl. SA allocation ({wall:.2f} s)"),
            "seconds": wall, "per_thread_seconds": [round(x, 3) for x in secs], "host_cpu": host_cpu_model(),
            "host_cores_available": len(cores)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process.  The child's stdout (the one
    JSON line of rank 0) and stderr pass through; the child's exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher around --gpus %d: starting %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=str, default="1GiB", help="bytes per GPU")
    ap.add_argument("--kind", type=str, default="random", choices=list(KINDS))
    ap.add_argument("--mode", type=str, default="auto", choices=["auto", "sacapart", "global"])
    ap.add_argument("--global-timeout", type=int, default=420, help="N > 1, --mode auto: seconds the global leg may take before the "
                    "line is printed with the sacapart leg alone (and the failure stated)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample-mib", type=int, default=0,
                    help="CPU baseline sample in MiB of the same buffer; 0 = N = 1: the whole buffer (1 GiB is ~60-80 s of one-core "
                         "divsufsort and doubles as the bit-exact comparison); N > 1: 256 MiB of every chunk")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs that are not part of `value`: recursion-only build, text/DNA configs, end-to-end FFI")
    ap.add_argument("--no-recursion-line", action="store_true", help="(kept for old command lines) same as --no-extras")
    ap.add_argument("--detail", type=str, default=None, help="where the full record goes (default gpurun_out/bench_detail.json)")
    ap.add_argument("--dump-stats", type=str, default=None, help="write the last build's dc3hip_stats as JSON here")
    args = ap.parse_args()
    if args.no_recursion_line:
        args.no_extras = True

    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the one-process-per-GPU job ourselves, as a fresh CHILD process (this
        # process has not touched the GPU yet and never will; no exec), relay its one JSON line and its exit code.
        # crates/sacapart/src/lib.rs:39-58 parallelises inside one call; this is the same convenience for the bench.
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    # One GPU: nothing of torch is needed, and nothing of it is imported — the library then runs on the HIP runtime it was
    # compiled against (`hip_runtime.match`), not on the older one a PyTorch wheel maps first (profiles/r05_crash_hunt.md).
    # N > 1: torch.distributed is the rendezvous (and must be imported BEFORE the library is loaded: one runtime per process).
    use_dist = world > 1
    torch = dist = None
    backend = os.environ.get("DC3HIP_BENCH_BACKEND", "nccl")
    if use_dist:
        import torch
        import torch.distributed as dist
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
        # one rank per GPU; DC3HIP_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing test on 1-GPU boxes)
        local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import stringsearch_amd as ss

    if not use_dist:
        ndev = ss.device_count()
        assert ndev > 0, "bench.py needs a GPU (there is no CPU path): " + ss.last_error()
        local_rank = local_rank % ndev

    per_gpu = parse_size(args.size)
    kind = KINDS[args.kind]

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        ss.device_synchronize(local_rank)       # hipDeviceSynchronize through the C ABI (all streams of this rank's device)

    def emit(full):
        """rank 0's ONE line: the compact object (< 4 KB, strict JSON); every block in full goes to the side file it names"""
        full["hip_runtime"] = ss.hip_versions()
        print(compact_line(full, write_detail(full, args.detail)), flush=True)

    # ---- N > 1: the library's own transport — created strictly (no silent fallback) and self-tested.  --mode auto runs the
    # sacapart leg FIRST and everything of the global leg (communicator, self-test, builds) afterwards under one watchdog: a
    # transport that cannot be had, fails its self-test or hangs costs the run its global number, never its line.
    G = None
    selftest = None
    do_global = world > 1 and args.mode in ("auto", "global")
    do_sacapart = world == 1 or args.mode in ("auto", "sacapart")

    class TransportRefused(RuntimeError):
        pass

    def global_setup():
        """(G, selftest record): this rank of the library's group over the job's transport, after the transport self-test.
        Raises on every rank when RCCL cannot be had (make_rank all-reduces the decision); TransportRefused = a run that must
        not be labelled RCCL over xGMI."""
        from stringsearch_amd.bench_global import make_rank, global_total
        gtotal, gwide, gclipped = global_total(per_gpu * world, kind)
        g = make_rank(ss, dist, backend, world, rank, local_rank, gtotal)      # raises on every rank if RCCL cannot be had
        seen = g.selftest()
        if backend == "nccl" and seen != world:
            raise TransportRefused(f"the RCCL communicator reports {seen} ranks, the job has {world}: refusing to call this run RCCL over xGMI")
        rccl_binding = None
        if backend == "nccl":
            # one RCCL per process: the library must have bound to the librccl torch already mapped (same file in
            # /proc/self/maps, and the only librccl mapped)
            path, pre = ss.GlobalRank.rccl_library()
            mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
            rccl_binding = {"library": path, "was_already_mapped_by_host_program": pre, "librccl_files_mapped": mapped}
            if not (len(mapped) == 1 and os.path.realpath(mapped[0]) == os.path.realpath(path)):
                raise TransportRefused(f"libdc3hip bound to {path} but the process maps {mapped}: two RCCL instances in one process")
        st = {"passed": True, "transport": g.transport(), "ranks_seen_by_transport": seen, "world_size": world, "rccl": rccl_binding,
              "what": "ragged all_to_all_v + ragged all_gather_v + host all-gather of known bytes through the library's communicator, every byte checked on every rank"}
        return g, st

    if do_global and not do_sacapart:
        try:
            G, selftest = global_setup()
        except BaseException as e:              # noqa: BLE001 - --mode global has nothing else to report
            print(f"bench.py: rank {rank}: no global leg: {e!r}", file=sys.stderr, flush=True)
            os._exit(4)
    if not do_sacapart:
        from stringsearch_amd.bench_global import run_global
        out = run_global(args, ss, dist, backend, world, rank, local_rank, per_gpu, kind, barrier, G=G)
        if rank == 0:
            out["transport_selftest"] = selftest
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit(out)
        return

    fresh = None
    if world == 1 and not args.no_extras:
        # What divsuftest's measure() would see (crates/divsuftest/src/main.rs:145-151: ONE un-warmed call in a fresh process,
        # the array a fresh `vec![0; n]`): a child process that loads the library, calls dc3hip_sufsort_i32 once and reports
        # (tools/first_call_probe.py).  It runs BEFORE this process touches the device: memory a process has just freed is
        # wiped by the driver before it is handed out again (about 30 ms per GiB), and a first call behind this bench's
        # own 45 GiB contexts would mostly measure that.  Never part of `value`.
        try:
            import subprocess
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_call_probe.py"), "untouched", str(per_gpu)],
                               capture_output=True, text=True, timeout=300)
            fresh = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:                # (reported, never fatal)
            fresh = {"error": repr(e)}

    from stringsearch_amd.partition import rank_chunk
    total_len = per_gpu * world
    off, n = (0, total_len) if world == 1 else rank_chunk(total_len, world, rank)   # sacapart/src/lib.rs:43-46

    ctx = ss.Context(n, device=local_rank)
    ctx.generate(n, args.seed, kind, offset=off)

    for _ in range(args.warmup):
        ctx.build()
    verify = {}
    chk0 = None
    if not args.no_verify:
        if args.warmup == 0:
            ctx.build()
        rc = ctx.sufcheck()                  # full-size GPU sufcheck (utils.c:160-241 semantics)
        assert rc == 0, f"rank {rank}: SA failed sufcheck ({rc}) — no throughput reported"
        verify["sufcheck_full"] = rc
        chk0 = ctx.checksum()

    barrier()
    t0 = time.perf_counter()
    kacc = KernelAcc()
    for _ in range(args.steps):
        ctx.build()
        kacc.add(ctx.stats())
    kernel_ms = kacc.build_ms
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if not args.no_verify:
        assert ctx.checksum() == chk0, "SA changed between identical builds"
        verify["idempotent_checksum"] = True
    dump = os.environ.get("DC3HIP_BENCH_DUMP_SA")     # tests: every rank leaves its chunk + SA for an oracle comparison
    if dump:
        np.save(os.path.join(dump, f"chunk_{rank}.npy"), ctx.text())
        np.save(os.path.join(dump, f"sa_{rank}.npy"), ctx.sa())

    st = ctx.stats()
    out = None
    if rank == 0:
        value = total_len * args.steps / dt / 1e6
        # `roofline` = the kernel family with the largest share of the build time; all families are kept beside it
        roof, roof_all = kernel_rooflines(kacc, args.steps, st, kernel_ms)
        out = {
            "metric": "MB/s of input indexed (SA build), 1 GiB bytes, 1/2/4/8 GPUs",
            "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{per_gpu / 2**30:g} GiB {args.kind} bytes per GPU (splitmix64 seed {args.seed}), "
                                   f"i32 SA, DC3 HIP, text and SA resident in HBM",
                       "bytes_per_gpu": n, "total_bytes": total_len,
                       "partitioning": "single SA" if world == 1 else f"sacapart: {world} chunks of len/{world}+1 bytes, one per GPU, no collective"},
            "path": {"text_sort_state": st.get("text_sort_state", 0), "taken": PATH_NAMES.get(st.get("text_sort_state", 0), "?"),
                     "levels": st["levels"],
                     "note": "`value` is produced by this path; the DC3 recursion on the same input is `dc3_recursion_only`, "
                             "low-entropy text and DNA are `per_config`"},
            "value_MiBps": total_len * args.steps / dt / 2**20,     # the reference prints binary units (divsuftest main.rs:179-183)
            "roofline": roof, "roofline_kernels": roof_all, "roofline_path": path_roofline(st, kernel_ms / args.steps),
            "msd": {k: st.get(k) for k in ("msd_sorts", "msd_fallbacks", "msd_max_subbucket")},
            # the speed of the bucket ordering's partition passes rests on "blocks with equal blockIdx % 8 share an XCD":
            # probed when the context is created and counted block by block in the timed builds (HW_REG_XCC_ID)
            "xcd_grouping_effective": bool(st.get("xcd_round_robin") == 1 and (st.get("xcd_blocks", 0) == 0 or st.get("xcd_group_hit", 0) >= 0.9)),
            "xcd_grouping": {k: st.get(k) for k in ("xcd_round_robin", "xcd_blocks", "xcd_group_hit")},
            "verify": verify, "arena_peak_GB": st["arena_peak"] / 1e9,
        }
        text = None
        if world == 1 and not args.no_cpu:
            text = ctx.text()
            sample_bytes = n if args.cpu_sample_mib <= 0 else min(n, args.cpu_sample_mib << 20)
            res = cpu_baseline(text, sample_bytes)
            if res is not None:
                cb, cpu_sa = res
                out["cpu_baseline"] = cb
                if sample_bytes == n and not args.no_verify:
                    # the north_star target: SA[0..n) bit-exact against the reference's divsufsort on the same buffer
                    out["verify"]["equal_cpu_reference"] = bool(np.array_equal(cpu_sa, ctx.sa()))
                    out["verify"]["cpu_reference"] = cb["kind"]
                    assert out["verify"]["equal_cpu_reference"], "GPU suffix array differs from the CPU reference"
                del cpu_sa
        if world == 1 and not args.no_extras:
            # ---- end-to-end FFI time (SURVEY §8d "(ii)"): dc3hip_sufsort_i32 on host buffers, H2D + build + D2H,
            # the figure comparable with the reference's measure() (divsuftest/src/main.rs:145-151). Never `value`.
            if text is None:
                text = ctx.text()
            sa_host = np.ones(n, dtype=np.int32)        # pre-touched, as a Vec reused by a caller would be
            ctx.close(); ctx = None                     # the one-shot call owns its (cached) context
            L = ss.lib()
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                rc = L.dc3hip_sufsort_i32(text.ctypes.data, sa_host.ctypes.data, n)
                ts.append(time.perf_counter() - t1)
                assert rc == 0, ss.last_error()
            warm = min(ts[1:])
            # `ms` = a warm call: the second call of the fresh child process when there is one (a host program that indexes
            # texts: 108 ms at the 56 GB/s most boxes copy at), beside the warm calls of THIS process — which by then has moved
            # some 40 GiB through pageable numpy arrays and whose copies the runtime has been seen to serve at half that rate
            # (190 ms; profiles/r06v_*) — both are reported.
            child_warm = (fresh or {}).get("second_call_ms")
            out["e2e_ffi"] = {"entry": "dc3hip_sufsort_i32(T, SA, n) on pageable host buffers",
                              "ms": child_warm if child_warm else warm * 1e3,
                              "MB/s": n / ((child_warm if child_warm else warm * 1e3) * 1e-3) / 1e6,
                              "ms_is": "second call of a fresh process" if child_warm else "warm call of the bench process",
                              "in_bench_process_ms": warm * 1e3,
                              "first_call_in_this_process_ms": ts[0] * 1e3,       # (behind the contexts this bench has just freed)
                              "first_call_fresh_process_ms": (fresh or {}).get("first_call_ms"), "fresh_process": fresh,
                              "includes": "H2D of n bytes, device build, D2H of 4n bytes; a first call also creates the cached context",
                              "pcie_floor_ms": 5.0 * n / 56e9 * 1e3}
            del sa_host
            ss.release_cache()
    if args.dump_stats and rank == 0:
        json.dump(st, open(args.dump_stats, "w"))
    if ctx is not None:
        ctx.close()
    if rank == 0 and world == 1 and not args.no_extras:
        if st.get("text_sort_state", 0) == 1:
            # The timed builds above finished in the whole-text shortcut (every 9-byte window distinct, no
            # recursion level built).  For reference, the same input through the DC3 recursion proper
            # (levels, tuples, merge): not part of `value`.
            ss.debug_set("no_text_shortcut", "1")
            try:
                with ss.Context(n, device=local_rank) as c2:
                    c2.generate(n, args.seed, kind, offset=off)
                    c2.build()
                    st2, m2, blk = build_block(ss, c2, min(args.steps, 3), "pmc_traffic_recursion.json")
                    out["dc3_recursion_only"] = {"switch": "DC3HIP_DEBUG=no_text_shortcut", "device_ms_per_step": m2,
                                                 "MBps": n / m2 / 1e3, "sufcheck": c2.sufcheck(),
                                                 "checksum_equal": (c2.checksum() == chk0) if chk0 is not None else None,
                                                 "levels": list(zip(st2["level_n"], st2["level_K"], st2["level_sorted"]))}
                    out["dc3_recursion_only"].update(blk)
            finally:
                ss.debug_unset("no_text_shortcut")
        # BASELINE.json configs[2] (low-entropy text) and the per-GPU class of configs[4] (DNA) at the same size:
        # device-resident builds, GPU sufcheck each; reported beside `value`, never part of it.
        per_cfg = {}
        with ss.Context(n, device=local_rank) as c3:
            for name, kd, seed in (("text", 2, 3), ("dna", 1, 5)):
                if kd == kind:
                    continue
                c3.generate(n, seed, kd)
                c3.build()
                st3, m, blk = build_block(ss, c3, 3, "pmc_traffic_text.json" if name == "text" else "pmc_traffic_dna.json")
                per_cfg[f"{name}_{per_gpu / 2**30:g}GiB"] = {
                    "ms": m, "MB/s": n / m / 1e3, "sufcheck": c3.sufcheck(), "levels": st3["levels"],
                    "path": PATH_NAMES.get(st3.get("text_sort_state", 0), "?"),
                    # levels ordered by the splitter (sample) ordering, and symbols per record where a level sorted a
                    # wider window than the triple (dc3_ssort.hip.hpp)
                    "splitter_ordering": {k: st3.get(k) for k in ("ssort_sorts", "ssort_fallbacks", "ssort_max_subbucket",
                                                                  "ssort_part_ms", "ssort_local_ms")},
                    "level_name_width": st3.get("level_name_width")}
                per_cfg[f"{name}_{per_gpu / 2**30:g}GiB"].update(blk)
                if st3.get("text_sort_state", 0) != 0:
                    # finished (or started) by the whole-text order: the DC3 recursion proper on the same text beside it
                    chk3 = c3.checksum()
                    ss.debug_set("no_text_shortcut", "1")
                    try:
                        with ss.Context(n, device=local_rank) as c5:
                            c5.generate(n, seed, kd)
                            c5.build(); c5.build()
                            per_cfg[f"{name}_{per_gpu / 2**30:g}GiB"]["dc3_recursion_only"] = {
                                "switch": "DC3HIP_DEBUG=no_text_shortcut", "ms": c5.stats()["build_ms"], "levels": c5.stats()["levels"],
                                "checksum_equal": c5.checksum() == chk3}
                    finally:
                        ss.debug_unset("no_text_shortcut")
        # the per-GPU chunk of BASELINE.json configs[4] (16 GiB DNA over 8 GPUs, sacapart chunks of 2 GiB + 1 byte, 64-bit
        # indices at the boundary): beyond 2^31 positions, device-resident, GPU sufcheck
        if per_gpu == 1 << 30 and not args.no_verify:
            nc = (1 << 31) + 1
            try:
                with ss.Context(nc, device=local_rank) as c6:
                    c6.generate(nc, 5, 1, offset=7 * nc)
                    c6.build()
                    ms = []
                    for _ in range(2):
                        c6.build(); ms.append(c6.stats()["build_ms"])
                    st6 = c6.stats()
                    m = sum(ms) / len(ms)
                    per_cfg["dna_2GiB_plus_1_chunk_of_configs4"] = {
                        "ms": m, "MB/s": nc / m / 1e3, "sufcheck": c6.sufcheck(), "levels": st6["levels"],
                        "path": PATH_NAMES.get(st6.get("text_sort_state", 0), "?"), "index_type": "u32 on the device, widened to i64 on delivery"}
            except ss.Dc3HipError as e:
                per_cfg["dna_2GiB_plus_1_chunk_of_configs4"] = {"skipped": str(e)}
        out["per_config"] = per_cfg
        # The GLOBAL mode (one suffix array over P ranks) as P loopback ranks on THIS GPU: the ranks
        # time-share the device, so wall_ms is about the SUM of all ranks' work (work_inflation = wall / single-device
        # build) — a correctness and total-work figure, not a scaling measurement.  Checked against the single-device
        # checksum of the same text.
        gl = []
        gn = min(n, 256 << 20)
        for gkind, gseed, gname in ((0, 2, "random"), (2, 3, "text")):
            with ss.Context(gn, device=local_rank) as c4:
                c4.generate(gn, gseed, gkind); c4.build()
                sms = []
                for _ in range(3):                       # (best of 3, as the loopback walls beside it)
                    c4.build(); sms.append(c4.stats()["build_ms"])
                single_ms = min(sms); single_chk = c4.checksum()
            for P in (2, 4, 8):
                # (a) the ranks' streams share the GPU freely: wall = about the sum of all ranks' work (as in rounds 3-4)
                with ss.LoopbackGroup(P, gn, device=local_rank) as g:
                    g.generate(gn, gseed, gkind)
                    g.build()
                    walls = []
                    for _ in range(3):
                        t1 = time.perf_counter(); g.build(); walls.append((time.perf_counter() - t1) * 1e3)
                    wall = min(walls)
                    gst = g.stats()
                    row = {"input": f"{gn >> 20} MiB {gname}", "ranks": P, "wall_ms": wall, "single_device_ms": single_ms,
                           "work_inflation": wall / single_ms, "checksum_equal_single_device": g.checksum() == single_chk,
                           "text_order": gst[0]["text_order"], "levels": gst[0]["levels"], "rank_exchanges": gst[0]["exchanges"],
                           "bytes_in_per_rank": [x["comm_bytes_in"] for x in gst],
                           "shard_counts": [x["shard_count"] for x in gst],
                           # (P streams share the device here: does the XCD-grouped reservation still find its XCD?)
                           "xcd_group_hit_per_rank": [round(x["ctx"]["xcd_group_hit"], 4) for x in gst],
                           "xcd_grouping_effective": all(x["ctx"]["xcd_round_robin"] == 1 and (x["ctx"]["xcd_blocks"] == 0 or x["ctx"]["xcd_group_hit"] >= 0.9) for x in gst)}
                # (b) what P GPUs would take: the ranks pass a device token (a rank's work_ms is then its own work, not its share
                # of a time-sliced GPU), the select-or-route policy decides as on xGMI (global_link_gbps=153: on a shared device
                # it would always select), and every collective is priced at the most bytes a rank exchanges with ONE peer /
                # 153 GB/s (one xGMI link); no overlap assumed
                with ss.debug_switches(global_link_gbps=153, global_device_token=1), ss.LoopbackGroup(P, gn, device=local_rank) as g:
                    g.generate(gn, gseed, gkind)
                    g.build()
                    best = None
                    for _ in range(3):
                        g.build()
                        st_ = g.stats()
                        pw = max(x["work_ms"] for x in st_) + max(x["link_ms"] for x in st_)
                        if best is None or pw < best[0]:
                            best = (pw, st_)
                    pred, gst = best
                    row.update({"predicted_wall_ms_on_P_gpus": pred, "predicted_MBps_on_P_gpus": gn / pred / 1e3,
                                "predicted_speedup_over_one_gpu": single_ms / pred,
                                "work_ms_per_rank": [round(x["work_ms"], 3) for x in gst], "link_ms_per_rank": [round(x["link_ms"], 3) for x in gst],
                                "collectives": gst[0]["collectives"], "selecting_pass1_as_on_xgmi": gst[0]["select_p1"],
                                "prediction_checksum_equal_single_device": g.checksum() == single_chk})
                    assert row["prediction_checksum_equal_single_device"], "global-mode shards (routed as on xGMI) differ from the single-device suffix array"
                gl.append(row)
                assert row["checksum_equal_single_device"], "global-mode shards differ from the single-device suffix array"
        out["global_mode_loopback"] = gl
        # One suffix array of MORE than 2^32 positions (what BASELINE.json configs[3] / configs[4] need; 64-bit positions,
        # "wide" global contexts): 2^32 + 2^20 + 3 random bytes over two loopback ranks time-sharing this GPU,
        # verified by the library's collective checker (no single-device array exists at this size).
        if per_gpu == 1 << 30 and not args.no_verify:
            wn = (1 << 32) + (1 << 20) + 3
            try:
                with ss.LoopbackGroup(2, wn, device=local_rank) as g:
                    g.generate(wn, 6, 0)
                    g.build()
                    t1 = time.perf_counter(); g.build(); wall = (time.perf_counter() - t1) * 1e3
                    gst = g.stats()
                    out["global_mode_beyond_2pow32"] = {
                        "input": f"{wn} random bytes", "ranks": 2, "transport": "loopback (both ranks on this GPU: wall = sum of their work)",
                        "wall_ms": wall, "MB/s_of_total_work": wn / wall / 1e3, "global_sufcheck": g.sufcheck(),
                        "shards_tile_0_n": gst[0]["shard_first"] == 0 and gst[1]["shard_first"] == gst[0]["shard_count"]
                                           and gst[0]["shard_count"] + gst[1]["shard_count"] == wn,
                        "positions": "64-bit", "bytes_in_per_rank": [x["comm_bytes_in"] for x in gst]}
                    assert out["global_mode_beyond_2pow32"]["global_sufcheck"] == 0 and out["global_mode_beyond_2pow32"]["shards_tile_0_n"]
            except ss.Dc3HipError as e:
                out["global_mode_beyond_2pow32"] = {"skipped": str(e)}
    if world > 1:
        sac = out                                   # rank 0: the sacapart leg's line; others: None
        if rank == 0 and not args.no_cpu:
            sample = (args.cpu_sample_mib << 20) if args.cpu_sample_mib > 0 else (256 << 20)
            cb = cpu_baseline_partitions(ss, world, per_gpu * world, sample, args.seed, kind, local_rank)
            if cb is not None:
                sac["cpu_baseline"] = cb
        if do_global:
            # ---- the global leg (defines `value` of an N > 1 line when it finishes).  A failure or a hang anywhere in it —
            # communicator, self-test, builds — must not cost the run its line: after --global-timeout seconds, or on an
            # exception, rank 0 prints the sacapart leg (the reference's own multi-GPU semantics, measured above) as `value`,
            # says so in `value_mode` and `global_mode.error`, and every rank exits 0 (ranks may be stuck inside a collective:
            # os._exit).
            import threading
            done = threading.Event()
            give_lock = threading.Lock()

            def give_up(why):
                with give_lock:
                    if done.is_set():
                        return
                    done.set()
                    if rank == 0:
                        sac["global_mode"] = {"error": why}
                        sac["value_sacapart"] = sac.get("value")
                        sac["value_mode"] = ("sacapart: N independent local suffix arrays, one chunk per GPU, no collective (crates/sacapart/src/lib.rs:39-58) — "
                                             "the global leg did not finish, see global_mode.error")
                        sac["transport_selftest"] = selftest
                        emit(sac)
                    sys.stdout.flush(); sys.stderr.flush()
                    os._exit(0)
            wd = threading.Timer(args.global_timeout, give_up, args=(f"no result after {args.global_timeout} s",))
            wd.daemon = True
            wd.start()
            outg = None
            try:
                from stringsearch_amd.bench_global import run_global
                if os.environ.get("DC3HIP_BENCH_FAIL_GLOBAL") == "1":        # (tests: the fallback line below)
                    raise RuntimeError("global leg failed on request (DC3HIP_BENCH_FAIL_GLOBAL=1)")
                G, selftest = global_setup()
                outg = run_global(args, ss, dist, backend, world, rank, local_rank, per_gpu, kind, barrier, G=G)
            except BaseException as e:          # noqa: BLE001 - reported in the line
                print(f"bench.py: rank {rank}: the global leg failed: {e!r}", file=sys.stderr, flush=True)
                give_up(repr(e))
                time.sleep(3600)                # (the watchdog thread is printing / exiting: never fall through)
            with give_lock:
                if done.is_set():               # the watchdog fired while the leg was returning: it owns the exit
                    time.sleep(3600)
                wd.cancel()
                done.set()
            if rank == 0:
                keep = ("value", "unit", "ms_per_step", "value_MiBps", "roofline", "roofline_path", "verify", "config", "path", "arena_peak_GB")
                outg["sacapart"] = {k: sac[k] for k in keep if k in sac}
                outg["sacapart"]["note"] = ("the reference's own multi-chunk semantics (crates/sacapart/src/lib.rs:39-58): N independent local "
                                            "suffix arrays, one chunk per GPU, no data-path collective; same K timed steps")
                outg["value_mode"] = "global (ONE suffix array over all ranks, rank exchange over the transport in `interconnect`)"
                outg["transport_selftest"] = selftest
                if "cpu_baseline" in sac:
                    outg["cpu_baseline"] = sac["cpu_baseline"]
                out = outg
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (`hip_runtime`: HIP_VERSION the library was compiled against vs the runtime it really ran on — the wheel's when
        # torch had to be imported, N > 1: profiles/r05_crash_hunt.md)
        emit(out)


if __name__ == "__main__":
    main()
