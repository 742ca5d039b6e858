#!/usr/bin/env python3
"""bench.py — MB/s of input indexed (suffix-array build) on MI355X, the metric of BASELINE.json.

A "step" = one device-resident DC3 suffix-array build of this rank's partition (text already in HBM,
SA left in HBM).  N GPUs = sacapart partitioning (crates/sacapart/src/lib.rs:39-58): the N x SIZE byte
text is cut into chunks of len/N + 1 bytes, rank c builds the independent local SA of chunk c — no
data-path collective, weak scaling.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size BYTES] [--kind random|dna]
                    [--cpu-sample-mib M] [--no-cpu] [--no-verify]
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
COPY_MEASURED_GBS = 4735.0   # read+write copy kernel measured on the box (profiles/r01_membench_access_patterns.txt)


def parse_size(s):
    s = s.strip().lower()
    mult = 1
    for suf, m in (("gib", 1 << 30), ("mib", 1 << 20), ("kib", 1 << 10), ("g", 1 << 30), ("m", 1 << 20), ("k", 1 << 10)):
        if s.endswith(suf):
            s, mult = s[: -len(suf)], m
            break
    return int(float(s) * mult)


def algorithmic_bytes(level_n):
    """B(n_l) = n_l * (46w + 29c_l) / 3, w = 4; c_0 = 1 (bytes), c_l = 4 deeper."""
    total = 0.0
    for lvl, n in enumerate(level_n):
        c = 1 if lvl == 0 else 4
        total += n * (46 * 4 + 29 * c) / 3.0
    return total


def cpu_baseline(text_u8, sample_bytes):
    """The reference's CPU path (libdivsufsort built from /root/reference into oracle/_ref) or, if that
    is absent, our C restatement of crates/dc3 — timed on one host core like divsuftest's measure()
    (crates/divsuftest/src/main.rs:145-151: wall clock around the call incl. the SA allocation)."""
    import numpy as np
    sample = np.ascontiguousarray(text_u8[:sample_bytes])
    ref = os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so")
    port = os.path.join(ROOT, "oracle", "liboracle_dc3.so")
    if os.path.exists(ref):
        L = ctypes.CDLL(ref); f = L.divsufsort; kind = "reference"
    elif os.path.exists(port):
        L = ctypes.CDLL(port); f = L.dc3_oracle_sufsort_i32; kind = "port"
    else:
        return None
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]; f.restype = ctypes.c_int32
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass
    t0 = time.perf_counter()
    sa = np.zeros(len(sample), dtype=np.int32)
    rc = f(sample.ctypes.data, sa.ctypes.data, len(sample))
    dt = time.perf_counter() - t0
    assert rc == 0
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": len(sample) / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": f"first {len(sample) / 2**20:.0f} MiB of the same buffer, one divsufsort() call, "
                      f"wall clock incl. SA allocation ({dt:.2f} s)",
            "host_cpu": model, "host_cores_available": os.cpu_count()}, sa


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=str, default="1GiB", help="bytes per GPU (partition size is size*N/N+1 rounding aside)")
    ap.add_argument("--kind", type=str, default="random", choices=["random", "dna", "text"])
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample-mib", type=int, default=256,
                    help="CPU baseline sample (MiB of the same buffer); 256 MiB is ~10-20 s of one-core divsufsort")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-recursion-line", action="store_true",
                    help="skip the extra (untimed-in-value) builds with the whole-text shortcut disabled")
    ap.add_argument("--dump-stats", type=str, default=None, help="write the last build's dc3hip_stats as JSON here")
    args = ap.parse_args()

    import numpy as np
    import torch            # first: libdc3hip.so then shares the HIP runtime torch loaded
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes", file=sys.stderr)
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # one rank per GPU; DC3HIP_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing test on 1-GPU boxes)
    backend = os.environ.get("DC3HIP_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import stringsearch_amd as ss

    from stringsearch_amd.partition import rank_chunk
    per_gpu = parse_size(args.size)
    total_len = per_gpu * world
    off, n = (0, total_len) if world == 1 else rank_chunk(total_len, world, rank)   # sacapart/src/lib.rs:43-46
    kind = {"random": 0, "dna": 1, "text": 2}[args.kind]

    ctx = ss.Context(n, device=local_rank)
    ctx.generate(n, args.seed, kind, offset=off)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.build()
    verify = {}
    if not args.no_verify:
        if args.warmup == 0:
            ctx.build()
        rc = ctx.sufcheck()                  # full-size GPU sufcheck (utils.c:160-241 semantics)
        assert rc == 0, f"rank {rank}: SA failed sufcheck ({rc}) — no throughput reported"
        verify["sufcheck_full"] = rc
        chk0 = ctx.checksum()

    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    dsw_ms = [0.0] * 3; dsw_launches = [0] * 3; dsw_elems = [0] * 3
    g_ms = 0.0; g_launches = 0; g_elems = 0
    for _ in range(args.steps):
        ctx.build()
        st = ctx.stats()
        kernel_ms += st["build_ms"]
        g_ms += st["gather_ms"]; g_launches += st["gather_launches"]; g_elems += st["gather_elems"]
        for k in range(3):
            dsw_ms[k] += st["downsweep_ms"][k]; dsw_launches[k] += st["downsweep_launches"][k]
            dsw_elems[k] += st["downsweep_elems"][k]
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if not args.no_verify:
        assert ctx.checksum() == chk0, "SA changed between identical builds"
        verify["idempotent_checksum"] = True

    st = ctx.stats()
    out = None
    if rank == 0:
        value = total_len * args.steps / dt / 1e6
        # dominant kernel: the stable 8-bit radix scatter k_rs_downsweep<Rec,...> — the record type
        # (8-byte pairs / 16-byte triple records / 20-byte mod-0 tuples) with the largest summed time.
        # algorithmic bytes per record-pass = the reference's scatter loop lib.rs:35-38: read a[i] (w) +
        # r[a[i]] (c) + write b[..] (w) = 2w + c = 12 B at w = c = 4 (scatter half).
        roof = None
        kc = max(range(3), key=lambda k: dsw_ms[k])
        if dsw_launches[kc]:
            rec_bytes = (8, 16, 20)[kc]
            rec_name = ("Rec8 (key,value) pairs", "Rec16 triple records (12-byte Rec12 when the key fits 64 bits)", "Tup0 mod-0 tuples")[kc]
            per_launch_elems = dsw_elems[kc] / dsw_launches[kc]
            avg_ms = dsw_ms[kc] / dsw_launches[kc]
            achieved = 12.0 * per_launch_elems / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": f"k_rs_downsweep<{rec_name.split()[0]}> (stable 8/9-bit-digit radix scatter of {rec_name})",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None,
                    "algorithmic_bytes_per_launch": 12.0 * per_launch_elems, "avg_launch_ms": avg_ms,
                    "launches_per_step": dsw_launches[kc] / args.steps,
                    "share_of_build_time": dsw_ms[kc] / kernel_ms,
                    "moved_bytes_per_launch": 2.0 * rec_bytes * per_launch_elems,
                    "moved_GBps": 2.0 * rec_bytes * per_launch_elems / (avg_ms * 1e-3) / 1e9,
                    "all_record_types_ms_per_step": [x / args.steps for x in dsw_ms]}
        pmc = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            pass
        if roof is not None and pmc is not None:
            key = ("downsweep_rec8", "downsweep_rec16", "downsweep_tup0")[kc]
            if key in pmc.get("bytes_per_record", {}):
                roof["traffic"] = pmc["bytes_per_record"][key] * per_launch_elems
                roof["traffic_source"] = pmc.get("source")
        roof_gather = None
        if g_launches:
            ge = g_elems / g_launches; gms = g_ms / g_launches
            # algorithmic bytes of the gather = what the reference's merge reads at random per sample suffix
            # (lib.rs:136-162, SURVEY §8d merge row): SA12 entry (w) + position (w) + one rank (w) + 2 symbols (2c)
            alg_g = 0.0
            for lvl, mm in enumerate(st["level_n"]):
                if mm < 2:
                    continue
                m02 = (mm + 2) // 3 + mm // 3
                alg_g += m02 * (3 * 4 + 2 * (1 if lvl == 0 else 4))
            alg_g /= (g_launches / args.steps)
            roof_gather = {"bound": "hbm", "kernel": "k_gather_tuples (one random 16-byte gather per sample suffix)",
                           "achieved": alg_g / (gms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": alg_g / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic": (pmc["bytes_per_record"]["gather_tuples"] * ge) if pmc and "gather_tuples" in pmc.get("bytes_per_record", {}) else None,
                           "algorithmic_bytes_per_launch": alg_g, "avg_launch_ms": gms,
                           "launches_per_step": g_launches / args.steps, "share_of_build_time": g_ms / kernel_ms,
                           "moved_bytes_per_launch": 36.0 * ge, "moved_GBps": 36.0 * ge / (gms * 1e-3) / 1e9,
                           "gathers_per_second_G": ge / (gms * 1e-3) / 1e9,
                           "traffic_source": pmc.get("source") if pmc else None}
        # `roofline` = the kernel with the largest share of the build; the other one is kept beside it
        roof_radix = roof
        for rr in (roof, roof_gather):
            if rr is not None:
                rr["measured_copy_GBps"] = COPY_MEASURED_GBS
                rr["moved_frac_of_measured_copy"] = rr["moved_GBps"] / COPY_MEASURED_GBS
        if roof_gather is not None and (roof is None or roof_gather["share_of_build_time"] >= roof["share_of_build_time"]):
            roof = roof_gather
        alg = algorithmic_bytes(st["level_n"])
        path = {"algorithmic_bytes_per_step": alg, "device_ms_per_step": kernel_ms / args.steps,
                "achieved_GBps": alg / (kernel_ms / args.steps * 1e-3) / 1e9,
                "frac_of_hbm_peak": alg / (kernel_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "levels": list(zip(st["level_n"], st["level_K"], st["level_sorted"])),
                "phase_ms": {k: round(v, 3) for k, v in st["phase_ms"].items() if v}}
        out = {
            "metric": "MB/s of input indexed (SA build), 1 GiB bytes, 1/2/4/8 GPUs",
            "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{per_gpu / 2**30:g} GiB {args.kind} bytes per GPU (splitmix64 seed {args.seed}), "
                                   f"i32 SA, DC3 HIP, text and SA resident in HBM",
                       "bytes_per_gpu": n, "total_bytes": total_len,
                       "partitioning": "single SA" if world == 1 else f"sacapart: {world} chunks of len/{world}+1 bytes, one per GPU, no collective"},
            "value_MiBps": total_len * args.steps / dt / 2**20,     # the reference prints binary units (divsuftest main.rs:179-183)
            "roofline": roof, "roofline_radix_scatter": roof_radix, "roofline_path": path, "verify": verify,
            "arena_peak_GB": st["arena_peak"] / 1e9,
        }
        if world == 1 and not args.no_cpu:
            text = ctx.text()
            sample_bytes = min(n, args.cpu_sample_mib << 20)
            res = cpu_baseline(text, sample_bytes)
            if res is not None:
                cb, cpu_sa = res
                out["cpu_baseline"] = cb
                if sample_bytes == n and not args.no_verify:
                    out["verify"]["equal_cpu_reference"] = bool(np.array_equal(cpu_sa, ctx.sa()))
    if args.dump_stats and rank == 0:
        json.dump(st, open(args.dump_stats, "w"))
    ctx.close()
    if rank == 0 and world == 1 and not args.no_recursion_line and st.get("text_sort_state", 0) == 1:
        # The timed builds above finished in the whole-text shortcut (every 9-byte window distinct, no
        # recursion level built).  For reference, the same input through the DC3 recursion proper
        # (levels, tuples, merge): not part of `value`.
        os.environ["DC3HIP_NO_TEXT_SHORTCUT"] = "1"
        try:
            with ss.Context(n, device=local_rank) as c2:
                c2.generate(n, args.seed, kind, offset=off)
                c2.build()
                ms = []
                for _ in range(min(args.steps, 3)):
                    c2.build(); ms.append(c2.stats()["build_ms"])
                st2 = c2.stats()
                out["dc3_recursion_only"] = {"switch": "DC3HIP_NO_TEXT_SHORTCUT=1", "device_ms_per_step": sum(ms) / len(ms),
                                             "MBps": n / (sum(ms) / len(ms)) / 1e3, "sufcheck": c2.sufcheck(),
                                             "checksum_equal": (c2.checksum() == chk0) if not args.no_verify else None,
                                             "levels": list(zip(st2["level_n"], st2["level_K"], st2["level_sorted"]))}
        finally:
            os.environ.pop("DC3HIP_NO_TEXT_SHORTCUT", None)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
