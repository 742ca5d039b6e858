"""GPU parity tests (-m gpu) of the GLOBAL mode: one suffix array over P ranks (include/dc3hip.h "GLOBAL mode").
The result is defined by the single-device build / divsufsort: the shards, concatenated in rank order, must equal
SA[0..n) of the whole text bit for bit.  P in {2,4,8} runs as loopback ranks on the one GPU of the box; the RCCL
transport is exercised with a one-rank communicator (RCCL refuses two ranks on one device)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ss():
    import stringsearch_amd as ss
    assert ss.device_count() >= 1, "gpu tests need a device; the library has no CPU fallback"
    return ss


class env:
    """with env(DC3HIP_GLOBAL_LOCAL_MAX=64, DC3HIP_NO_HYBRID=1): policy variables go to the environment, every other DC3HIP_*
    name is a test switch and travels in DC3HIP_DEBUG (stringsearch_amd.debug_switches) — the library reads nothing else."""

    def __init__(self, **kv):
        import stringsearch_amd as ss
        self.kv = {k: v for k, v in kv.items() if k in ss.POLICY_VARS or not k.startswith("DC3HIP_")}
        self.dbg = ss.debug_switches(**{k: v for k, v in kv.items() if k not in self.kv})

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)
        self.dbg.__enter__()

    def __exit__(self, *a):
        self.dbg.__exit__()
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def want_sa(oracle, data):
    return (oracle.ref_sufsort(data) if oracle.ref is not None else oracle.sufsort(data)).astype(np.int64)


def inputs(oracle):
    rng = np.random.default_rng(11)
    out = {}
    out["random256"] = oracle.gen(300_001, 2, 0)
    out["dna"] = oracle.gen(250_000, 5, 1)
    out["text"] = oracle.gen(400_003, 3, 2)
    x = oracle.gen(200_000, 7, 0)
    out["dup_block"] = np.concatenate([x, x[50_000:120_000], oracle.gen(30_002, 8, 0)])
    out["all_equal"] = np.full(70_001, 65, dtype=np.uint8)
    out["period3"] = np.tile(np.frombuffer(b"abc", dtype=np.uint8), 40_000)[:119_999]
    out["two_symbols"] = rng.integers(0, 2, 150_000).astype(np.uint8)
    out["zeros_inside"] = (rng.integers(0, 3, 100_000) * 127).astype(np.uint8)
    return out


@pytest.mark.parametrize("P", [2, 4, 8])
def test_loopback_matches_single_device_all_levels_distributed(ss, oracle, P):
    """Every level down to 64 symbols runs distributed (key-range selection, rank exchange, sliced merge): the
    concatenated shards equal the reference SA on inputs with 2..18 recursion levels."""
    data = inputs(oracle)
    with env(DC3HIP_GLOBAL_LOCAL_MAX=64, DC3HIP_GLOBAL_NO_TEXT_ORDER=1):
        with ss.LoopbackGroup(P, max(len(v) for v in data.values())) as g:
            for name, t in data.items():
                g.set_text(t)
                g.build()
                got = g.sa()
                assert np.array_equal(got, want_sa(oracle, t)), (name, P)
                st = g.stats()
                assert all(s["nranks"] == P and s["exchanges"] >= 2 for s in st), name
                assert sum(s["shard_count"] for s in st) == len(t)


@pytest.mark.parametrize("P", [2, 3, 4])
def test_loopback_wide_windows_all_levels_distributed(ss, oracle, P):
    """As above with the splitter ordering's threshold lowered: wherever a rank's share of a level has 8192 samples the
    straight key-range sort orders W-symbol windows (4-7 symbols per 16-byte record instead of the triple) by sampled
    splitters; names from wider windows change which slots the discarding recursion keeps on every level below."""
    data = inputs(oracle)
    data["text_3m"] = oracle.gen(3_000_001, 4, 2)
    with env(DC3HIP_GLOBAL_LOCAL_MAX=64, DC3HIP_GLOBAL_NO_TEXT_ORDER=1, DC3HIP_SSORT_MIN=8192):
        with ss.LoopbackGroup(P, max(len(v) for v in data.values())) as g:
            for name, t in data.items():
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), want_sa(oracle, t)), (name, P)


@pytest.mark.parametrize("P", [2, 4, 8])
def test_loopback_default_policy(ss, oracle, P):
    """Default thresholds (small levels finished locally, whole-text order allowed) and n % 3 in {0,1,2}."""
    with ss.LoopbackGroup(P, 6_000_002) as g:
        for n, kind, seed in [(6_000_000, 0, 2), (5_000_002, 0, 4), (5_000_001, 2, 3), (4_500_000, 1, 5)]:
            t = oracle.gen(n, seed, kind)
            g.set_text(t)
            g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, t)), (n, kind, P)
            st = g.stats()
            if kind in (0, 1):
                assert all(s["text_order"] == 1 for s in st)        # every 9-byte (DNA: 39-symbol) window distinct
            with ss.Context(n) as c:
                c.set_text(t); c.build()
                assert g.checksum() == c.checksum()                 # checksum of shards == single-device checksum


def test_loopback_small_alphabet_whole_text_order(ss, oracle):
    """The distributed whole-text order with 3L-symbol windows (small alphabets): random texts over 2..5 symbols finish
    at level 0 on every rank; a repeated block of 6000 symbols sends all ranks on to the recursion (one of 700 is settled by
    the second, deeper tie pass); an alphabet containing 0x00 and a
    run of the smallest symbol at the end of the text.  Bit-exact against divsufsort."""
    rng = np.random.default_rng(41)
    n = (1 << 22) + 5                 # above the size the ranks would finish locally (DC3HIP_GLOBAL_LOCAL_MAX)
    with ss.LoopbackGroup(3, n) as g:
        for sigma in (2, 3, 4, 5):
            for variant in ("random", "repeat", "short_repeat", "zero_run_at_end"):
                t = rng.integers(0, sigma, size=n, dtype=np.uint8)
                if variant == "repeat":
                    t[n // 2:n // 2 + 6000] = t[50:6050]
                if variant == "short_repeat":
                    t[n // 2:n // 2 + 700] = t[50:750]
                if variant == "zero_run_at_end":
                    t[n - 150:] = 0; t[300:420] = 0
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), want_sa(oracle, t)), (sigma, variant)
                st = g.stats()
                if variant in ("random", "short_repeat"):    # (a repeat below 2048 symbols is settled by the second, deeper tie pass)
                    assert all(s["text_order"] == 1 for s in st), (sigma, variant, [s["text_order"] for s in st])
                if variant == "repeat":
                    assert all(s["text_order"] == 0 and s["levels"] >= 2 for s in st), sigma


@pytest.mark.parametrize("P", [2, 5])
def test_loopback_whole_text_order_on_12_byte_records(ss, oracle, P):
    """The distributed whole-text order on 12-byte records (the default beyond 2^31 positions, forced here by
    DC3HIP_TEXT_ORDER12=1): random bytes (9-byte windows), DNA and a binary text (3L-symbol windows), and a text with a
    repeated block, which no rank may finish (all ranks go on to the recursion)."""
    rng = np.random.default_rng(43)
    n = (1 << 22) + 11
    with env(DC3HIP_TEXT_ORDER12=1), ss.LoopbackGroup(P, n) as g:
        cases = {"bytes": rng.integers(0, 256, size=n, dtype=np.uint8), "dna": oracle.gen(n, 5, 1),
                 "binary": rng.integers(0, 2, size=n, dtype=np.uint8)}
        rep = cases["dna"].copy(); rep[n // 3:n // 3 + 5000] = rep[10:5010]
        cases["dna_repeat"] = rep
        for label, t in cases.items():
            g.set_text(t)
            g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, t)), (label, P)
            st = g.stats()
            assert all(s["text_order"] == (0 if label == "dna_repeat" else 1) for s in st), (label, [s["text_order"] for s in st])
            assert all(s["ctx"]["downsweep_launches"][1] > 0 for s in st if s["shard_count"]), label      # 12-byte passes ran


def test_loopback_generated_blocks_and_tiny_inputs(ss, oracle):
    """Ranks generate only their own block (offset-addressable generator); n in {0,1,2,3,...} and n < P."""
    with ss.LoopbackGroup(4, 100_000) as g:
        for n in (0, 1, 2, 3, 4, 5, 7, 64, 65, 1000, 99_999):
            g.generate(n, 9, 1)
            g.build()
            t = oracle.gen(n, 9, 1)
            assert np.array_equal(g.sa(), want_sa(oracle, t) if n else np.zeros(0, dtype=np.int64)), n
    with env(DC3HIP_GLOBAL_LOCAL_MAX=0, DC3HIP_GLOBAL_NO_TEXT_ORDER=1):
        with ss.LoopbackGroup(8, 3000) as g:
            for n in (70, 100, 257, 2999):
                g.generate(n, 3, 2)
                g.build()
                assert np.array_equal(g.sa(), want_sa(oracle, oracle.gen(n, 3, 2))), n


def test_loopback_64mib_random_and_text_properties(ss, oracle):
    """BASELINE configs[1] size through 4 ranks: random (distributed whole-text order) and low-entropy text
    (distributed recursion) — checksum of the shards equals the single-device checksum; GPU sufcheck of the
    concatenated result."""
    n = 64 << 20
    for kind, seed in ((0, 2), (2, 3)):
        with ss.LoopbackGroup(4, n) as g:
            g.generate(n, seed, kind)
            g.build()
            chk = g.checksum()
            sa = g.sa()
            st = g.stats()
        with ss.Context(n) as c:
            c.generate(n, seed, kind)
            c.build()
            assert c.checksum() == chk, kind
            c.set_sa(sa.astype(np.int32))
            assert c.sufcheck() == 0
        assert (st[0]["text_order"] == 1) == (kind == 0)
        assert all(s["comm_bytes_in"] > 0 for s in st)


def test_rccl_transport_single_rank(ss, oracle):
    """The RCCL backend end to end with a one-rank communicator: dlopen of librccl, unique id, ncclCommInitRank,
    grouped send/recv paths with no peers, ncclAllGather of the host counts."""
    uid = ss.GlobalRank.rccl_unique_id()
    assert len(uid) == 128
    # one RCCL per process: the library binds to the librccl the host program (torch) already mapped, and no second one
    path, pre = ss.GlobalRank.rccl_library()
    mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
    assert len(mapped) == 1 and os.path.realpath(mapped[0]) == os.path.realpath(path), (path, mapped)
    # ... and when the HIP runtime in use is a wheel's own (torch/lib/libamdhip64.so), the RCCL is the wheel's too, whether
    # torch had mapped it already (pre) or the library was the first to need it (the system's librccl on top of the
    # wheel's runtime aborts at exit)
    # (torch is NOT imported here: a process that loaded libdc3hip on the system's HIP runtime and imports torch afterwards
    #  ends up with two runtimes and two RCCLs, and aborts at exit — bench.py and smoke() import torch first for that reason)
    hip = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
    assert len(hip) == 1, hip
    assert os.path.dirname(os.path.realpath(path)) == os.path.dirname(os.path.realpath(hip[0])), (path, hip, pre)
    with env(DC3HIP_GLOBAL_FORCE_DIST=1, DC3HIP_GLOBAL_LOCAL_MAX=1000):
        r = ss.GlobalRank.rccl(uid, 0, 1, 0, 3_000_000)
    try:
        assert "RCCL" in r.transport()
        for kind, n in ((2, 777_777), (0, 2_500_000)):
            t = oracle.gen(n, 2, kind)
            r.set_text_block(t, len(t))
            r.build()
            first, sa = r.shard_sa()
            assert first == 0 and np.array_equal(sa, want_sa(oracle, t))
            st = r.stats()
            assert st["local_from_level"] != 0 and (st["exchanges"] >= 2 or st["text_order"] == 1)
    finally:
        r.close()


def test_bench_global_two_processes_on_one_gpu(ss, oracle, tmp_path):
    """bench.py --mode global as the driver would launch it (python -m torch.distributed.run, 2 processes), both ranks on
    GPU 0 with the host-staged transport over gloo (RCCL refuses two ranks on one device): the rank-0 line is well formed
    and the dumped shards concatenate to the reference suffix array of the whole text."""
    import subprocess, sys, socket
    from conftest import ROOT, bench_line
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    size = 3 << 20
    for kind, extra in (("random", {}), ("text", {"DC3HIP_GLOBAL_LOCAL_MAX": "4096"}), ("random", {"DC3HIP_DEBUG": "global_force_wide"})):
        env2 = dict(os.environ, DC3HIP_BENCH_BACKEND="gloo", DC3HIP_BENCH_DUMP_SA=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "global", "--size", str(size),
               "--kind", kind, "--steps", "2", "--warmup", "1", "--detail", str(tmp_path / "detail.json")]
        p = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
        line, full = bench_line(p.stdout)
        assert line["n_gpus"] == 2 and "global SA" in line["config"]["partitioning"] and line["config"]["total_bytes"] == 2 * size
        if extra.get("DC3HIP_DEBUG"):     # the 64-bit-position mode of texts beyond 2^32 bytes, two processes
            assert line["verify"]["shards_tile_0_n"] and line["verify"]["global_sufcheck"] == 0 and "64-bit" in line["config"]["workload"]
        else:
            assert line["verify"]["shards_tile_0_n"] and line["verify"]["equal_single_device_checksum"]
        assert all(b > 0 for b in full["interconnect"]["bytes_in_per_rank_per_step"]) and line["interconnect"]["comm_ms"] > 0
        assert line["transport_selftest"]["passed"] is True and line["transport_selftest"]["ranks_seen_by_transport"] == 2
        text = oracle.gen(2 * size, 2, {"random": 0, "text": 2}[kind])
        got = np.concatenate([np.load(tmp_path / f"gshard_{r}.npy") for r in range(2)])
        assert np.array_equal(got, want_sa(oracle, text)), kind


def test_routed_order_and_bucket_sort_of_a_key_range(ss, oracle):
    """The distributed whole-text order in its routed form (every rank packs its own block, records go to the owner of their
    key range by one all-to-all) against the round-2 form (every rank evaluates every position: DC3HIP_GLOBAL_NO_ROUTE=1),
    with the bucket (MSD) ordering forced onto the small key-range slices (DC3HIP_MSD_MIN: digits are taken from
    image - range start) and without it: same shards, equal to the reference suffix array.  Random bytes, DNA (39-symbol
    windows) and a text with a planted repeat (the order is not distinct: falls through to the distributed recursion)."""
    n = 6_000_011
    rnd = oracle.gen(n, 21, 0)
    dna = oracle.gen(n, 22, 1)
    rep = rnd.copy(); rep[4_000_000:4_050_000] = rep[100_000:150_000]
    for name, t in (("random", rnd), ("dna", dna), ("planted_repeat", rep)):
        want = want_sa(oracle, t)
        for P in (2, 4, 7):
            for extra in ({"DC3HIP_MSD_MIN": 4096}, {"DC3HIP_NO_MSD": 1}, {"DC3HIP_GLOBAL_NO_ROUTE": 1, "DC3HIP_MSD_MIN": 4096}):
                with env(**extra):
                    with ss.LoopbackGroup(P, n) as g:
                        g.set_text(t)
                        g.build()
                        st = g.stats()
                        assert np.array_equal(g.sa(), want), (name, P, extra)
                        if name != "planted_repeat":
                            assert all(s["text_order"] == 1 for s in st), (name, P, extra)
                            if "DC3HIP_GLOBAL_NO_ROUTE" not in extra:
                                # routed: a rank receives about n / P records, and sent about (P-1)/P of its block
                                assert all(abs(s["shard_count"] - n / P) < 0.25 * n / P for s in st), [s["shard_count"] for s in st]
                                assert all(s["comm_bytes_out"] > 0 for s in st)
                            if "DC3HIP_MSD_MIN" in extra and "DC3HIP_GLOBAL_NO_ROUTE" not in extra:
                                # (unrouted slices have arbitrary image ranges: the bucket ordering may give up on them)
                                assert all(s["ctx"]["msd_sorts"] >= 1 and s["ctx"]["msd_fallbacks"] == 0 for s in st), (name, P, extra)


def test_unrouted_bucket_order_of_texts_below_2pow32(ss, oracle):
    """The top-level whole-text order of a text below 2^32 bytes in the wide contexts' form (gorder_text_msd: every rank
    selects the images of its range straight from its replica of the text inside partition pass 1 of the bucket ordering, 8-byte
    words, tie rounds with lazily compared windows; nothing but the text blocks crosses the transport), forced onto small
    texts (DC3HIP_WIDE_MSD_MIN=1): same shards as the reference suffix array for random bytes, DNA, a binary alphabet and a
    text with a short planted repeat (settled by the deeper tie rounds); a long planted repeat is not distinct within the
    budget of the tie rounds -> the distributed recursion builds it, still equal to the reference."""
    n = 6_000_011
    rng = np.random.default_rng(77)
    rnd = oracle.gen(n, 21, 0)
    dna = oracle.gen(n, 22, 1)
    binary = (rng.integers(0, 2, size=3_000_001, dtype=np.uint8) + 65)
    short = rnd.copy(); short[4_000_000:4_000_300] = short[100_000:100_300]
    long_ = dna.copy(); long_[1_000_000:3_500_000] = long_[5:2_500_005]
    for name, t in (("random", rnd), ("dna", dna), ("binary", binary), ("short_repeat", short), ("long_repeat", long_)):
        want = want_sa(oracle, t)
        for P in (2, 4, 7):
            with env(DC3HIP_WIDE_MSD_MIN=1):
                with ss.LoopbackGroup(P, len(t)) as g:
                    g.set_text(t)
                    g.build()
                    st = g.stats()
                    assert np.array_equal(g.sa(), want), (name, P)
                    if name == "binary":
                        pass          # (a binary alphabet at this size is not offered to any whole-text order: only the array counts)
                    elif name != "long_repeat":
                        assert all(s["text_order"] == 1 and s["wide_msd"] == 1 for s in st), (name, P, [(s["text_order"], s["wide_msd"]) for s in st])
                        # unrouted: a rank received the other ranks' text blocks and nothing else
                        assert all(s["comm_bytes_in"] <= len(t) + 4096 for s in st), [s["comm_bytes_in"] for s in st]
                        assert all(abs(s["shard_count"] - len(t) / P) < 0.25 * len(t) / P for s in st), [s["shard_count"] for s in st]
                    else:
                        assert all(s["text_order"] == 0 for s in st), (name, P)
                    g.build()                                  # idempotent (the context's buffers are reused)
                    assert np.array_equal(g.sa(), want), (name, P, "second build")


def test_selecting_partition_pass_of_the_global_orderings(ss, oracle):
    """The default form of the global mode's whole-text and whole-level orderings (MsdPass1KeysSel): every rank walks its
    replica of the string, and partition pass 1 of the single device's bucket ordering keeps the words of the rank's image
    range only — no records built, none routed.  Forced onto small inputs (DC3HIP_MSD_MIN=4096): same array as the
    reference for random bytes (top level: byte images straight off the text bits), bytes with zero runs against the zero
    padding and short repeats (tied images, distinct windows), a planted long repeat and generated text (the recursion:
    name triples selected at the levels below), for 2, 3, 4 and 7 ranks, with and without the wider image; and the
    previous forms behind DC3HIP_GLOBAL_NO_SELECT=1."""
    rng = np.random.default_rng(404)
    n = 5_000_011
    rnd = rng.integers(0, 256, size=n, dtype=np.uint8)
    z = rnd.copy()
    for at in rng.integers(0, n - 16, size=20_000):
        z[at:at + int(rng.integers(1, 12))] = 0
    z[-13:] = 0
    rep = rnd.copy()
    for at, src in zip(rng.integers(0, n - 16, size=30_000), rng.integers(0, n - 16, size=30_000)):
        ln = int(rng.integers(5, 10))
        rep[at:at + ln] = rep[src:src + ln].copy()
    long_ = rnd.copy(); long_[1_000_000:1_400_000] = long_[3_000_000:3_400_000]
    text = oracle.gen(3_000_001, 5, 2)
    for name, t in (("random", rnd), ("zero_runs", z), ("short_repeats", rep), ("long_repeat", long_), ("text", text)):
        want = want_sa(oracle, t)
        for P in (2, 3, 4, 7):
            # (without the whole-text order random bytes reach level 1 with 16 M names: its whole-level order selects name triples)
            for extra in ({}, {"DC3HIP_NO_PACK_STRIP": "1"}, {"DC3HIP_GLOBAL_NO_SELECT": "1"}, {"DC3HIP_GLOBAL_NO_TEXT_ORDER": "1", "DC3HIP_GLOBAL_LOCAL_MAX": "64"}):
                if extra and P in (3, 7):
                    continue
                if "DC3HIP_GLOBAL_NO_TEXT_ORDER" in extra and name != "random":
                    continue
                with env(DC3HIP_MSD_MIN=4096, **extra):
                    with ss.LoopbackGroup(P, len(t)) as g:
                        g.set_text(t)
                        g.build()
                        st = g.stats()
                        assert np.array_equal(g.sa(), want), (name, P, extra)
                        if "DC3HIP_GLOBAL_NO_SELECT" in extra:
                            assert all(s["select_p1"] == 0 for s in st), (name, P)
                        elif name in ("random", "zero_runs", "short_repeats"):
                            assert all(s["select_p1"] >= 1 for s in st), (name, P, extra, [s["select_p1"] for s in st])
                            if name == "random" and "DC3HIP_GLOBAL_NO_TEXT_ORDER" not in extra:       # (nothing but the text blocks crossed the transport)
                                assert all(s["text_order"] == 1 and s["comm_bytes_in"] <= len(t) + 4096 for s in st), (name, P)
                        g.build()
                        assert np.array_equal(g.sa(), want), (name, P, extra, "second build")
    # select or route, by the per-rank cost model (gselect_pays): ranks that share a device have no link to pay and select
    # whatever P is; told that the links are xGMI (DC3HIP_GLOBAL_LINK_GBPS=153, what an RCCL group reports) the same groups
    # select up to 4 ranks and route beyond — same array either way — and the build says what P GPUs would take
    want = want_sa(oracle, rnd)
    for link, P, selects in (("0", 8, True), ("153", 2, True), ("153", 4, True), ("153", 5, False), ("153", 8, False), ("25", 8, True)):
        with env(DC3HIP_MSD_MIN=4096, DC3HIP_GLOBAL_LINK_GBPS=link, DC3HIP_GLOBAL_DEVICE_TOKEN=1):
            with ss.LoopbackGroup(P, len(rnd)) as g:
                g.set_text(rnd)
                g.build()
                st = g.stats()
                assert np.array_equal(g.sa(), want), (link, P)
                assert all((s["select_p1"] >= 1) == selects for s in st), (link, P, [s["select_p1"] for s in st])
                assert all(s["work_ms"] > 0 and s["collectives"] >= 1 and s["link_ms"] > 0 for s in st), st
                # the ranks pass a device token: a rank's own work is well below the wall time of P ranks sharing the GPU
                assert max(s["work_ms"] for s in st) <= max(s["wall_ms"] for s in st) + 1e-6
                if P == 8:
                    assert sum(s["work_ms"] for s in st) <= 1.25 * max(s["wall_ms"] for s in st) + 5.0, [(s["work_ms"], s["wall_ms"]) for s in st]


def test_transport_selftest_and_recovery_after_a_failed_collective(ss, oracle):
    """dc3hip_global_selftest (ragged all-to-all / all-gather of known bytes, every byte checked) on loopback groups, and
    the failure semantics of the header: after a collective that failed on every rank (a one-symbol text in a wide
    context without the deepening is refused with -4) the group works again through ANY entry point — the self-test and a build — without the
    loopback_build wrapper having to reset it (round-2 advisor finding)."""
    for P in (2, 3, 8):
        with ss.LoopbackGroup(P, 100_000) as g:
            assert g.selftest() == [P] * P
    with env(DC3HIP_GLOBAL_FORCE_WIDE=1, DC3HIP_NO_WIDE_DEEPEN=1):
        with ss.LoopbackGroup(2, 50_000) as g:
            g.set_text(b"a" * 20_000)
            with pytest.raises(ss.Dc3HipError):
                g._collective(lambda r: r.build())             # every rank through dc3hip_global_build on its own thread
            assert g.selftest() == [2, 2]                      # the world is clean again
            t = oracle.gen(40_000, 9, 0)
            g.set_text(t)
            g._collective(lambda r: r.build())
            assert np.array_equal(g.sa(), want_sa(oracle, t))


@pytest.mark.parametrize("P", [3, 5, 7, 16])
def test_loopback_odd_rank_counts_and_max_ranks(ss, oracle, P):
    """Rank counts that do not divide anything evenly (splitter arithmetic, empty key ranges, digit ranges of the rank
    exchange that do not split evenly) and the maximum of 16 ranks; lengths chosen so that n % 3 and the level lengths
    hit all dummy / no-dummy combinations."""
    with env(DC3HIP_GLOBAL_LOCAL_MAX=200, DC3HIP_GLOBAL_NO_TEXT_ORDER=1):
        with ss.LoopbackGroup(P, 260_000) as g:
            for n, kind, seed in [(100_000, 2, 1), (100_001, 1, 2), (100_002, 0, 3), (259_999, 2, 4), (17_003, 1, 5), (4_099, 2, 6)]:
                t = oracle.gen(n, seed, kind)
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), want_sa(oracle, t)), (n, kind, P)
    # skewed keys: one byte value dominates, so most of a level's samples fall into ONE rank's key range
    rng = np.random.default_rng(P)
    t = np.where(rng.random(150_000) < 0.97, 97, rng.integers(98, 101, 150_000)).astype(np.uint8)
    with env(DC3HIP_GLOBAL_LOCAL_MAX=200):
        with ss.LoopbackGroup(P, len(t)) as g:
            g.set_text(t)
            g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, t))
            counts = [s["shard_count"] for s in g.stats()]
            assert sum(counts) == len(t) and max(counts) <= 3 * len(t) // P + 64      # output slices stay balanced


def test_loopback_env_matrix_agrees(ss, oracle):
    """The orderings the global level driver can take (whole-level order, image-range prefix sort, straight key-range
    sort, discarding on/off) give the same bytes."""
    t = np.concatenate([oracle.gen(1_200_000, 3, 2), oracle.gen(600_000, 2, 0), oracle.gen(1_200_000, 3, 2)[100_000:700_000]])
    want = want_sa(oracle, t)
    for extra in ({}, {"DC3HIP_NO_HYBRID": 1}, {"DC3HIP_NO_FULLSORT": 1}, {"DC3HIP_NO_DISCARD": 1},
                  {"DC3HIP_NO_HYBRID": 1, "DC3HIP_NO_DISCARD": 1}, {"DC3HIP_NO_SMALL_TIES": 1},
                  {"DC3HIP_NO_HYBRID8": 1}, {"DC3HIP_NO_HYBRID8": 1, "DC3HIP_NO_DISCARD": 1},
                  # the straight key-range sort on W-symbol windows, by the splitter ordering (threshold lowered)
                  {"DC3HIP_SSORT_MIN": 8192}, {"DC3HIP_SSORT_MIN": 8192, "DC3HIP_NO_HYBRID8": 1},
                  {"DC3HIP_SSORT_MIN": 8192, "DC3HIP_NO_HYBRID8": 1, "DC3HIP_NO_DISCARD": 1},
                  {"DC3HIP_SSORT_MIN": 8192, "DC3HIP_NO_HYBRID8": 1, "DC3HIP_NO_WIDE_WINDOW": 1}):
        with env(DC3HIP_GLOBAL_LOCAL_MAX=5000, **extra):
            with ss.LoopbackGroup(4, len(t)) as g:
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), want), extra


def test_single_rank_forced_distributed_fits_the_arena(ss, oracle):
    """One rank doing ALL the distributed work (DC3HIP_GLOBAL_FORCE_DIST=1) is the memory worst case of the level driver
    (every key range, every rank range and every exchange block is the whole level): it must fit the context's arena —
    a rank whose splitters degenerate is in the same position — and give the single-device checksum."""
    n = 48 << 20
    with env(DC3HIP_GLOBAL_FORCE_DIST=1):
        for kind, seed in ((0, 2), (2, 3), (1, 5)):
            with ss.LoopbackGroup(1, n) as g:
                g.generate(n, seed, kind)
                g.build()
                chk = g.checksum()
                st = g.stats()[0]
                assert st["local_from_level"] != 0
            with ss.Context(n) as c:
                c.generate(n, seed, kind)
                c.build()
                assert c.checksum() == chk, kind


@pytest.mark.timeout(120)
def test_loopback_rank_failure_releases_the_group(ss, oracle):
    """A rank that fails (work arena too small: -2) must not leave the others waiting in a collective: the group build
    returns the failing rank's error promptly, and the group is usable again with a smaller text."""
    n = 3_000_000
    with env(DC3HIP_ARENA_BYTES=48 << 20, DC3HIP_GLOBAL_LOCAL_MAX=1000, DC3HIP_GLOBAL_NO_TEXT_ORDER=1):
        g = ss.LoopbackGroup(4, n)
    try:
        g.set_text(oracle.gen(n, 3, 2))
        with pytest.raises(ss.Dc3HipError) as ei:
            g.build()
        assert ei.value.code == -2 and "arena" in str(ei.value)
        small = oracle.gen(40_000, 3, 2)
        g.set_text(small)
        g.build()
        assert np.array_equal(g.sa(), want_sa(oracle, small))
    finally:
        g.close()


@pytest.mark.parametrize("P,n,kind,seed,label", [(4, 1 << 30, 0, 4, "configs[3] shape: random bytes over 4 ranks (1/4 scale)"),
                                                (8, 512 << 20, 1, 5, "configs[4] shape: DNA over 8 ranks, i64 shards (1/32 scale)"),
                                                (8, 256 << 20, 1, 6, "configs[4] shape, distributed recursion (no whole-text order)")])
def test_baseline_multi_gpu_config_shapes_on_loopback(ss, P, n, kind, seed, label):
    """BASELINE.json configs[3] / configs[4] as far as one GPU can host them: the same rank counts and input classes
    through the global mode with loopback ranks (the arenas of all ranks share this GPU's 288 GB, hence the reduced
    sizes).  The shards tile [0, n), their checksums add up to the single-device checksum, and the concatenated array
    passes the GPU sufcheck; shards are fetched as int64 as configs[4] asks."""
    recursion = "recursion" in label
    with env(**({"DC3HIP_GLOBAL_NO_TEXT_ORDER": 1} if recursion else {})), ss.LoopbackGroup(P, n) as g:
        g.generate(n, seed, kind)
        g.build()
        chk = g.checksum()
        st = g.stats()
        parts, nxt = [], 0
        for r in g.ranks:
            first, s = r.shard_sa(np.int64)
            assert first == nxt and s.dtype == np.int64, label
            nxt += len(s)
            parts.append(s.astype(np.int32))
        assert nxt == n
    sa = np.concatenate(parts)
    del parts
    with ss.Context(n) as c:
        c.generate(n, seed, kind)
        c.build()
        assert c.checksum() == chk, label
        c.set_sa(sa)
        assert c.sufcheck() == 0, label
    assert all(s["comm_bytes_in"] >= n * (P - 1) // P - P for s in st)  # at least the other ranks' text blocks crossed the transport
    # random bytes by 9-byte windows, random DNA by 39-symbol windows: both finish in the distributed whole-text order
    assert (st[0]["text_order"] == 1) == (not recursion)
    if recursion:
        assert st[0]["levels"] >= 2 and st[0]["exchanges"] >= 1, (st[0]["levels"], st[0]["exchanges"])


def test_loopback_spread_over_visible_devices(ss, oracle):
    """device = DC3HIP_DEVICE_SPREAD: rank r on device r % (visible devices), peer copies as the transport (on this
    one-GPU box every rank lands on device 0; the device-spreading arithmetic and hipMemcpyDefault copies are the
    same code a multi-GPU node runs)."""
    from stringsearch_amd.global_sa import DEVICE_SPREAD
    t = oracle.gen(1_000_003, 6, 2)
    with env(DC3HIP_GLOBAL_LOCAL_MAX=2000):
        with ss.LoopbackGroup(4, len(t), device=DEVICE_SPREAD) as g:
            g.set_text(t)
            g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, t))


@pytest.mark.parametrize("P", [2, 3, 8])
def test_wide_mode_small_texts_match_the_oracle(ss, oracle, P):
    """The wide mode (64-bit positions, texts beyond DC3HIP_MAX_N) forced onto small texts so that the oracle can judge it:
    random bytes, DNA, binary, an alphabet with 0x00, texts ending in a run of the smallest symbol; bit-exact shards
    (fetched as int64), the collective verifier agrees, the u32 getter refuses.  Planted repeats of 100, 400, 20 000 and
    150 000 symbols are settled by the tie rounds (each 16 times deeper than the last); a text half of which is a copy of
    the other half goes through the deepening by rank look-ups (and is refused with -4 on every rank without it, as a text
    over one symbol always is)."""
    rng = np.random.default_rng(47)
    # (texts this small stay below the bucket ordering's threshold: the 16-byte LSD form, unrouted selection)
    with env(DC3HIP_GLOBAL_FORCE_WIDE=1), ss.LoopbackGroup(P, 3_000_000) as g:
        cases = {"bytes": rng.integers(0, 256, size=2_500_003, dtype=np.uint8), "dna": oracle.gen(3_000_000, 5, 1),
                 "binary": rng.integers(0, 2, size=1_000_001, dtype=np.uint8) + 7,
                 "with_zero_byte": rng.integers(0, 3, size=777_777, dtype=np.uint8), "tiny": rng.integers(0, 256, size=300, dtype=np.uint8)}
        d = cases["dna"].copy(); d[-40:] = d.min(); cases["dna_min_run_at_end"] = d
        for label, t in cases.items():
            g.set_text(t)
            g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, t)), (label, P)
            assert g.sufcheck() == 0, label
            st = g.stats()
            assert all(s["text_order"] == 1 and s["levels"] == 1 for s in st)
            with pytest.raises(ss.Dc3HipError) as ei:
                g.ranks[0].shard_sa(np.uint32)
            assert ei.value.code == -4
        short = cases["dna"].copy(); short[2_000_000:2_000_100] = short[5:105]        # a repeat shorter than the 256-symbol window: built
        g.set_text(short); g.build()
        assert np.array_equal(g.sa(), want_sa(oracle, short)) and g.sufcheck() == 0
        mid = cases["dna"].copy(); mid[1_000_000:1_000_400] = mid[5:405]               # a longer one: settled by the deeper second attempt
        g.set_text(mid); g.build()
        assert np.array_equal(g.sa(), want_sa(oracle, mid)) and g.sufcheck() == 0
        rep = cases["dna"].copy(); rep[1_000_000:1_020_000] = rep[5:20_005]            # beyond 8192 symbols: settled by the rounds that
        g.set_text(rep); g.build()                                                     # go 16 times deeper each (round 3)
        assert np.array_equal(g.sa(), want_sa(oracle, rep)) and g.sufcheck() == 0
        rep3 = cases["dna"].copy()                                                     # three copies of a 150 000-symbol block, one
        rep3[1_000_000:1_150_000] = rep3[5:150_005]; rep3[2_500_000:2_650_000] = rep3[5:150_005]   # of them reaching close to the end
        g.set_text(rep3); g.build()
        assert np.array_equal(g.sa(), want_sa(oracle, rep3)) and g.sufcheck() == 0
        huge = cases["dna"].copy(); huge[1_500_000:2_900_000] = huge[5:1_400_005]      # more than 2^20 positions share their window:
        g.set_text(huge); g.build()                                                    # deepened by rank look-ups (round 4)
        assert np.array_equal(g.sa(), want_sa(oracle, huge)) and g.sufcheck() == 0
        assert all(s["wide_deepen_rounds"] >= 1 for s in g.stats())
        with env(DC3HIP_WIDE_CORRUPT="1"):                                             # (the verifier of a deepened order sees a swap too)
            g.build()
        assert g.sufcheck() == -3
        for bad, switch in ((huge, "1"), (np.full(100_000, 65, dtype=np.uint8), "1")):  # refused WITHOUT the deepening (with it both build: round 5)
            with env(DC3HIP_NO_WIDE_DEEPEN=switch), ss.LoopbackGroup(P, 3_000_000) as g2:
                g2.set_text(bad)
                with pytest.raises(ss.Dc3HipError) as ei:
                    g2.build()
                assert ei.value.code == -4, ei.value
                g2.set_text(cases["tiny"]); g2.build()               # the group is usable again
                assert np.array_equal(g2.sa(), want_sa(oracle, cases["tiny"]))
        g.set_text(cases["bytes"]); g.build()                        # (and so is this one, after a deepened build)
        assert np.array_equal(g.sa(), want_sa(oracle, cases["bytes"]))
        # the verifier must see a damaged array (test hook: two neighbours swapped / one position out of range)
        for how, code in (("1", -3), ("2", -2)):
            with env(DC3HIP_WIDE_CORRUPT=how):
                g.build()
            assert g.sufcheck() == code, how
        g.build()
        assert g.sufcheck() == 0


@pytest.mark.parametrize("P", [2, 3, 5])
def test_wide_mode_bucket_ordering_on_small_texts(ss, oracle, P):
    """The wide mode's bucket ordering on 8-byte words (dc3_wide_msd.hip.hpp: a rank maps the images of its range onto
    [0, 2^E), partition pass 1 selects and computes them straight from the text and drops the bucket's own bits from the
    word, passes 2 and 3 and the tie pass work on 8-byte words) forced onto small texts (DC3HIP_WIDE_MSD_MIN=1) so that the oracle can judge it: bit-exact int64 shards, the collective verifier agrees, and the 16-byte LSD
    form (DC3HIP_NO_WIDE_MSD=1) gives the same checksum.  Repeats are settled by the same tie rounds."""
    rng = np.random.default_rng(48)
    cases = {"bytes": rng.integers(0, 256, size=2_500_003, dtype=np.uint8), "dna": oracle.gen(3_000_000, 5, 1),
             "binary": rng.integers(0, 2, size=1_000_001, dtype=np.uint8) + 7,
             "with_zero_byte": rng.integers(0, 3, size=777_777, dtype=np.uint8),
             "bytes_100k": rng.integers(0, 256, size=100_001, dtype=np.uint8)}
    d = cases["dna"].copy(); d[-40:] = d.min(); cases["dna_min_run_at_end"] = d
    rep = cases["dna"].copy(); rep[1_000_000:1_020_000] = rep[5:20_005]; cases["dna_planted_repeat"] = rep
    sums = {}
    wants = {label: want_sa(oracle, t) for label, t in cases.items()}
    for extra in ({"DC3HIP_WIDE_MSD_MIN": 1}, {"DC3HIP_NO_WIDE_MSD": 1}):
        with env(DC3HIP_GLOBAL_FORCE_WIDE=1, **extra), ss.LoopbackGroup(P, 3_000_000) as g:
            for label, t in cases.items():
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), wants[label]), (label, P, extra)
                assert g.sufcheck() == 0, label
                st = g.stats()
                assert all(s["text_order"] == 1 and s["levels"] == 1 for s in st)
                assert all(s["wide_msd"] == (0 if "DC3HIP_NO_WIDE_MSD" in extra else 1) for s in st), (label, [s["wide_msd"] for s in st])
                sums.setdefault(label, set()).add(g.checksum())
            huge = cases["dna"].copy(); huge[1_500_000:2_900_000] = huge[5:1_400_005]      # deepened by rank look-ups, on every rank
            g.set_text(huge); g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, huge)) and g.sufcheck() == 0
            g.set_text(cases["bytes"]); g.build()                        # the sort's buffers come back
            assert np.array_equal(g.sa(), wants["bytes"])
    assert all(len(v) == 1 for v in sums.values()), sums


@pytest.mark.parametrize("P", [2, 3, 5])
def test_wide_mode_deepening_by_rank_lookups(ss, oracle, P):
    """Repetitive texts in the wide mode (64-bit positions: no recursion): where windows still agree after the symbol
    compares, all ranks exchange their shards, build the inverse of the order so far and settle 17 times the depth per round
    with 17 rank look-ups per compare (wide_deepen).  Two copies of one text, eight copies of a block with a few point
    mutations, a 1.4 M-symbol repeat, a period of 5000 symbols (300 suffixes per group), repeats reaching the end of the
    text; the bucket ordering and the 16-byte LSD form; bit-exact against divsufsort, accepted by the collective verifier
    (which checks a deepened order in linear time against its own inverse).  Groups of any size (round 5: beyond 1024
    suffixes sharing a window the text used to be refused): a period of 1000 symbols (2900 suffixes per group), a text
    over ONE symbol (a single group of all suffixes), a long run of one symbol inside random text, a short period —
    ordered by the segmented LSD sort of the big groups' members (wide_big_syms / wide_big_isa)."""
    rng = np.random.default_rng(4900 + P)
    base = oracle.gen(1_200_000, 31, 1)
    cases = {"two_copies": np.concatenate([base, base])}
    blk = rng.integers(0, 256, size=300_000, dtype=np.uint8)
    eight = np.tile(blk, 8)
    for at in rng.integers(0, len(eight), size=40):
        eight[at] ^= 1
    cases["eight_copies_with_mutations"] = eight
    h = oracle.gen(3_000_000, 5, 1).copy(); h[1_500_000:2_900_000] = h[5:1_400_005]
    cases["long_repeat"] = h
    cases["period_5000"] = np.tile(rng.integers(0, 4, size=5000, dtype=np.uint8) + 65, 300)[:1_499_999]
    e = oracle.gen(2_000_000, 7, 0).copy(); e[-700_000:] = e[100:700_100]
    cases["repeat_at_the_end"] = e
    for extra in ({"DC3HIP_WIDE_MSD_MIN": 1}, {"DC3HIP_NO_WIDE_MSD": 1}):
        with env(DC3HIP_GLOBAL_FORCE_WIDE=1, **extra), ss.LoopbackGroup(P, 3_000_000) as g:
            for label, t in cases.items():
                want = want_sa(oracle, t)
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), want), (label, P, extra)
                assert g.sufcheck() == 0, (label, P)
                st = g.stats()
                if label in ("two_copies", "long_repeat", "period_5000"):       # (the others may fit the symbol compares' budget)
                    assert all(s["wide_deepen_rounds"] >= 1 for s in st), (label, [s["wide_deepen_rounds"] for s in st])
                g.build()
                assert np.array_equal(g.sa(), want), (label, P, extra, "second build")
            run = rng.integers(0, 256, size=700_001, dtype=np.uint8); run[200_000:500_000] = 78      # 300 000 x 'N'
            big = {"period_1000": np.tile(rng.integers(0, 4, size=1000, dtype=np.uint8) + 65, 2900),     # 2900 suffixes per group
                   "one_symbol": np.full(150_001, 65, dtype=np.uint8),
                   "run_of_one_symbol_in_random_text": run,
                   "period_3": np.tile(np.frombuffer(b"abc", dtype=np.uint8), 70_000)[:209_998]}
            for label, t in big.items():
                want = want_sa(oracle, t)
                g.set_text(t)
                g.build()
                assert np.array_equal(g.sa(), want), (label, P, extra)
                assert g.sufcheck() == 0, (label, P)
                assert all(s["wide_deepen_rounds"] >= 1 for s in g.stats()), label
            t = rng.integers(0, 256, size=1_000_003, dtype=np.uint8)
            g.set_text(t); g.build()
            assert np.array_equal(g.sa(), want_sa(oracle, t))
            assert all(s["wide_deepen_rounds"] == 0 for s in g.stats())


def test_wide_mode_beyond_2pow32(ss):
    """A single suffix array of more than 2^32 positions (what BASELINE.json configs[3] / configs[4] need): 2^32 + 2^20 + 3
    random bytes over two loopback ranks on this GPU; the collective verifier (range + strict suffix order across the
    whole array, shard sizes adding up to n) accepts it; builds are idempotent.  (The verifier itself is tested against
    the oracle and against refused inputs at small sizes in test_wide_mode_small_texts_match_the_oracle.)"""
    n = (1 << 32) + (1 << 20) + 3
    with ss.LoopbackGroup(2, n) as g:
        g.generate(n, 6, 0)
        g.build()
        st = g.stats()
        assert sum(s["shard_count"] for s in st) == n and st[0]["shard_first"] == 0 and st[1]["shard_first"] == st[0]["shard_count"]
        assert g.sufcheck() == 0
        chk = g.checksum()
        assert all(s["ctx"]["level_tied"][0] < n // 500 for s in st)       # 41 image bits for 2^31 records per rank: hardly any ties
        assert all(s["wide_msd"] == 1 for s in st)                         # ordered by the bucket ordering on 8-byte words
        g.build()
        assert g.checksum() == chk
        # ... and an INDEPENDENT verdict at this size: the reference's own sufcheck() built with 64-bit indices
        # (oracle/_ref/libdivsufsort64_ref.so, c-sources/utils.c:160-241) on the shards fetched to host memory (34 GB of
        # int64 + 4.3 GB of text; about 80 s of one host core).  rc 0 <=> this array is the suffix array of this text.
        import ctypes
        from conftest import ROOT
        path = os.path.join(ROOT, "oracle", "_ref", "libdivsufsort64_ref.so")
        if os.path.exists(path) and os.environ.get("DC3HIP_SKIP_SLOW_REFERENCE_CHECK") != "1":
            ref = ctypes.CDLL(path)
            ref.sufcheck.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
            ref.sufcheck.restype = ctypes.c_int32
            sa = np.zeros(n, dtype=np.int64)
            for r in g.ranks:
                first, cnt = r.shard()
                view = sa[first:first + cnt]
                assert ss.lib().dc3hip_global_get_shard_i64(r._h, view.ctypes.data) == 0
            text = np.zeros(n, dtype=np.uint8)
            with ss.Context(1 << 30) as c:
                for off in range(0, n, 1 << 30):
                    m = min(1 << 30, n - off)
                    c.generate(m, 6, 0, offset=off)
                    text[off:off + m] = c.text()
            assert int(ref.sufcheck(text.ctypes.data, sa.ctypes.data, n, 0)) == 0
