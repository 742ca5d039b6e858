"""CPU: the library's own planner (dc3hip_global_plan — the sizing rules its allocations use, no device touched) says what a
rank of BASELINE.json configs[3] (4 GiB over 4 MI355X) and configs[4] (16 GiB over 8 MI355X, 64-bit positions) needs in
HBM, ordering and deepening by rank look-ups included, and that it stays below the 288 GB of one MI355X with margin: the
first real N-GPU run cannot die on an allocation nobody computed.  The table printed here is the one in DESIGN.md section 6."""
import pytest

import stringsearch_amd as ss

GIB = 1 << 30
CASES = [("configs[3]: 4 GiB random bytes, 4 ranks", 4 * GIB, 4), ("configs[4]: 16 GiB DNA, 8 ranks", 16 * GIB, 8),
         ("1 GiB per rank, 2 ranks (bench --gpus 2)", 2 * GIB, 2), ("1 GiB per rank, 8 ranks (bench --gpus 8)", 8 * GIB, 8)]


def test_per_rank_hbm_of_the_multi_gpu_configs_fits_one_mi355x(capsys):
    rows = []
    for label, n, P in CASES:
        p = ss.global_plan(n, P)
        assert p["struct_size"] > 0 and p["total_n"] == n and p["nranks"] == P
        assert p["wide"] == (1 if n > (1 << 32) - (1 << 24) else 0)
        assert p["hbm_bytes"] == 288_000_000_000
        assert p["peak_bytes"] == p["text_bytes"] + p["context_bytes"] + p["arena_bytes"] + max(p["order_bytes"], p["deepen_bytes"])
        assert p["peak_bytes"] <= 0.9 * p["hbm_bytes"], (label, p)             # 10 % of the HBM left for the runtime and RCCL
        if p["wide"]:
            # room for groups of tied suffixes beyond 1024 members (48 bytes per member) on top of the peak
            spare = p["hbm_bytes"] * 0.95 - p["peak_bytes"]
            assert spare / p["big_group_bytes_per_member"] >= 1 << 28, (label, spare)
        rows.append((label, p))
    with capsys.disabled():
        print("\n| workload | positions | text | context + arena | ordering | deepening | peak per rank | of 288 GB |")
        print("|---|---|---|---|---|---|---|---|")
        for label, p in rows:
            gb = lambda v: f"{v / 1e9:.1f}"
            print(f"| {label} | {'64-bit' if p['wide'] else '32-bit'} | {gb(p['text_bytes'])} | {gb(p['context_bytes'] + p['arena_bytes'])} | "
                  f"{gb(p['order_bytes']) if p['wide'] else 'in the arena'} | {gb(p['deepen_bytes']) if p['wide'] else '-'} | {gb(p['peak_bytes'])} | "
                  f"{p['peak_bytes'] / p['hbm_bytes']:.0%} |")


def test_plan_rejects_bad_arguments():
    for n, P in ((-1, 2), (1 << 30, 0), (1 << 30, 17), ((1 << 40) + 1, 8)):
        with pytest.raises(ss.Dc3HipError):
            ss.global_plan(n, P)
