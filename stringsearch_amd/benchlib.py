"""Shared pieces of bench.py and the global-mode bench (roofline bookkeeping).  Host logic only."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
HBM_ACHIEVABLE_GBS = 6290.0  # float4 copy measured by the same guide ("6.29 TB/s measured (float4 copy, 79%)")
KINDS = {"random": 0, "dna": 1, "text": 2}


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


def kernel_rooflines(st_acc, steps, st_last, kernel_ms):
    """roofline objects of the two kernel families that dominate builds: the stable radix scatter
    k_rs_downsweep<Rec,...> (per record type) and the random tuple gather k_gather_tuples."""
    dsw_ms, dsw_launches, dsw_elems, g_ms, g_launches, g_elems = st_acc
    # algorithmic bytes per record-pass of the scatter = the reference's loop lib.rs:35-38: read a[i] (w) +
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
                "launches_per_step": dsw_launches[kc] / steps,
                "share_of_build_time": dsw_ms[kc] / kernel_ms,
                "moved_bytes_per_launch": 2.0 * rec_bytes * per_launch_elems,
                "moved_GBps": 2.0 * rec_bytes * per_launch_elems / (avg_ms * 1e-3) / 1e9,
                "all_record_types_ms_per_step": [x / steps for x in dsw_ms]}
    roof_gather = None
    if g_launches:
        ge = g_elems / g_launches; gms = g_ms / g_launches
        # algorithmic bytes of the gather = what the reference's merge reads at random per sample suffix
        # (lib.rs:136-162, SURVEY §8d merge row): SA12 entry (w) + position (w) + one rank (w) + 2 symbols (2c)
        alg_g = 0.0
        for lvl, mm in enumerate(st_last["level_n"]):
            if mm < 2:
                continue
            m02 = (mm + 2) // 3 + mm // 3
            alg_g += m02 * (3 * 4 + 2 * (1 if lvl == 0 else 4))
        alg_g /= (g_launches / steps)
        roof_gather = {"bound": "hbm", "kernel": "k_gather_tuples (one random tuple gather per sample suffix)",
                       "achieved": alg_g / (gms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg_g / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_launch": alg_g, "avg_launch_ms": gms,
                       "launches_per_step": g_launches / steps, "share_of_build_time": g_ms / kernel_ms,
                       "gathers_per_second_G": ge / (gms * 1e-3) / 1e9}
    # PMC traffic is NOT measured by this run (counters need rocprofv3): the per-record figures of the last
    # committed counter collection are replayed, labelled as such, and never enter `achieved`/`frac`.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        pmc = None
    if pmc is not None:
        bpr = pmc.get("bytes_per_record", {})
        if roof is not None:
            key = ("downsweep_rec8", "downsweep_rec16", "downsweep_tup0")[kc]
            if key in bpr:
                roof["traffic"] = bpr[key] * dsw_elems[kc] / dsw_launches[kc]
                roof["traffic_replayed_from"] = {"file": "profiles/pmc_traffic.json", "source": pmc.get("source"),
                                                 "commit": pmc.get("commit"), "note": "replayed per-record constant x this run's records; not a counter read of this run"}
        if roof_gather is not None and "gather_tuples" in bpr:
            roof_gather["traffic"] = bpr["gather_tuples"] * g_elems / g_launches
            roof_gather["traffic_replayed_from"] = {"file": "profiles/pmc_traffic.json", "source": pmc.get("source"), "commit": pmc.get("commit")}
    if roof is not None:
        roof["achievable_copy_GBps"] = HBM_ACHIEVABLE_GBS
        roof["moved_frac_of_achievable"] = roof["moved_GBps"] / HBM_ACHIEVABLE_GBS
    return roof, roof_gather


def path_roofline(st, ms):
    alg = algorithmic_bytes(st["level_n"])
    return {"algorithmic_bytes_per_step": alg, "device_ms_per_step": ms,
            "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "levels": list(zip(st["level_n"], st["level_K"], st["level_sorted"])),
            "phase_ms": {k: round(v, 3) for k, v in st["phase_ms"].items() if v}}


PATH_NAMES = {0: "dc3 recursion", 1: "whole-text order (all windows distinct: 9 bytes, or 3L symbols of a small alphabet; no recursion level built)",
              2: "whole-text order reused as level 1's sorted samples, then dc3 recursion", 3: "whole-text order abandoned, dc3 recursion"}


