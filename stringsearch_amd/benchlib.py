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


def kernel_sources_sha():
    """sha256 over the sources the timed kernels and their orchestration are compiled from: every csrc/*.hpp of the single-device
    path (device kernels *.hip.hpp, the sorts / orderings / level driver dc3_host_*.hpp).  Not covered: dc3hip.hip (the C ABI
    and the plumbing of the one-shot calls) and dc3_global_*.hpp (the multi-rank host driver) — neither decides which kernels
    a single-device build runs or what they move.  A PMC collection is only replayed into `roofline.traffic` when it was made
    on exactly these sources (profiles/pmc_traffic.json carries the hash of the tree it profiled): the GPU box has no .git, so
    a commit id cannot be checked there."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "stringsearch_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hpp"))):
        if os.path.basename(f).startswith("dc3_global_"):
            continue
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class KernelAcc:
    """Per-kernel-family HIP-event times summed over the timed steps (dc3hip_stats of every build)."""
    FAMS = ("downsweep0", "downsweep1", "downsweep2", "gather", "partition", "msd_part", "msd_part_keys", "msd_local", "ssort_part", "ssort_local")

    def __init__(self):
        self.ms = {k: 0.0 for k in self.FAMS}
        self.launches = {k: 0 for k in self.FAMS}
        self.elems = {k: 0 for k in self.FAMS}
        self.build_ms = 0.0

    def add(self, st):
        self.build_ms += st["build_ms"]
        for k in range(3):
            self._add(f"downsweep{k}", st["downsweep_ms"][k], st["downsweep_launches"][k], st["downsweep_elems"][k])
        self._add("gather", st["gather_ms"], st["gather_launches"], st["gather_elems"])
        self._add("partition", st["partition_ms"], st["partition_launches"], st["partition_elems"])
        self._add("msd_part", st.get("msd_part_ms", 0.0), st.get("msd_part_launches", 0), st.get("msd_part_elems", 0))
        self._add("msd_local", st.get("msd_local_ms", 0.0), st.get("msd_local_launches", 0), st.get("msd_local_elems", 0))
        self._add("msd_part_keys", st.get("msd_part_keys_ms", 0.0), st.get("msd_part_keys_launches", 0), st.get("msd_part_keys_elems", 0))
        self._add("ssort_part", st.get("ssort_part_ms", 0.0), st.get("ssort_part_launches", 0), st.get("ssort_part_elems", 0))
        self._add("ssort_local", st.get("ssort_local_ms", 0.0), st.get("ssort_local_launches", 0), st.get("ssort_local_elems", 0))

    def _add(self, k, ms, launches, elems):
        self.ms[k] += ms; self.launches[k] += launches; self.elems[k] += elems


# family -> (kernel name, algorithmic bytes per record, physically moved bytes per record, key in pmc_traffic.json)
# Algorithmic bytes of a scatter pass = the reference's loop lib.rs:35-38: read a[i] (w) + r[a[i]] (c) + write b[..] (w) =
# 2w + c = 12 B at w = c = 4 (scatter half) — the bucket partition pass does that loop's work for one
# digit exactly like a stable pass does, so it is priced the same; the in-LDS local order replaces the remaining passes of
# the sort in one launch and is priced as ONE such pass (a lower bound of what it replaces).
_FAM = {
    "downsweep0": ("k_rs_downsweep<Rec8> (stable 8/9-bit-digit radix scatter of 8-byte (key,value) records)", 12.0, 16.0, "downsweep_rec8"),
    "downsweep1": ("k_rs_downsweep<Rec12/Rec16> (stable radix scatter of triple records)", 12.0, 32.0, "downsweep_rec16"),
    "downsweep2": ("k_rs_downsweep<Tup0> (stable radix scatter of mod-0 tuples)", 12.0, 40.0, "downsweep_tup0"),
    "partition": ("k_part_msd (window partition of (destination,value) pairs: inverse permutations)", 8.0, 16.0, "part_msd"),
    "msd_part": ("k_msd_part (bucket partition of the prefix-sort words: non-stable radix scatter, XCD-grouped reservation)", 12.0, 16.0, "msd_part"),
    # pass 1 that makes its words from the text (the pack kernel only counted): Step 0 of the level (lib.rs:62-70: one
    # index written per position, w = 4) + the scatter loop of one radix pass (12) in one launch = 16 algorithmic bytes;
    # it physically reads 1 text byte and writes one 8-byte word per position
    "msd_part_keys": ("k_msd_part_keys (bucket partition pass 1 that also makes the words from the text: Step 0 + one radix pass, "
                      "XCD-grouped reservation; image d1 bits wider than the word, the bucket's own bits dropped)", 16.0, 9.0, "msd_part_keys"),
    "msd_local": ("k_msd_local (in-LDS order of the sub-buckets: the last passes of the sort in one launch)", 12.0, 16.0, "msd_local"),
    # splitter ordering of the 12- / 16-byte sample-triple records (text levels): the same scatter loop, moved bytes 2 x 16
    # (+ 2 for the digit side array in the partition pass); priced like one radix pass each
    "ssort_part": ("k_ss_part (partition of the sample-triple records over sampled splitters, XCD-grouped reservation)", 12.0, 34.0, "ssort_part"),
    "ssort_local": ("k_ss_local (in-LDS comparison order of the sub-buckets: the last passes of the sort in one launch)", 12.0, 32.0, "ssort_local"),
}


def load_pmc(pmc_name):
    """(profiles/<pmc_name> or None, collected on exactly these kernel sources?)"""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
    except Exception:
        return None, False
    return pmc, pmc.get("kernel_sources_sha") == kernel_sources_sha()


def kernel_rooflines(acc, steps, st_last, kernel_ms=None, pmc_name="pmc_traffic.json"):
    """roofline objects of the kernel families of a build, keyed by family; `dominant` = the one with the largest share
    of the build time.  achieved = algorithmic bytes per launch / average launch duration (HIP events recorded on the
    build's own stream around every launch).  pmc_name: the PMC collection of THIS workload (default build, recursion only,
    text) whose per-record HBM bytes are replayed into `traffic`."""
    kernel_ms = kernel_ms if kernel_ms is not None else acc.build_ms
    kernel_ms = max(kernel_ms, 1e-9)
    pmc, pmc_ok = load_pmc(pmc_name)
    out = {}
    for fam, (name, alg_b, moved_b, pkey) in _FAM.items():
        if not acc.launches[fam]:
            continue
        per_launch = acc.elems[fam] / acc.launches[fam]
        avg_ms = acc.ms[fam] / acc.launches[fam]
        achieved = alg_b * per_launch / (avg_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBS, "traffic": None,
             "algorithmic_bytes_per_launch": alg_b * per_launch, "avg_launch_ms": avg_ms,
             "launches_per_step": acc.launches[fam] / steps, "share_of_build_time": acc.ms[fam] / kernel_ms,
             "moved_bytes_per_launch": moved_b * per_launch,
             "moved_GBps": moved_b * per_launch / (avg_ms * 1e-3) / 1e9,
             "achievable_copy_GBps": HBM_ACHIEVABLE_GBS}
        r["moved_frac_of_achievable"] = r["moved_GBps"] / HBM_ACHIEVABLE_GBS
        if pmc_ok and pkey in pmc.get("bytes_per_record", {}):
            r["traffic"] = pmc["bytes_per_record"][pkey] * per_launch
            r["traffic_from"] = {"file": "profiles/" + pmc_name, "source": pmc.get("source"), "commit": pmc.get("commit"),
                                 "kernel_sources_sha": pmc.get("kernel_sources_sha"),
                                 "note": "per-record bytes of the rocprofv3 --pmc passes made on exactly these kernel sources x this run's records per launch"}
        out[fam] = r
    if acc.launches["gather"]:
        ge = acc.elems["gather"] / acc.launches["gather"]; gms = acc.ms["gather"] / acc.launches["gather"]
        # algorithmic bytes of the gather = what the reference's merge reads at random per sample suffix
        # (lib.rs:136-162, SURVEY §8d merge row): SA12 entry (w) + position (w) + one rank (w) + 2 symbols (2c)
        alg_g = 0.0
        for lvl, mm in enumerate(st_last["level_n"]):
            if mm < 2:
                continue
            m02 = (mm + 2) // 3 + mm // 3
            alg_g += m02 * (3 * 4 + 2 * (1 if lvl == 0 else 4))
        alg_g /= (acc.launches["gather"] / steps)
        r = {"bound": "hbm", "kernel": "k_gather_tuples (one random tuple gather per sample suffix)",
             "achieved": alg_g / (gms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": alg_g / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
             "algorithmic_bytes_per_launch": alg_g, "avg_launch_ms": gms,
             "launches_per_step": acc.launches["gather"] / steps, "share_of_build_time": acc.ms["gather"] / kernel_ms,
             "gathers_per_second_G": ge / (gms * 1e-3) / 1e9}
        if pmc_ok and "gather_tuples" in pmc.get("bytes_per_record", {}):
            r["traffic"] = pmc["bytes_per_record"]["gather_tuples"] * ge
        out["gather"] = r
    dominant = max(out.values(), key=lambda r: r["share_of_build_time"]) if out else None
    if dominant is not None and pmc is not None and not pmc_ok:
        dominant["traffic_note"] = ("profiles/%s was collected on other kernel sources (sha %s, these are %s): not replayed"
                                    % (pmc_name, pmc.get("kernel_sources_sha"), kernel_sources_sha()))
    return dominant, out


def path_roofline(st, ms, pmc_name="pmc_traffic.json"):
    """Whole-build figure.  A build that ran DC3 levels is priced by SURVEY §8(d) — B(n_l) = n_l (46 w + 29 c_l) / 3 summed
    over the levels that really ran — and its fraction of the HBM peak follows from those ALGORITHMIC bytes.  A build that
    finished in the whole-text order ran no DC3 level: §8(d)'s 71 B per input byte is not its work (priced that way it
    would "move" more than the 8 TB/s peak), so it is priced by the bytes it MOVED — the PMC collection of this workload on
    exactly these kernel sources, or, without one, the kernels' design traffic — and says so in `priced_by`."""
    pmc, pmc_ok = load_pmc(pmc_name)
    moved = pmc.get("whole_build_bytes_streaming_corrected") if (pmc is not None and pmc_ok) else None
    out = {"device_ms_per_step": ms,
           "levels": list(zip(st["level_n"], st["level_K"], st["level_sorted"])),
           "phase_ms": {k: round(v, 3) for k, v in st["phase_ms"].items() if v},
           "moved_bytes_per_step": moved,
           "moved_bytes_from": ("profiles/" + pmc_name + " (rocprofv3 FETCH_SIZE / WRITE_SIZE of one build of this workload, gfx950 corrections)") if moved else None}
    if moved:
        out["moved_GBps"] = moved / (ms * 1e-3) / 1e9
        out["moved_frac_of_hbm_peak"] = out["moved_GBps"] / HBM_PEAK_GBS
    if st.get("text_sort_state", 0) == 1:
        n = st["level_n"][0] if st["level_n"] else 0
        # design traffic of the whole-text order, per position: count pass 1 B, pass 1 from the text 1 + 8, pass 2 8 + 8, local
        # order 8 + 5, tie pass 1 (+ gathers for the tied)  = 40 B
        design = 40.0 * n
        use = moved if moved else design
        out.update({"priced_by": "moved bytes (whole-text order: no DC3 level ran, SURVEY 8(d)'s per-level bytes do not apply)" + ("" if moved else " — design traffic, no PMC collection on these sources"),
                    "algorithmic_bytes_per_step": None, "design_bytes_per_step": design,
                    "achieved_GBps": use / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": use / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    else:
        alg = algorithmic_bytes(st["level_n"])
        out.update({"priced_by": "SURVEY 8(d): B(n_l) = n_l (46 w + 29 c_l) / 3 over the DC3 levels this build ran",
                    "algorithmic_bytes_per_step": alg,
                    "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    return out


def build_block(ss, ctx, n_builds, pmc_name):
    """Timed builds of a prepared context -> a bench block with its own `roofline` (dominant kernel family), every family
    and the whole-build figure, priced with the PMC collection of this workload."""
    acc = KernelAcc()
    ms = []
    for _ in range(n_builds):
        ctx.build()
        st = ctx.stats()
        acc.add(st); ms.append(st["build_ms"])
    m = sum(ms) / len(ms)
    roof, roof_all = kernel_rooflines(acc, n_builds, st, acc.build_ms, pmc_name)
    return st, m, {"roofline": roof, "roofline_kernels": roof_all, "roofline_path": path_roofline(st, m, pmc_name)}


PATH_SHORT = {0: "dc3 recursion", 1: "whole-text order (all windows distinct, no recursion level built)",
              2: "whole-text order reused as level 1's sorted samples, then dc3 recursion", 3: "whole-text order abandoned, dc3 recursion"}
PATH_NAMES = {0: "dc3 recursion", 1: "whole-text order (all windows distinct: 9 bytes, or 3L symbols of a small alphabet; no recursion level built)",
              2: "whole-text order reused as level 1's sorted samples, then dc3 recursion", 3: "whole-text order abandoned, dc3 recursion"}


# ---- the ONE line bench.py prints ---------------------------------------------------------------------------------------
# The reference's harness prints a three-row table (crates/divsuftest/src/main.rs:168-188); a harness must be able to read
# ours.  Round 5's line had grown to 31 KB and the driver could no longer parse it: the last stdout line is now a COMPACT
# object (< MAX_LINE_BYTES, strict JSON: no NaN / Infinity), everything else goes to a side file named in `detail`.
MAX_LINE_BYTES = 4096


def _r(x, sig=6):
    """numbers to `sig` significant digits (ints stay ints); non-finite floats become None (strict JSON)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _cut(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "…"


def _roof(r):
    """the roofline object of the contract, without the prose"""
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
            "launches_per_step", "share_of_build_time")
    out = {"kernel": _cut(r.get("kernel", "?"), 80)}
    out.update({k: r.get(k) for k in keep if k in r})
    return out


def _roof_path(rp):
    if not rp:
        return None
    by = rp.get("priced_by", "")
    return {"frac": rp.get("frac_of_hbm_peak"), "achieved_GBps": rp.get("achieved_GBps"),
            "priced_by": "moved bytes (PMC)" if by.startswith("moved") and rp.get("moved_bytes_per_step") else
                         ("design traffic (no PMC on these sources)" if by.startswith("moved") else "SURVEY 8(d) algorithmic bytes"),
            "bytes_per_step": rp.get("algorithmic_bytes_per_step") or rp.get("moved_bytes_per_step") or rp.get("design_bytes_per_step"),
            "moved_bytes_per_step": rp.get("moved_bytes_per_step"), "device_ms_per_step": rp.get("device_ms_per_step")}


def compact_line(full, detail_path=None):
    """full bench record (every block bench.py measured) -> the compact object of the last stdout line."""
    c = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    cfg = full.get("config", {})
    c["config"] = {"workload": _cut(cfg.get("workload", ""), 200)}
    for k in ("bytes_per_gpu", "total_bytes", "clipped_to_DC3HIP_MAX_N"):
        if k in cfg:
            c["config"][k] = cfg[k]
    if "partitioning" in cfg:
        c["config"]["partitioning"] = _cut(cfg["partitioning"], 120)
    if "value_mode" in full:
        c["value_mode"] = _cut(full["value_mode"], 100)
    p = full.get("path", {})
    c["path"] = {"taken": _cut(PATH_SHORT.get(p.get("text_sort_state"), p.get("taken", "?")), 90), "levels": p.get("levels")}
    if "text_sort_state" in p:
        c["path"]["text_sort_state"] = p["text_sort_state"]
    c["value_MiBps"] = full.get("value_MiBps")
    c["roofline"] = _roof(full.get("roofline"))
    c["roofline_path"] = _roof_path(full.get("roofline_path"))
    ro = full.get("dc3_recursion_only")
    if ro:
        rp = ro.get("roofline_path") or {}
        c["dc3_recursion_only"] = {"ms": ro.get("device_ms_per_step"), "MBps": ro.get("MBps"), "frac": rp.get("frac_of_hbm_peak"),
                                   "levels": ro.get("levels"), "sufcheck": ro.get("sufcheck"), "checksum_equal": ro.get("checksum_equal"),
                                   "roofline": {k: v for k, v in (_roof(ro.get("roofline")) or {}).items()
                                                if k in ("kernel", "frac", "avg_launch_ms", "traffic", "algorithmic_bytes_per_launch")}}
    pc = full.get("per_config")
    if pc:
        c["per_config"] = {}
        for name, b in pc.items():
            if "skipped" in b:
                c["per_config"][name] = {"skipped": _cut(b["skipped"], 60)}
                continue
            rp = b.get("roofline_path") or {}
            e = {"ms": b.get("ms"), "MBps": b.get("MB/s"), "sufcheck": b.get("sufcheck"), "levels": b.get("levels")}
            if rp:
                e["frac"] = rp.get("frac_of_hbm_peak"); e["moved_bytes"] = rp.get("moved_bytes_per_step")
            c["per_config"][name] = e
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        c["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "seconds", "host_cpu", "host_cores_available") if k in cb}
        c["cpu_baseline"]["sample"] = _cut(cb.get("sample", ""), 200)
    c["verify"] = full.get("verify")
    if "e2e_ffi" in full:
        e = full["e2e_ffi"]
        c["e2e_ffi"] = {k: e.get(k) for k in ("ms", "MB/s", "ms_is", "in_bench_process_ms", "first_call_ms", "first_call_fresh_process_ms", "first_call_in_this_process_ms", "pcie_floor_ms") if k in e}
    gl = full.get("global_mode_loopback")
    if gl:
        # predicted speed-up over one GPU at P ranks (own work under the device token + link model); full rows in the detail file
        c["global_mode_predicted_speedup"] = {f"{_cut(r['input'], 24)} x{r['ranks']}": r.get("predicted_speedup_over_one_gpu") for r in gl}
    gb = full.get("global_mode_beyond_2pow32")
    if gb:
        c["global_mode_beyond_2pow32"] = {k: gb.get(k) for k in ("wall_ms", "global_sufcheck", "shards_tile_0_n", "skipped") if k in gb}
    ic = full.get("interconnect")
    if ic:
        c["interconnect"] = {"transport": _cut(ic.get("transport", ""), 80), "comm_ms": ic.get("comm_ms"),
                             "bytes_in_max_per_step": max(ic.get("bytes_in_per_rank_per_step") or [0]),
                             "achieved_GBps_in_slowest_rank": ic.get("achieved_GBps_in_slowest_rank"), "peak_GBps_in": ic.get("peak_GBps_in")}
    sp = full.get("sacapart")
    if sp:
        c["sacapart"] = {"value": sp.get("value"), "unit": sp.get("unit"), "ms_per_step": sp.get("ms_per_step"),
                         "roofline_frac": (sp.get("roofline") or {}).get("frac"), "verify": sp.get("verify")}
    if "value_sacapart" in full:
        c["value_sacapart"] = full["value_sacapart"]
    gm = full.get("global_mode")
    if gm:
        c["global_mode"] = {"error": _cut(gm.get("error", ""), 300)}
    ts = full.get("transport_selftest")
    if ts:
        c["transport_selftest"] = {"passed": ts.get("passed"), "transport": _cut(ts.get("transport", ""), 80),
                                   "ranks_seen_by_transport": ts.get("ranks_seen_by_transport"), "world_size": ts.get("world_size")}
    for k in ("xcd_grouping_effective", "arena_peak_GB", "hip_runtime"):
        if k in full:
            c[k] = full[k]
    c["detail"] = detail_path
    c = _r(c)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    # (cannot happen with the fields above; if a future field makes it happen, drop the optional blocks rather than the line)
    for k in ("global_mode_predicted_speedup", "per_config", "e2e_ffi", "global_mode_beyond_2pow32", "sacapart", "interconnect"):
        if len(line.encode()) < MAX_LINE_BYTES:
            break
        c.pop(k, None)
        c["dropped_for_size"] = c.get("dropped_for_size", []) + [k]
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(line.encode()) < MAX_LINE_BYTES, len(line)
    return line


def write_detail(full, where=None):
    """the full record (every block, every kernel family) beside the compact line; returns the path written (relative to the
    repository unless `where` names another place) or None"""
    for path in ([where] if where else []) + [os.path.join("gpurun_out", "bench_detail.json"), "bench_detail.json"]:
        try:
            os.makedirs(os.path.dirname(os.path.join(ROOT, path)) or ".", exist_ok=True)
            with open(os.path.join(ROOT, path), "w") as f:
                json.dump(_r(full, 9), f, indent=1, allow_nan=False, default=str)
            return path
        except Exception:
            continue
    return None

