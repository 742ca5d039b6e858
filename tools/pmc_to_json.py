#!/usr/bin/env python3
"""Turn the per-kernel FETCH_SIZE / WRITE_SIZE sums of tools/pmc_build.sh (ONE build of one of bench.py's workloads) into
profiles/pmc_traffic<suffix>.json: HBM bytes per record for the kernel families bench.py reports, and the whole build's bytes.
Corrections per MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced streaming read -> streaming kernels use 2*FETCH; the random 16-byte gather is an uncalibrated access width and
is left uncorrected (lower bound).
usage: pmc_to_json.py DIR TAG     TAG in default | recursion | text | dna  (the TAG given to tools/pmc_build.sh)"""
import csv, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
tag = sys.argv[2] if len(sys.argv) > 2 else "default"
suffix = "" if tag == "default" else "_" + tag


def load(i, ctr):
    out = {}
    for row in csv.DictReader(open(f"{d}/pmc_{tag}_g{i}_by_kernel.csv")):
        out[row["kernel"]] = float(row[ctr]) * 1024.0
    return out


F, W = load(1, "FETCH_SIZE"), load(2, "WRITE_SIZE")
st = json.load(open(f"{d}/pmc_{tag}_stats.json"))


def tot(pred, fetch_mul):
    return sum(fetch_mul * F.get(k, 0) + W.get(k, 0) for k in set(F) | set(W) if pred(k))


res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, program directly after --), ONE build of the workload; "
                 "KiB*1024, streaming kernels 2*FETCH+WRITE (gfx950 half-count), gather FETCH+WRITE uncorrected",
       "workload": {"tag": tag, "n": st["level_n"][0] if st["level_n"] else None, "levels": st["levels"], "text_sort_state": st.get("text_sort_state")},
       "bytes_per_record": {}, "per_kernel_bytes": {}}
e = st["downsweep_elems"]
if e[0]: res["bytes_per_record"]["downsweep_rec8"] = tot(lambda k: "k_rs_downsweep<dc3::Rec8" in k, 2) / e[0]
if e[1]: res["bytes_per_record"]["downsweep_rec16"] = tot(lambda k: "k_rs_downsweep<dc3::Rec16" in k or "k_rs_downsweep<dc3::Rec12" in k, 2) / e[1]
if e[2]: res["bytes_per_record"]["downsweep_tup0"] = tot(lambda k: "k_rs_downsweep<dc3::Tup0" in k, 2) / e[2]
if st["gather_elems"]: res["bytes_per_record"]["gather_tuples"] = tot(lambda k: "k_gather_tuples" in k, 1) / st["gather_elems"]
if st["partition_elems"]: res["bytes_per_record"]["part_msd"] = tot(lambda k: "k_part_msd" in k or "k_tup_part" in k or "k_tup8_part" in k, 2) / st["partition_elems"]
if st.get("msd_part_elems"): res["bytes_per_record"]["msd_part"] = tot(lambda k: "k_msd_part<" in k, 2) / st["msd_part_elems"]
if st.get("msd_part_keys_elems"): res["bytes_per_record"]["msd_part_keys"] = tot(lambda k: "k_msd_part_keys" in k or "k_wide_part1" in k, 2) / st["msd_part_keys_elems"]
if st.get("msd_local_elems"): res["bytes_per_record"]["msd_local"] = tot(lambda k: "k_msd_local" in k, 2) / st["msd_local_elems"]
if st.get("ssort_part_elems"): res["bytes_per_record"]["ssort_part"] = tot(lambda k: "k_ss_part" in k, 2) / st["ssort_part_elems"]
if st.get("ssort_local_elems"): res["bytes_per_record"]["ssort_local"] = tot(lambda k: "k_ss_local" in k, 2) / st["ssort_local_elems"]
for k in sorted(set(F) | set(W), key=lambda k: -(2 * F.get(k, 0) + W.get(k, 0))):
    if "k_generate" in k or "k_check_" in k:
        continue
    res["per_kernel_bytes"][k] = {"fetch_raw": F.get(k, 0), "write": W.get(k, 0)}
try:
    import subprocess
    res["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    res["commit"] = None
sys.path.insert(0, ".")
from stringsearch_amd.benchlib import kernel_sources_sha
res["kernel_sources_sha"] = kernel_sources_sha()       # bench.py replays these figures only on exactly these sources
build = lambda k: "k_generate" not in k and "k_check_" not in k        # (the generator and the verifier are not the build)
res["whole_build_bytes_streaming_corrected"] = (sum(2 * F.get(k, 0) + W.get(k, 0) for k in set(F) | set(W) if build(k) and "gather" not in k)
                                                + tot(lambda k: "k_gather_tuples" in k, 1))
res["build_ms_under_pmc"] = st["build_ms"]
res["levels"] = list(zip(st["level_n"], st["level_sorted"]))
json.dump(res, open(f"profiles/pmc_traffic{suffix}.json", "w"), indent=1)
print(json.dumps(res["bytes_per_record"], indent=1)); print("whole build GB:", res["whole_build_bytes_streaming_corrected"] / 1e9)
