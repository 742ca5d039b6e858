"""CPU: what the generated ISA (csrc/dc3hip.gfx950.s, the assembly of the very compilation that made libdc3hip.so) says about
each kernel's occupancy on gfx950: VGPRs (512 per SIMD lane, granules of 8), LDS (160 KB per CU), scratch.  Lists the
kernels with >= 512 threads per block that fit fewer than 2 blocks per CU, and every kernel that spills.
usage: python tools/occupancy_report.py [all]        (tests/test_occupancy.py keeps the partition kernels at two blocks)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(asm=None):
    """[(demangled name, max threads per block, VGPRs, static LDS bytes, scratch bytes, blocks per CU by VGPRs)]
    (.vgpr_count of gfx90a+ metadata is the unified total, accumulation registers included: .agpr_count is not added)"""
    s = open(asm or os.path.join(ROOT, "stringsearch_amd", "csrc", "dc3hip.gfx950.s")).read()
    meta = s[s.index("amdhsa.kernels:"):]
    rows = []
    for e in re.split(r"\n  - (?=\.agpr_count)", meta)[1:]:
        def g(k, e=e):
            m = re.search(r"^\s+\." + k + r":\s+(\S+)", e, re.M)
            return m.group(1) if m else None
        name, wg = g("name"), int(g("max_flat_workgroup_size"))
        v, a = int(g("vgpr_count")), int(g("agpr_count") or 0)
        lds, sc = int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size"))
        waves = (wg + 63) // 64
        per_simd = (waves + 3) // 4
        gran = max(8, (max(v, a) + 7) // 8 * 8)
        by_v = min(8, 512 // gran) // per_simd
        rows.append((name, wg, max(v, a), lds, sc, by_v))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    return [(d,) + r[1:] for r, d in zip(rows, names)]


if __name__ == "__main__":
    show_all = len(sys.argv) > 1
    for d, wg, v, lds, sc, by_v in kernels():
        if show_all or sc > 0 or (wg >= 512 and by_v < 2):
            print(f"block<={wg:4d} vgpr {v:3d} static_lds {lds:6d} scratch {sc:4d} blocks/CU(vgpr) {by_v}  {d[:150]}")
