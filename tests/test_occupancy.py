"""CPU: the partition kernels whose speed rests on two 1024-thread blocks per CU stay within 64 VGPRs (512 per SIMD lane,
8 waves per SIMD for two such blocks) — read off the ISA of the compilation that made the library
(tools/occupancy_report.py).  Round 4 found k_msd_part_keys' stripped instantiations and the splitter ordering's counting
kernels at 68-80 VGPRs, one block per CU, by reading that file; nothing in a functional test notices."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

TWO_BLOCKS = ("k_msd_part_keys<", "k_msd_part<", "k_ss_part<", "k_ss_count1<", "k_ss_hist2<", "k_ss_pack_count1<", "k_tup8_part1<",
              "k_ss_local<", "k_merge<", "k_tup_local<", "k_invperm_local", "k_part_msd<")


def test_partition_kernels_fit_two_blocks_per_cu():
    asm = os.path.join(ROOT, "stringsearch_amd", "csrc", "dc3hip.gfx950.s")
    if not os.path.exists(asm):
        pytest.skip("the library has not been built here (make -C stringsearch_amd/csrc writes the ISA beside it)")
    import occupancy_report as occ
    rows = occ.kernels(asm)
    assert len(rows) > 200, len(rows)
    seen = 0
    bad = []
    for name, wg, vgpr, lds, scratch, by_v in rows:
        if wg >= 1024 and any(k in name for k in TWO_BLOCKS):
            seen += 1
            if by_v < 2:
                bad.append((name[:100], vgpr))
            assert scratch <= 128, (name[:100], scratch)        # (a few spilled registers at most: the price of the bound)
    assert seen >= 20, seen
    assert not bad, bad


def test_fused_mod0_pass_fits_two_blocks_per_cu():
    """the stable radix pass that also selects the mod-0 tuples (k_rs_downsweep<Tup0C, .., Mod0LoaderC>): 512-thread blocks with a
    64 KB tile — two per CU by LDS, so its registers must allow two as well (round 5: one 1024-thread block of 106 VGPRs before)"""
    asm = os.path.join(ROOT, "stringsearch_amd", "csrc", "dc3hip.gfx950.s")
    if not os.path.exists(asm):
        pytest.skip("the library has not been built here")
    import occupancy_report as occ
    rows = [r for r in occ.kernels(asm) if "k_rs_downsweep<dc3::Tup0C, 256" in r[0]]
    assert rows, "no instantiation of the fused mod-0 pass in the ISA"
    for name, wg, vgpr, lds, scratch, by_v in rows:
        assert wg == 512 and by_v >= 2 and scratch == 0, (name[:120], wg, vgpr, scratch)
