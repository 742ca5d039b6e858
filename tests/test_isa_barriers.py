"""CPU (cross-compile only): every s_barrier of every gfx950 kernel of the library is reached with no LDS write or atomic
of the same wave pending (tools/isa_barrier_scan.py walks the control-flow graph of the generated ISA backwards from
each barrier until an `s_waitcnt lgkmcnt(0)`).  hipcc's __syncthreads() normally guarantees it; in round 3 it did not for
a barrier inside a rolled loop whose pending ds_add_u32 came around the back-edge (the first k_ss_hist2), and the LDS
histogram was read while other SIMDs' adds were still queued: sub-bucket counts off by a few records, a GPU memory fault
three kernels later, one real-text build in eight.  The scan flags exactly that kernel in the old sources and nothing in
the current ones; this test keeps it that way for every kernel."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_barrier_is_reached_with_a_pending_lds_write(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import isa_barrier_scan as scan
    asm = tmp_path / "dc3hip.gfx950.s"
    src = os.path.join(ROOT, "stringsearch_amd", "csrc", "dc3hip.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only", "-o", str(asm), src],
                   check=True, cwd=os.path.dirname(src), timeout=900)
    kernels = flagged = 0
    names = []
    for name, body in scan.functions(str(asm)):
        kernels += 1
        if scan.scan(name, body):
            flagged += 1
            names.append(name)
    assert kernels > 200, kernels                  # the whole kernel set was seen
    assert flagged == 0, names


def test_the_scan_sees_a_pending_write():
    """the scanner on a hand-made kernel with the hazard (loop back-edge) and on its fixed form"""
    import isa_barrier_scan as scan
    bad = """
	s_waitcnt lgkmcnt(0)
	s_barrier
.LBB0_1:
	s_cbranch_scc1 .LBB0_2
	s_barrier
	ds_read_b32 v1, v2
	s_waitcnt lgkmcnt(0)
.LBB0_2:
	ds_add_u32 v3, v4
	s_cbranch_vccz .LBB0_3
	s_branch .LBB0_1
.LBB0_3:
	s_waitcnt lgkmcnt(0)
	s_barrier
	s_endpgm
""".split("\n")
    assert len(scan.scan("k", bad)) == 1
    good = [l for l in bad]
    i = good.index("\ts_cbranch_vccz .LBB0_3")
    good.insert(i, "\ts_waitcnt lgkmcnt(0)")
    assert scan.scan("k", good) == []
