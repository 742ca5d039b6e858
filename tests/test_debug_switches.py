"""CPU: the one variable that carries every test / diagnosis switch of the library (DC3HIP_DEBUG) as the
Python helpers compose it: nesting, restoring, old-style variables folded in, policy variables left alone — and that the
library's own sources read no other switch from the environment (at most 12 getenv sites, the last verdict's bound)."""
import glob
import os
import re

import stringsearch_amd as ss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cur():
    return dict(tok.partition("=")[::2] for tok in filter(None, os.environ.get("DC3HIP_DEBUG", "").split(",")))


def test_debug_switches_compose_and_restore():
    before = os.environ.get("DC3HIP_DEBUG")
    with ss.debug_switches(no_text_shortcut=1, msd_min=4096):
        assert _cur() == {**_cur(), "no_text_shortcut": "1", "msd_min": "4096"}
        with ss.debug_switches(DC3HIP_NO_HYBRID=1, msd_min=8192):            # old-style names are accepted; inner values win
            cur = _cur()
            assert cur["no_hybrid"] == "1" and cur["msd_min"] == "8192" and cur["no_text_shortcut"] == "1"
        cur = _cur()
        assert "no_hybrid" not in cur and cur["msd_min"] == "4096"
        ss.debug_set("text_order12", 0)
        assert _cur()["text_order12"] == "0"                                  # "=0" must survive: it forbids, it does not force
        ss.debug_unset("text_order12")
    assert os.environ.get("DC3HIP_DEBUG") == before


def test_old_style_variables_are_folded_in_and_policy_variables_are_not():
    os.environ["DC3HIP_NO_DISCARD"] = "1"
    os.environ["DC3HIP_GLOBAL_LOCAL_MAX"] = "64"
    os.environ["DC3HIP_BENCH_BACKEND"] = "gloo"
    try:
        ss.adopt_legacy_env()
        assert "DC3HIP_NO_DISCARD" not in os.environ and _cur().get("no_discard") == "1"
        assert os.environ["DC3HIP_GLOBAL_LOCAL_MAX"] == "64" and os.environ["DC3HIP_BENCH_BACKEND"] == "gloo"
    finally:
        ss.debug_unset("no_discard")
        os.environ.pop("DC3HIP_GLOBAL_LOCAL_MAX", None)
        os.environ.pop("DC3HIP_BENCH_BACKEND", None)


def test_the_library_reads_few_variables():
    sites = []
    for f in glob.glob(os.path.join(ROOT, "stringsearch_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(ROOT, "stringsearch_amd", "csrc", "*.hip")):
        for ln in open(f):
            code = ln.split("//")[0]
            sites += re.findall(r'getenv\("([A-Z0-9_]+)"\)', code)
    assert len(sites) <= 12, sites
    assert set(sites) <= set(ss.POLICY_VARS), set(sites) - set(ss.POLICY_VARS)


def _names_in_sources():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "stringsearch_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(ROOT, "stringsearch_amd", "csrc", "*.hip")):
        names |= set(re.findall(r'dbg_(?:on|off|num|real)\("([a-z0-9_]+)"', open(f).read()))
    return names


def test_debug_names_list_equals_what_the_sources_look_up():
    assert _names_in_sources() == set(ss.api.DEBUG_NAMES), _names_in_sources() ^ set(ss.api.DEBUG_NAMES)


def test_gate_variables_of_the_test_suite_survive_adoption():
    """(round-5 advice) adopt_legacy_env used to swallow every unknown DC3HIP_* name: DC3HIP_RUN_HOSTMOCK=1 was gone before
    tests/test_hostmock.py evaluated its skipif, and DC3HIP_DEBUG gained a stray run_hostmock=1."""
    os.environ["DC3HIP_RUN_HOSTMOCK"] = "1"
    os.environ["DC3HIP_NO_SUCH_SWITCH"] = "1"
    try:
        ss.adopt_legacy_env()
        assert os.environ.get("DC3HIP_RUN_HOSTMOCK") == "1" and os.environ.get("DC3HIP_NO_SUCH_SWITCH") == "1"
        assert "run_hostmock" not in _cur() and "no_such_switch" not in _cur()
    finally:
        os.environ.pop("DC3HIP_RUN_HOSTMOCK", None)
        os.environ.pop("DC3HIP_NO_SUCH_SWITCH", None)


def test_bench_global_reads_force_wide_in_either_spelling():
    from stringsearch_amd.bench_global import global_total
    before = os.environ.get("DC3HIP_DEBUG")
    try:
        for v, want in (("global_force_wide", True), ("global_force_wide=1", True), ("msd_min=4096,global_force_wide=1", True),
                        ("global_force_wide=0", False), ("", False)):
            os.environ["DC3HIP_DEBUG"] = v
            assert global_total(1 << 20, 0)[1] is want, v
    finally:
        if before is None:
            os.environ.pop("DC3HIP_DEBUG", None)
        else:
            os.environ["DC3HIP_DEBUG"] = before


def test_header_lists_the_switches_and_there_are_at_most_30():
    """(round-5 verdict, item 9) the test-only switch surface: at most 30 names, listed in include/dc3hip.h."""
    hdr = open(os.path.join(ROOT, "include", "dc3hip.h")).read()
    block = hdr[hdr.index("test-only    ONE variable"):hdr.index("*/", hdr.index("test-only    ONE variable"))]
    listed = set()
    for line in block.splitlines():
        if re.search(r"^\s*\*\s+(structure|sorts|memory|global mode)?\s{2,}", line) and ("=" in line or "," in line or "no_" in line):
            body = re.sub(r"\(.*?\)", "", line.split("*", 1)[1])
            body = re.sub(r"^\s*(structure|sorts|memory|global mode)", "", body)
            listed |= {tok.strip().split("=")[0] for tok in body.split(",") if re.fullmatch(r"[a-z0-9_]+(=[0-9|]*)?", tok.strip())}
    assert listed == set(ss.api.DEBUG_NAMES), (listed ^ set(ss.api.DEBUG_NAMES))
    assert len(ss.api.DEBUG_NAMES) <= 30
