"""CPU: the HOST half of libdc3hip under AddressSanitizer and ThreadSanitizer on a mock HIP runtime (tools/hostmock: the
library's own translation unit compiled by hipcc with the sanitizer on the host pass only, linked against hip_mock.cpp —
no GPU, no GPU sanitizer).  Kernels do not run, so every build beyond n = 2 fails at its first device-to-host read; what runs
for real is everything that owns host memory and threads: loopback groups, persistent rank threads, contexts, the pinned
pool, the one-shot cache, failed collectives, teardown in any order, from several host threads at once.

Two sanitized compilations of the translation unit take about two minutes: the test runs when DC3HIP_RUN_HOSTMOCK=1 (the
round's run is profiles/r05c_hostmock_asan_tsan_lifecycle.log); without it only the pieces are checked for presence."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tools", "hostmock")


def test_hostmock_sources_are_there():
    for f in ("hip_mock.cpp", "lifecycle.cpp", "Makefile"):
        assert os.path.exists(os.path.join(MOCK, f)), f
    mk = open(os.path.join(MOCK, "Makefile")).read()
    assert "-fsanitize=$*" in mk and "-fno-gpu-sanitize" in mk          # host pass only: never a GPU sanitizer


@pytest.mark.skipif(os.environ.get("DC3HIP_RUN_HOSTMOCK") != "1", reason="two sanitized compilations (2 min): set DC3HIP_RUN_HOSTMOCK=1")
@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_lifecycle_fuzz_is_clean_under_the_sanitizer(san):
    p = subprocess.run(["make", "-C", MOCK, san, "ITERS=60", "WORKERS=4"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["groups"] > 50 and res["contexts"] > 50 and res["builds_failed_as_expected"] > 100, res
