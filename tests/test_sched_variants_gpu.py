"""The receivers' and the echo canceller's parity tests against builds of the library in which EVERY translation unit was
compiled with one LLVM instruction scheduler -- the default, max-ilp, iterative-ilp (tools/build_sched_variants.sh ->
tools/experiments/libspangpu_sched_*.so).  The product chooses a scheduler per unit by measurement (csrc/Makefile); the lanes
of a channel (a DPP quad in the modem kernels, 4 / 8 / 16 lanes in the canceller) hand data over through LDS inside one
wavefront, and a hand-over that is only right under the instruction order one scheduler happens to produce shows here as a
mismatch under another (round 4 found two such places late; this is the net for the next one).  Each build runs in a process of
its own (SPANGPU_LIB selects the library before it is loaded).  Builds that are not there are skipped, not failed: they are
made by hand, not by build()."""
import glob
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.soak]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = sorted(glob.glob(os.path.join(ROOT, "tools", "experiments", "libspangpu_sched_*.so")))


@pytest.mark.skipif(not VARIANTS, reason="no scheduler variant builds (bash tools/build_sched_variants.sh)")
@pytest.mark.parametrize("lib", VARIANTS or [None], ids=lambda p: os.path.basename(p)[len("libspangpu_sched_"):-3] if p else "none")
def test_parity_under_one_scheduler_for_every_unit(built, lib):
    env = dict(os.environ, SPANGPU_LIB=lib)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_v29_gpu.py"), os.path.join(ROOT, "tests", "test_v17_gpu.py"),
           os.path.join(ROOT, "tests", "test_v27ter_gpu.py"), os.path.join(ROOT, "tests", "test_echo_gpu.py"),
           # round 6: the receivers off their fixed points, and the cadence matcher (its tables are handed from lane to lane
           # through a wave's own LDS copy, no barrier)
           os.path.join(ROOT, "tests", "test_modem_offset_gpu.py"), os.path.join(ROOT, "tests", "test_cadence_gpu.py"),
           "-k", "matches_oracle or golden_direct or bank_parity or cadences"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(l for l in p.stdout.splitlines() if not l.startswith(("E2026", "W2026")))[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
