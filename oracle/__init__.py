"""oracle -- TEST INFRASTRUCTURE ONLY.

CPU checkers for the hot path.  Nothing under ``spandsp_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do.

* ``oracle.restated``  -- ctypes binding of ``oracle/liboracle.so`` (our plain-C
  restatement, ``oracle/*.c``).
* ``oracle.ref``       -- ctypes binding of ``oracle/_ref/libspandsp_ref.so`` (the
  real reference compiled from /root/reference by ``oracle/Makefile``), when
  that file exists.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libspandsp_ref.so")
# the reference as it ships (-O2 -ffast-math -msse2, SPANDSP_USE_SSE2): for cpu_baseline timing only, never for parity
REF_FAST_SO = os.path.join(HERE, "_ref", "libspandsp_ref_fast.so")


def build(verbose=False):
    """Compile liboracle.so, and _ref/ when /root/reference is present."""
    out = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed")


def have_ref():
    return os.path.exists(REF_SO)


def have_ref_fast():
    return os.path.exists(REF_FAST_SO)
