"""CPU-side checks of the drop-in boundary: libspangpu.so builds for gfx950, loads,
exports every symbol include/*.h declares, and refuses to run without a GPU (no
CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header, macro="SPANGPU_API"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("#"))
    return sorted(set(re.findall(macro + r"\s+[^;(]*?\b(\w+)\s*\(", text)))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return set(ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in ("T", "W", "D", "B"))


def test_library_exports_every_declared_symbol(built):
    from spandsp_amd import engine
    L = C.CDLL(engine.LIB_PATH)
    headers = [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]
    assert "spangpu.h" in headers
    total = 0
    for h in headers:
        names = _declared_symbols(h)
        for n in names:
            assert hasattr(L, n), "%s declares %s but libspangpu.so does not export it" % (h, n)
        total += len(names)
    assert total >= 20


def test_the_spandsp_named_primitives_are_an_opt_in_library(built):
    """include/spangpu_prims.h's names (vec_*, cvec_*, power_meter_*, godard_ted_*, periodogram* ...) are what libspandsp calls from
    inside its own modules: libspangpu.so must export none of them (a process loading both would have those calls captured),
    libspangpu_prims.so every one, and nothing else."""
    from spandsp_amd import engine
    names = _declared_symbols("spangpu_prims.h", "SPANGPU_PRIMS_API")
    assert len(names) >= 30 and "vec_dot_prodf" in names and "periodogram" in names and "power_meter_update" in names
    main = _exported(engine.LIB_PATH)
    prims = _exported(engine.PRIMS_LIB_PATH)
    for n in names:
        assert n not in main, "libspangpu.so exports %s" % n
        assert n in prims, "spangpu_prims.h declares %s but libspangpu_prims.so does not export it" % n
    assert prims == set(names), sorted(prims ^ set(names))
    # and the main library exports nothing else under libspandsp's INTERNAL helper names either
    for n in main:
        assert not re.match(r"(vec_|cvec_|power_meter_|godard_ted_|periodogram|fixed_|dds_|top_bit|saturate)", n), n


def test_no_cpu_fallback(built):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    from spandsp_amd import engine
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible")
    makers = [lambda: engine.ToneBank(engine.DTMF, 64), lambda: engine.V29Bank(64, 9600), lambda: engine.V27terBank(64, 4800),
              lambda: engine.V17Bank(64, 14400), lambda: engine.EchoBank(64, 128, 1), lambda: engine.FskBank(engine.FSK_V21CH2, 64),
              lambda: engine.MctBank(engine.MCT_FAX_CED_OR_PREAMBLE, 64), lambda: engine.TxBank(engine.TX_DTMF, 64),
              lambda: engine.V29TxBank(64), lambda: engine.V27terTxBank(64), lambda: engine.V17TxBank(64),
              lambda: engine.AwgnBank([1]*64, [-30.0]*64)]
    for make in makers:
        with pytest.raises(engine.SpanGpuError) as ei:
            make()
        assert ei.value.code == -1      # SPANGPU_ERR_NO_DEVICE


def test_bad_arguments_are_refused_before_any_device_work(built):
    """Argument validation does not need a GPU: every create call names its mistake with SPANGPU_ERR_BAD_ARG (-2)
    or SPANGPU_ERR_UNSUPPORTED (-6)."""
    from spandsp_amd import engine
    bad = [lambda: engine.V29Bank(64, 1234), lambda: engine.V27terBank(64, 9600), lambda: engine.V17Bank(64, 2400),
           lambda: engine.EchoBank(64, 100, 1), lambda: engine.FskBank(engine.FSK_V21CH2, 0), lambda: engine.FskBank(99, 64),
           lambda: engine.MctBank(42, 64), lambda: engine.TxBank(99, 64), lambda: engine.V29TxBank(64, 2400),
           lambda: engine.V27terTxBank(64, 9600), lambda: engine.V17TxBank(64, 2400), lambda: engine.ToneBank(engine.DTMF, 0),
           lambda: engine.AwgnBank([], [])]
    for make in bad:
        with pytest.raises(engine.SpanGpuError) as ei:
            make()
        assert ei.value.code in (-2, -6), ei.value


def test_goertzel_fac_matches_oracle(built):
    """Host-side constant the ABI exposes (make_goertzel_descriptor, tone_detect.c:60-68)."""
    import numpy as np
    from oracle import restated as orc
    from spandsp_amd import engine
    for f in [697, 770, 852, 941, 1209, 1336, 1477, 1633, 700, 900, 1100, 1300, 1500, 1700,
              1380, 1500, 1620, 1740, 1860, 1980, 1140, 1020, 780, 660, 540, 350, 440, 397.5]:
        a = np.float32(engine.goertzel_fac(f))
        b = np.float32(orc.goertzel_fac(f))
        assert a.tobytes() == b.tobytes(), f


def test_product_does_not_import_oracle():
    """The package must never import, link or open anything under oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "spandsp_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|liboracle|libspandsp_ref|oracle/|oracle\.")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert not bad.search(text), "%s reaches into oracle/" % os.path.join(dirpath, fn)
