"""The boundary as a C compiler sees it.

Every other test binds the library through ctypes, which checks no prototype: a header that does not compile as C99, or a
prototype that drifted from the reference's, would pass them all.  Here

  * tests/c_callers/*.c (own code: a DTMF loop-back through dtmf_tx() / dtmf_rx(), a V.29 page through v29_rx() with a put_bit
    callback, the echo canceller's block call, and every public header in one unit) are compiled `gcc -std=c99 -pedantic -Wall
    -Wextra -Werror` and as C++ against include/ alone and linked -lspangpu_prims -lspangpu;
  * in the build container, where /root/reference is present: every spandsp-named prototype of include/spangpu_spandsp.h and
    include/spangpu_prims.h is held against the reference's own declaration of that name by a C compiler -- a translation unit
    that includes only the REFERENCE's headers and initialises, for every name, a function pointer declared with OUR
    prototype's text from the reference's function (-Werror=incompatible-pointer-types): argument order, argument types,
    return type;
  * on the GPU box the three programs run (tests marked gpu)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_callers")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "spandsp_amd")
PROGRAMS = ["headers", "dtmf_loopback", "v29_page", "echo_block"]
REF = "/root/reference/src"


def run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, "%s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr)
    return p.stdout


def build(name, out_dir):
    src = os.path.join(SRC, name + ".c")
    obj = os.path.join(out_dir, name + ".o")
    exe = os.path.join(out_dir, name)
    run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-c", src, "-o", obj])
    # the same source as C++ (no -pedantic there: struct filter_s ends in a flexible array member, as the reference's does)
    run(["g++", "-std=c++11", "-x", "c++", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-c", src, "-o", obj + "pp"])
    run(["gcc", "-o", exe, obj, "-L" + LIBDIR, "-lspangpu_prims", "-lspangpu", "-lm", "-Wl,-rpath," + LIBDIR])
    return exe


@pytest.mark.parametrize("name", PROGRAMS)
def test_c_callers_compile_and_link(built, tmp_path, name):
    exe = build(name, str(tmp_path))
    if name == "headers":
        run([exe])                                     # needs no device: loads both libraries, asks for the device count


def _prototypes(header, macro):
    """(name, 'ret (*chk_name)(args)') for every spandsp-named function the header declares"""
    text = open(os.path.join(INC, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"SPANGPU_SLOW_CALL\(\"[^\"]*\"\)", "", text)       # (an attribute behind two prototypes: not part of the type)
    out = []
    for m in re.finditer(macro + r"\s+([^;{}]*?)\b(\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if name.startswith("spangpu_"):
            continue
        out.append((name, ret, args))
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's headers are not here (GPU box)")
def test_prototypes_are_the_references(built, tmp_path):
    protos = _prototypes("spangpu_spandsp.h", "SPANGPU_API") + _prototypes("spangpu_prims.h", "SPANGPU_PRIMS_API")
    names = [p[0] for p in protos]
    assert len(protos) > 150 and "dtmf_rx" in names and "v29_rx_init" in names and "echo_can_update" in names and "periodogram" in names
    # not declared by the reference under that name in a public header: the library's own additions to the spandsp-named set
    own = set()
    lines = ["#include <stdlib.h>", "#include <inttypes.h>", "#include <string.h>", "#include <stdio.h>", "#include <math.h>", "#include <stdbool.h>",
             "#include <limits.h>"]
    for h in ("telephony", "alloc", "logging", "fast_convert", "queue", "complex", "dds", "tone_detect", "tone_generate", "super_tone_rx", "dtmf",
              "bell_r2_mf", "saturated", "dc_restore", "bit_operations", "echo", "async", "power_meter", "vector_float", "complex_vector_float", "godard", "v29rx", "v29tx", "v27ter_rx",
              "v27ter_tx", "v17rx", "v17tx", "awgn", "g711", "fsk", "modem_connect_tones", "sig_tone", "complex_filters", "math_fixed", "arctan2"):
        lines.append('#include "spandsp/%s.h"' % h)
    ref_text = ""
    for h in os.listdir(os.path.join(REF, "spandsp")):
        if h.endswith(".h"):
            ref_text += open(os.path.join(REF, "spandsp", h), errors="ignore").read()
    checked = 0
    for name, ret, args in protos:
        if not re.search(r"SPAN_DECLARE(_NONSTD)?\([^)]*\)\s*" + name + r"\s*\(", ref_text):
            own.add(name)
            continue
        lines.append("static %s (*chk_%s)(%s) = %s;" % (ret, name, args, name))
        checked += 1
    lines.append("int main(void) { return 0; }")
    src = os.path.join(str(tmp_path), "proto_check.c")
    open(src, "w").write("\n".join(lines) + "\n")
    defs = ["-DHAVE_MATH_H", "-DHAVE_STDBOOL_H", "-DHAVE_SINF", "-DHAVE_COSF", "-DHAVE_TANF", "-DHAVE_ASINF", "-DHAVE_ACOSF", "-DHAVE_ATANF",
            "-DHAVE_ATAN2F", "-DHAVE_CEILF", "-DHAVE_FLOORF", "-DHAVE_POWF", "-DHAVE_EXPF", "-DHAVE_LOGF", "-DHAVE_LOG10F", "-DHAVE_LRINT", "-DHAVE_LRINTF",
            "-DHAVE_LONG_DOUBLE", "-DHAVE_STDLIB_H", "-DHAVE_STRING_H", "-DHAVE_INTTYPES_H", "-DHAVE_STDINT_H", "-DHAVE_TGMATH_H"]
    run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Werror", "-Werror=incompatible-pointer-types", "-Wno-unused-variable", "-Wno-unused-function"] + defs + ["-I" + REF, src])
    assert checked > 140, (checked, sorted(own))
    # what is ours alone under a spandsp-looking name must be known: nothing slips through unnoticed
    assert own <= {"goertzel_state_t"} | OWN_NAMES, sorted(own - OWN_NAMES)


# names in the spandsp-named headers that the reference does not declare as a public function: none expected beyond these
OWN_NAMES = set()


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROGRAMS[1:])
def test_c_callers_run(built, tmp_path, name):
    exe = build(name, str(tmp_path))
    out = run([exe])
    assert name in out
