"""The constant modem tables libspangpu builds at bank creation (spandsp_amd/csrc/modem_tables.c) against the
reference build's generated headers, frozen in tests/golden/modem_tables.npz.  Host code only: runs without a GPU."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tables_bit_identical(built):
    from spandsp_amd import engine
    g = np.load(os.path.join(GOLDEN, "modem_tables.npz"))
    t = engine.modem_tables()
    for k in ("rrc_re", "rrc_im", "sine", "sqrt_tab", "v27_4800_re", "v27_4800_im", "v27_2400_re", "v27_2400_im"):
        assert t[k].tobytes() == g[k].tobytes(), k
    assert t["godard"].tobytes() == g["godard"][:7].tobytes()
    # the trigger / step constants the V.29 bank hard-codes (src/Makefile.am:559-560)
    assert list(g["godard"][7:9]) == [1000.0, 30.0] and list(g["steps"]) == [5, 1]
