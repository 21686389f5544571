"""The constant modem tables libspangpu builds at bank creation (spandsp_amd/csrc/modem_tables.c) against the
reference build's generated headers, frozen in tests/golden/modem_tables.npz.  Host code only: runs without a GPU."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tables_bit_identical(built):
    from spandsp_amd import engine
    g = np.load(os.path.join(GOLDEN, "modem_tables.npz"))
    t = engine.modem_tables()
    for k in ("rrc_re", "rrc_im", "sine", "sqrt_tab", "v27_4800_re", "v27_4800_im", "v27_2400_re", "v27_2400_im",
              "v17_re", "v17_im"):
        assert t[k].tobytes() == g[k].tobytes(), k
    assert t["godard"].tobytes() == g["godard"][:7].tobytes()
    assert t["v17_godard"].tobytes() == g["v17_godard"][:7].tobytes()
    # trigger / step constants of the V.17 Godard descriptor (src/Makefile.am:490-491)
    assert list(g["v17_godard"][7:9]) == [1000.0, 100.0] and list(g["v17_steps"]) == [15, 1]
    # the trigger / step constants the V.29 bank hard-codes (src/Makefile.am:559-560)
    assert list(g["godard"][7:9]) == [1000.0, 30.0] and list(g["steps"]) == [5, 1]


def test_v17_signal_space(built):
    """Constellations and soft-decision maps: CRC-32s recorded from the reference build; entry by entry when it is here."""
    import oracle
    from test_oracle_pin import v17_signal_space
    t = v17_signal_space()                  # asserts the CRCs
    if oracle.have_ref():
        from oracle import ref
        r = ref.v17_signal_space()
        for k in r:
            assert np.array_equal(np.asarray(t[k]), r[k]), k
