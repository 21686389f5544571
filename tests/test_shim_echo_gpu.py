"""GPU tests of the spandsp-named echo canceller entry points (include/spangpu_spandsp.h, spandsp_amd/csrc/shim_echo.c)
in the reference's own calling sequence (tests/echo_tests.c:577-594): tx' = echo_can_hpf_tx(ec, tx);
clean = echo_can_update(ec, tx', rx) -- sample by sample, and the block call that does the same in one launch."""
import ctypes as C

import numpy as np
import pytest

from test_echo_gpu import make_channels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(built):
    from spandsp_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    vp, ci, i16 = C.c_void_p, C.c_int, C.c_int16
    lib.echo_can_init.restype = vp
    lib.echo_can_init.argtypes = [ci, ci]
    lib.echo_can_free.argtypes = [vp]
    lib.echo_can_flush.restype = None
    lib.echo_can_flush.argtypes = [vp]
    lib.echo_can_adaption_mode.restype = None
    lib.echo_can_adaption_mode.argtypes = [vp, ci]
    lib.echo_can_update.restype = i16
    lib.echo_can_update.argtypes = [vp, i16, i16]
    lib.echo_can_hpf_tx.restype = i16
    lib.echo_can_hpf_tx.argtypes = [vp, i16]
    lib.spangpu_echo_can_update_block.argtypes = [vp, vp, vp, vp, vp, ci, ci]
    lib.echo_can_snapshot.restype = None
    lib.echo_can_snapshot.argtypes = [vp]
    lib.spangpu_echo_can_snapshot_taps.argtypes = [vp, vp, ci]
    return lib


MODE = 0x01 | 0x02 | 0x04 | 0x20 | 0x40         # adaption, NLP, CNG, TX HPF, RX HPF


def test_sample_by_sample_sequence(L):
    from oracle import restated as orc
    tx, rx = make_channels(5, 160*40, 128, seed=9)
    tx, rx = tx[4, :160*8], rx[4, :160*8]               # the channel with a DC offset: the HPFs matter
    ec = L.echo_can_init(128, MODE)
    assert ec
    o = orc.EchoCan(128, MODE)
    n = 700
    want = o.run(tx[:n], rx[:n], True)
    got = np.zeros(n, np.int16)
    for i in range(n):
        t = L.echo_can_hpf_tx(ec, int(tx[i]))
        got[i] = L.echo_can_update(ec, t, int(rx[i]))
    assert np.array_equal(got, want)
    # flush and mode change from the host, then a block
    L.echo_can_flush(ec)
    o.flush()
    L.echo_can_adaption_mode(ec, 0x01 | 0x20)
    o.adaption_mode(0x01 | 0x20)
    want2 = o.run(tx[n:], rx[n:], True)
    m = len(tx) - n
    clean = np.zeros(m, np.int16)
    tx_out = np.zeros(m, np.int16)
    t2 = np.ascontiguousarray(tx[n:])
    r2 = np.ascontiguousarray(rx[n:])
    assert L.spangpu_echo_can_update_block(ec, t2.ctypes.data, r2.ctypes.data, clean.ctypes.data, tx_out.ctypes.data, m, 1) == 0
    assert np.array_equal(clean, want2)
    assert not np.array_equal(tx_out, t2) and abs(int(tx_out[-200:].astype(np.int64).mean())) < 60     # DC removed from what goes to the line
    L.echo_can_free(ec)
    assert not L.echo_can_init(100, MODE)               # unsupported length: NULL, like an allocation failure


def test_snapshot_keeps_the_working_tap_set(L):
    """echo_can_snapshot() (src/echo.c:376-379): a copy of tap set 0 as it stands, unchanged by what follows."""
    from oracle import restated as orc
    tx, rx = make_channels(3, 160*40, 128, seed=12)
    tx, rx = np.ascontiguousarray(tx[1, :160*20]), np.ascontiguousarray(rx[1, :160*20])
    ec = L.echo_can_init(128, MODE)
    o = orc.EchoCan(128, MODE)
    got = np.zeros(128, np.int16)
    assert L.spangpu_echo_can_snapshot_taps(ec, got.ctypes.data, 128) == 128 and not got.any()      # nothing taken yet
    n = 160*12
    clean = np.zeros(n, np.int16)
    assert L.spangpu_echo_can_update_block(ec, tx.ctypes.data, rx.ctypes.data, clean.ctypes.data, None, n, 0) == 0
    o.run(tx[:n], rx[:n], False)
    L.echo_can_snapshot(ec)
    want = o.snapshot()["taps16"][0]
    assert want.any()
    m = 160*8
    clean2 = np.zeros(m, np.int16)
    assert L.spangpu_echo_can_update_block(ec, tx[n:].ctypes.data, rx[n:].ctypes.data, clean2.ctypes.data, None, m, 0) == 0
    L.echo_can_flush(ec)                                                # the working set is cleared, the copy is not
    assert L.spangpu_echo_can_snapshot_taps(ec, got.ctypes.data, 128) == 128
    assert np.array_equal(got, want)
    short = np.zeros(40, np.int16)
    assert L.spangpu_echo_can_snapshot_taps(ec, short.ctypes.data, 40) == 40 and np.array_equal(short, want[:40])
    L.echo_can_free(ec)


def test_tx_out_equals_separate_hpf(built):
    """spangpu_echo_update_tx(): the filtered transmit samples it returns are those echo_can_hpf_tx() gives."""
    from spandsp_amd import engine
    lib = engine.lib()
    lib.spangpu_echo_update_tx.argtypes = [C.c_void_p]*5 + [C.c_int, C.c_int, C.c_longlong, C.c_int]
    lib.spangpu_echo_hpf_tx.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong]
    n_ch, n = 9, 480
    tx, rx = make_channels(n_ch, 160*40, 64, seed=10)
    tx, rx = np.ascontiguousarray(tx[:, :n]), np.ascontiguousarray(rx[:, :n])
    a = engine.EchoBank(n_ch, 64, MODE)
    b = engine.EchoBank(n_ch, 64, MODE)
    clean_a = np.zeros_like(tx)
    txo_a = np.zeros_like(tx)
    assert lib.spangpu_echo_update_tx(a.h, tx.ctypes.data, rx.ctypes.data, clean_a.ctypes.data, txo_a.ctypes.data,
                                      engine.MEM_HOST, n, n, 1) == 0
    txo_b = np.zeros_like(tx)
    assert lib.spangpu_echo_hpf_tx(b.h, tx.ctypes.data, txo_b.ctypes.data, n, n) == 0
    clean_b = b.update_host(txo_b, rx, False)
    assert np.array_equal(txo_a, txo_b)
    assert np.array_equal(clean_a, clean_b)
