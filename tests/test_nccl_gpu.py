"""The multi-GPU gathers over the real RCCL backend, on one rank (all a one-GPU box allows): the collectives read send
buffers that the kernels of REAL banks wrote on the device -- the digit bytes of a DTMF bank's launches, the ERLE floats
of echo_erle_kernel, the event bytes of a V.29 bank -- and what arrives on "rank 0" is compared with what the banks
report through their own host paths.  (The world-size-2 exchange itself is tested with gloo in test_parallel.py.)"""
import ctypes
import os
import socket

import numpy as np
import pytest

import synth
from test_oracle_pin import GOLDEN, use_golden_modem_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_digit_and_record_gathers_read_what_the_detector_kernel_wrote(built, nccl_world1):
    import torch
    from spandsp_amd import engine
    from spandsp_amd.parallel import DigitGather, ResultGather
    dev = nccl_world1
    n_ch, steps, every = 4096, 60, 5
    sig, _ = synth.dtmf_channels(n_ch, 160*steps, seed=77)
    frames = torch.tensor(sig.reshape(n_ch, steps, 160).transpose(1, 0, 2).copy(), device=dev)      # [step][channel][160]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for kind in ("digits", "records"):
        bank = engine.ToneBank(engine.DTMF, n_ch)
        bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
        g = DigitGather(1, 0, n_ch, 2, dev, every=every) if kind == "digits" else ResultGather(1, 0, n_ch, 2, dev, every=every)
        want = []
        n_digits = 0
        for s in range(steps):
            g.aim(bank)
            bank.rx_device(ctypes.c_void_p(frames[s].data_ptr()), 160, 160)
            blk = bank.blocks()                 # the bank's own host path (a copy of the same records)
            g.submit(bank)
            exp_d = np.zeros((2, n_ch), np.uint8)
            exp_r = np.zeros((2, n_ch), np.uint32)
            for r in blk:
                exp_r[r["block"], r["channel"]] = np.uint32(r["hit"]) | (np.uint32(r["code"]) << 8) | (np.uint32(r["flags"]) << 16)
                if (r["flags"] & engine.BLK_CHANGE) and r["code"]:
                    exp_d[r["block"], r["channel"]] = r["code"]
                    n_digits += 1
            want.append((exp_d, exp_r, set((int(r["block"]), int(r["channel"])) for r in blk)))
            if (s + 1) % every == 0:
                g.drain()
                if kind == "digits":
                    got = g.digits().cpu().numpy()[0]          # [every, 2, n_ch]
                    for k in range(every):
                        assert np.array_equal(got[k], want[s + 1 - every + k][0]), (kind, s, k)
                else:
                    got = g.latest().cpu().numpy().view(np.uint32)[0].reshape(every, 2, n_ch)
                    for k in range(every):
                        exp = want[s + 1 - every + k]
                        for (b, c) in exp[2]:
                            assert got[k, b, c] == exp[1][b, c], (kind, s, k, b, c)
        assert n_digits > n_ch//4
        if kind == "digits":
            bank.set_digits_ring(None, 0, 0)
        else:
            bank.set_records_buffer(None, 0)
        bank.close()


def test_float_gather_of_the_erle_kernel(built, nccl_world1):
    import torch
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_paths as bp
    from spandsp_amd import engine
    from spandsp_amd.parallel import FloatGather
    dev = nccl_world1
    n_ch, nf = 2048, 25
    tx, rx = bp.synth_echo(n_ch, nf, dev, seed=0xEC41)
    clean = torch.empty(n_ch, 160, dtype=torch.int16, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank = engine.EchoBank(n_ch, bp.ECHO_TAPS, bp.ECHO_MODE)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    bank.stats(True)
    g = FloatGather(1, 0, n_ch, dev)
    fb = n_ch*160*2
    sum_rx = np.zeros(n_ch, np.float64)
    sum_clean = np.zeros(n_ch, np.float64)
    for k in range(nf):
        bank.update_device(ctypes.c_void_p(tx.data_ptr() + k*fb), ctypes.c_void_p(rx.data_ptr() + k*fb),
                           ctypes.c_void_p(clean.data_ptr()), 160, 160)
        bank.sync()
        sum_rx += (rx[k].double()**2).sum(dim=1).cpu().numpy()
        sum_clean += (clean.double()**2).sum(dim=1).cpu().numpy()
    bank.erle_device(ctypes.c_void_p(g.send.data_ptr()))       # echo_erle_kernel writes the RCCL send buffer
    bank.sync()
    g.gather()
    got = g.result().cpu().numpy()[0]
    host = bank.erle_host()
    assert np.array_equal(got.view(np.uint32), host.view(np.uint32))
    # and the figure is what the definition says: 10 log10(energy received / energy left)
    ok = (sum_rx > 0) & (sum_clean > 0)
    ref = 10.0*np.log10(sum_rx[ok]/sum_clean[ok])
    assert ok.sum() > n_ch//2
    assert np.max(np.abs(got[ok] - ref)) < 1e-3
    bank.close()


def test_bits_gather_of_a_v29_bank(built, nccl_world1):
    import torch
    from spandsp_amd import engine
    from spandsp_amd.parallel import BitsGather
    dev = nccl_world1
    use_golden_modem_tables()
    g0 = np.load(os.path.join(GOLDEN, "v29_9600.npz"))
    x = g0["amp"]
    n_ch = 512
    steps = len(x)//160
    per = 200                                   # a 160-sample frame of V.29 9600: 48 bauds x 4 bits, and room for reports
    frames = torch.tensor(np.broadcast_to(x[:steps*160].reshape(steps, 1, 160), (steps, n_ch, 160)).copy(), device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank = engine.ModemBank(engine.V29, n_ch, 9600)
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    g = BitsGather(1, 0, n_ch, per, dev)
    got_ev = []
    for s in range(steps):
        bank.rx_device(ctypes.c_void_p(frames[s].data_ptr()), 160, 160)
        g.submit(bank)
        counts, ev = g.events()
        counts = counts.cpu().numpy()[0]
        ev = ev.cpu().numpy()[0]
        host = bank.events()                    # the bank's own host path
        for c in (0, 1, n_ch//2, n_ch - 1):
            assert counts[c] == len(host[c]) and np.array_equal(ev[c, :counts[c]], host[c]), (s, c)
        got_ev.append(ev[n_ch - 1, :counts[n_ch - 1]].copy())
    assert np.array_equal(np.concatenate(got_ev), g0["events"][:sum(len(e) for e in got_ev)])
    assert sum(len(e) for e in got_ev) > 1500
    bank.close()


def test_a_gather_refuses_a_bank_on_another_stream(built, nccl_world1):
    """The ordering contract of spandsp_amd/parallel.py, enforced: a collective is ordered behind the CURRENT torch stream, so a
    bank that launches on a stream of its own (what a fresh bank does) must not have its results gathered -- the gather would
    read a frame the kernel is still writing.  It raises; with the bank on the current stream it goes through."""
    import torch
    from spandsp_amd import engine
    from spandsp_amd.parallel import FloatGather, ResultGather, require_current_stream
    dev = nccl_world1
    n_ch = 512
    sig, _ = synth.dtmf_channels(n_ch, 160, seed=78)
    frame = torch.tensor(sig, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    bank = engine.ToneBank(engine.DTMF, n_ch)               # its own stream
    assert bank.get_stream() != stream.cuda_stream
    g = ResultGather(1, 0, n_ch, 2, dev, every=1)
    g.aim(bank)
    bank.rx_device(ctypes.c_void_p(frame.data_ptr()), 160, 160)
    with pytest.raises(RuntimeError, match="current torch stream"):
        g.submit(bank)
    bank.sync()
    bank.set_stream(ctypes.c_void_p(stream.cuda_stream))
    assert bank.get_stream() == stream.cuda_stream
    g2 = ResultGather(1, 0, n_ch, 2, dev, every=1)
    g2.aim(bank)
    bank.rx_device(ctypes.c_void_p(frame.data_ptr()), 160, 160)
    g2.submit(bank)
    g2.drain()
    assert g2.latest() is not None
    require_current_stream(bank, torch.zeros(4))            # host tensors (gloo): nothing to check
    ec = engine.EchoBank(64, 128, 1)
    fg = FloatGather(1, 0, 64, dev)
    with pytest.raises(RuntimeError):
        fg.gather(ec)
    ec.set_stream(ctypes.c_void_p(stream.cuda_stream))
    fg.gather(ec)
    fg.result()
    bank.close()
    ec.close()
