"""CPU (gloo, world_size 2) tests of the multi-GPU plumbing in spandsp_amd/parallel.py: channel sharding, the
double-buffered gather of result records to rank 0 -- through copy_records() and through the zero-copy aim() /
set_records_buffer() path bench.py --gpus N uses -- and the gather of per-channel floats (the ERLE result of the echo
configuration).  Only the kernel is stood in for (there is no GPU here, and the library has no CPU path): the stand-in
bank writes the record words a launch would write, to wherever the gather aimed it."""
import ctypes
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_covers_everything():
    from spandsp_amd.parallel import shard_range
    for total in (1, 7, 64, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class FakeBank:
    """Stands in for engine.ToneBank.copy_records(): writes the step's record words."""

    def __init__(self, rank, n):
        self.rank = rank
        self.n = n
        self.step = 0

    def copy_records(self, dst_ptr, dst_bytes):
        words = (np.arange(self.n, dtype=np.int64)*3 + self.rank*1000003 + self.step*17).astype(np.int32)
        assert dst_bytes >= words.nbytes
        ctypes.memmove(dst_ptr, words.ctypes.data, words.nbytes)
        self.step += 1
        return words.nbytes


def _worker(rank, world, port, n_ch, steps, out, every=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spandsp_amd.parallel import ResultGather
    g = ResultGather(world, rank, n_ch, max_blocks=2, device=torch.device("cpu"), every=every)
    bank = FakeBank(rank, 2*n_ch)
    ok = True
    for s in range(steps):
        g.submit(bank)
        if rank == 0 and s >= 2:
            # by now the gather of step s-2 has completed
            pass
    g.drain()
    if rank == 0:
        got = g.latest().numpy()                    # [world, every, words]: the last (possibly partial) interval
        last = steps - 1
        filled = (last % every) + 1
        for r in range(world):
            for k in range(filled):
                step = last - (filled - 1) + k
                want = (np.arange(2*n_ch, dtype=np.int64)*3 + r*1000003 + step*17).astype(np.int32)
                ok = ok and np.array_equal(got[r, k], want)
        out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("every,steps", [(1, 5), (5, 13), (4, 8)])
def test_result_gather_gloo_world2(every, steps):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 257, steps, out, every)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=10) is True


class AimedBank:
    """Stands in for a ToneBank on the zero-copy path: set_records_buffer() is what ResultGather.aim() calls, and
    launch() does what a kernel launch does with it -- writes this step's record words straight to that address."""

    def __init__(self, rank, n):
        self.rank = rank
        self.n = n
        self.step = 0
        self.dst = None
        self.dst_bytes = 0
        self.copies = 0

    def set_records_buffer(self, ptr, nbytes):
        self.dst = ptr
        self.dst_bytes = nbytes

    def copy_records(self, dst_ptr, dst_bytes):
        self.copies += 1
        raise AssertionError("the aimed path must not copy records")

    def launch(self):
        words = (np.arange(self.n, dtype=np.int64)*5 + self.rank*7777 + self.step*31).astype(np.int32)
        assert self.dst is not None and self.dst_bytes >= words.nbytes
        ctypes.memmove(self.dst, words.ctypes.data, words.nbytes)
        self.step += 1


def _aimed_worker(rank, world, port, total_ch, steps, every, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spandsp_amd.parallel import ResultGather, shard_range
    lo, hi = shard_range(total_ch, world, rank)
    n_ch = hi - lo
    assert n_ch == total_ch//world                    # equal shards (what bench.py runs)
    g = ResultGather(world, rank, n_ch, max_blocks=2, device=torch.device("cpu"), every=every)
    bank = AimedBank(rank, 2*n_ch)
    for s in range(steps):
        g.aim(bank)
        bank.launch()
        g.submit(bank)
    g.drain()
    ok = True
    if rank == 0:
        got = g.latest().numpy()
        last = steps - 1
        filled = (last % every) + 1
        for r in range(world):
            for k in range(filled):
                step = last - (filled - 1) + k
                want = (np.arange(2*n_ch, dtype=np.int64)*5 + r*7777 + step*31).astype(np.int32)
                ok = ok and np.array_equal(got[r, k], want)
        out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("every,steps", [(1, 4), (5, 12)])
def test_result_gather_aimed_path_gloo_world2(every, steps):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_aimed_worker, args=(r, 2, port, 2*193, steps, every, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=10) is True


def _float_worker(rank, world, port, n_ch, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spandsp_amd.parallel import FloatGather
    g = FloatGather(world, rank, n_ch, torch.device("cpu"))
    ok = True
    for round_ in range(3):
        # what spangpu_echo_erle(bank, send, SPANGPU_MEM_DEVICE) does: writes one float per channel into the send buffer
        vals = (np.arange(n_ch, dtype=np.float32)*0.25 + rank*100.0 + round_).astype(np.float32)
        ctypes.memmove(g.send.data_ptr(), vals.ctypes.data, vals.nbytes)
        g.gather()
        res = g.result()
        if rank == 0:
            for r in range(world):
                want = (np.arange(n_ch, dtype=np.float32)*0.25 + r*100.0 + round_).astype(np.float32)
                ok = ok and np.array_equal(res[r].numpy(), want)
        else:
            ok = ok and res is None
    if rank == 0:
        out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_float_gather_gloo_world2():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_float_worker, args=(r, 2, port, 1031, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=10) is True


class FakeDigitBank:
    """Stands in for engine.ToneBank with a digits buffer set: a launch writes one byte per block and channel to where it
    was aimed."""

    def __init__(self, rank, n_ch):
        self.rank = rank
        self.n_ch = n_ch
        self.step = 0

    @staticmethod
    def expected(rank, step, n_ch):
        rng = np.random.default_rng(1000*rank + step)
        d = np.where(rng.random((2, n_ch)) < 0.1, rng.integers(48, 58, (2, n_ch)), 0).astype(np.uint8)
        return d

    def set_digits_ring(self, dst_ptr, slice_bytes, n_slices):
        self.ring = (dst_ptr, slice_bytes, n_slices)
        self.next = 0

    def launch(self):
        dst_ptr, slice_bytes, n_slices = self.ring
        d = self.expected(self.rank, self.step, self.n_ch)
        assert d.nbytes <= slice_bytes
        ctypes.memmove(dst_ptr + self.next*slice_bytes, d.ctypes.data, d.nbytes)
        self.next = (self.next + 1) % n_slices
        self.step += 1


def _digit_worker(rank, world, port, steps, every, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spandsp_amd.parallel import DigitGather
    n_ch = 1031                                 # 2 x 1031 bytes: not a whole number of words
    g = DigitGather(world, rank, n_ch, 2, torch.device("cpu"), every=every)
    bank = FakeDigitBank(rank, n_ch)
    for s in range(steps):
        g.aim(bank)
        bank.launch()
        g.submit(bank)
    g.drain()
    ok = True
    if rank == 0:
        dg = g.digits().numpy()
        ok = dg.shape == (world, every, 2, n_ch)
        last_interval = ((steps - 1)//every)*every
        for r in range(world):
            for k in range(every):
                step = last_interval + k
                if step < steps:
                    ok = ok and np.array_equal(dg[r, k], FakeDigitBank.expected(r, step, n_ch))
        out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("every,steps", [(1, 4), (5, 10), (5, 13)])
def test_digit_gather_gloo_world2(every, steps):
    """bench.py --gpus N (--gather digits): each rank's digit bytes of every step, gathered to rank 0 once per interval."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_digit_worker, args=(r, 2, port, steps, every, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=10) is True


class FakeModemBank:
    """Stands in for engine.ModemBank: copy_events() lays a step's counts and event bytes where it is told."""

    def __init__(self, rank, n_ch):
        self.rank = rank
        self.n_ch = n_ch
        self.step = 0

    @staticmethod
    def expected(rank, step, n_ch, per):
        rng = np.random.default_rng(7000*rank + step)
        counts = rng.integers(0, per + 1, n_ch).astype(np.int32)
        ev = rng.integers(-5, 2, (n_ch, per)).astype(np.int8)
        return counts, ev

    def copy_events(self, dst_ptr, nbytes, per):
        counts, ev = self.expected(self.rank, self.step, self.n_ch, per)
        assert nbytes == self.n_ch*(4 + per)
        ctypes.memmove(dst_ptr, counts.ctypes.data, counts.nbytes)
        ctypes.memmove(dst_ptr + counts.nbytes, ev.ctypes.data, ev.nbytes)
        self.step += 1

    def sync(self):
        pass


def _bits_worker(rank, world, port, steps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spandsp_amd.parallel import BitsGather
    n_ch, per = 1031, 25                        # 1031 x 29 bytes: not a whole number of words
    g = BitsGather(world, rank, n_ch, per, torch.device("cpu"))
    bank = FakeModemBank(rank, n_ch)
    fine = True
    for step in range(steps):
        g.submit(bank)
        got = g.events()                        # this step's, on rank 0
        if rank == 0:
            counts, ev = got
            for r in range(world):
                wc, we = FakeModemBank.expected(r, step, n_ch, per)
                fine = fine and np.array_equal(counts[r].numpy(), wc) and np.array_equal(ev[r].numpy(), we)
    g.drain()
    if rank == 0:
        out.put(bool(fine))
    dist.barrier()
    dist.destroy_process_group()


def test_bits_gather_gloo_world2():
    """the modem banks' bit stream words of every step, gathered to rank 0."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_bits_worker, args=(r, 2, port, 7, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=10) is True
