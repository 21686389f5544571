"""Multi-GPU plumbing for the channel banks.

Channels are fully independent (no cross-channel state anywhere on the path), so
they shard across GPUs as contiguous ranges with NO data-path collective; the only
exchange is a gather of the small per-channel result records to rank 0 after a
step (RCCL over xGMI when the process group backend is "nccl"; gloo in the CPU
tests).  One process per GPU, torch.distributed only.
"""
import torch
import torch.distributed as dist


def require_current_stream(bank, tensor):
    """The ordering contract of every gather below, checked: a collective started with async_op=True is ordered behind what the
    CURRENT torch stream of the tensor's device holds at that moment -- not behind a bank that launches on some other stream.
    So the bank whose results are about to travel must launch on the current stream (bank.set_stream(current_stream), as
    bench.py does), or the caller must have synchronised the bank itself (BitsGather does: bank.sync()).  Raises if neither
    the bank's stream is the current one nor `tensor` is host memory (gloo: the calls are synchronous there)."""
    if not tensor.is_cuda:
        return
    get = getattr(bank, "get_stream", None)
    if get is None:
        return
    cur = torch.cuda.current_stream(tensor.device).cuda_stream
    mine = get()
    mine = getattr(mine, "value", mine) or 0
    if int(mine) != int(cur):
        raise RuntimeError("gather started while the bank launches on stream 0x%x and the current torch stream is 0x%x: the collective would "
                           "not wait for the bank's kernel (bank.set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), or "
                           "run the gather inside `with torch.cuda.stream(...)` of the bank's stream)" % (int(mine), int(cur)))


def shard_range(total_channels, world, rank):
    """Contiguous [lo, hi) channel range of `rank`; sizes differ by at most one."""
    base, extra = divmod(total_channels, world)
    lo = rank*base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


class ResultGather:
    """Double-buffered asynchronous gather of a bank's block-record words to rank 0.

    Zero-copy use: call `aim(bank)` BEFORE each step's launch -- the kernel then writes its records straight into the
    next slice of the send buffer (spangpu_bank_set_records_buffer) -- and `submit(bank)` after it.  (Without `aim`,
    submit() copies the bank's records device-to-device.)  Every `every` steps the filled buffer is gathered to rank 0
    (one collective per reporting interval: `every` = 5 is a 100 ms report at 20 ms ticks).  The previous use of a buffer is waited for before it is refilled,
    so a gather overlaps the kernels of the following interval.  On rank 0, `latest()` returns the
    [world, every, max_blocks*n_ch] int32 words of the most recently completed gather.
    """

    def __init__(self, world, rank, n_ch, max_blocks, device, every=1):
        self.world = world
        self.rank = rank
        self.n = max_blocks*n_ch
        self.every = max(1, int(every))
        size = self.n*self.every
        self.send = [torch.zeros(size, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = None
        if rank == 0:
            self.recv = [[torch.zeros(size, dtype=torch.int32, device=device) for _ in range(world)]
                         for _ in range(2)]
        self.handles = [None, None]
        self.count = 0
        self.done_slot = None

    def _start(self, slot):
        self.handles[slot] = dist.gather(self.send[slot], gather_list=self.recv[slot] if self.rank == 0 else None,
                                         dst=0, async_op=True)

    def _claim(self):
        slot = (self.count//self.every) & 1
        sub = self.count % self.every
        if sub == 0 and self.handles[slot] is not None:
            self.handles[slot].wait()
            self.handles[slot] = None
            self.done_slot = slot
        return slot, sub

    def aim(self, bank):
        """Point the bank's next launch at this step's slice of the send buffer."""
        slot, sub = self._claim()
        bank.set_records_buffer(self.send[slot].data_ptr() + sub*self.n*4, self.n*4)
        self.aimed = True

    def submit(self, bank):
        slot, sub = self._claim()
        if not getattr(self, "aimed", False):
            bank.copy_records(self.send[slot].data_ptr() + sub*self.n*4, self.n*4)
        self.aimed = False
        self.count += 1
        if sub == self.every - 1:
            if hasattr(bank, "join"):
                bank.join()             # a bank in queue mode: its second stream's launches are behind the collective too
            require_current_stream(bank, self.send[slot])
            self._start(slot)

    def drain(self):
        """Send a partly filled interval, wait for everything in flight."""
        sub = self.count % self.every
        if sub != 0:
            slot = (self.count//self.every) & 1
            self._start(slot)
            self.count += self.every - sub
        first = (self.count//self.every) & 1
        for k in range(2):
            slot = (first + k) & 1
            if self.handles[slot] is not None:
                self.handles[slot].wait()
                self.handles[slot] = None
                self.done_slot = slot

    def latest(self):
        if self.rank != 0 or self.done_slot is None:
            return None
        return torch.stack(self.recv[self.done_slot]).view(self.world, self.every, self.n)


class DigitGather(ResultGather):
    """As ResultGather, but what travels is one BYTE per block and channel -- the digit the block delivered, 0 for none
    (spangpu_bank_set_digits_buffer: the detector kernel writes the bytes itself, beside its records) -- instead of the
    32-bit record words: a quarter of the volume.  With eight ranks reporting to one, a 100 ms report of 65536 channels
    per rank is then 0.66 MB per link instead of 2.6 MB.  `aim(bank)` before a step's launch (it acts once per interval: the launches then fill successive
    slices of the send buffer themselves), `submit(bank)` after it closes the step.  On rank 0, `digits()` returns the most
    recently completed gather as uint8 [world, every, max_blocks, n_ch]."""

    def __init__(self, world, rank, n_ch, max_blocks, device, every=1):
        self.n_ch = n_ch
        self.max_blocks = max_blocks
        self.nbytes = max_blocks*n_ch
        words = (self.nbytes + 3)//4
        super().__init__(world, rank, words, 1, device, every=every)

    def aim(self, bank):
        slot, sub = self._claim()
        if sub == 0:
            # one call per reporting interval: the launches of the interval fill successive slices themselves
            bank.set_digits_ring(self.send[slot].data_ptr(), self.n*4, self.every)

    def submit(self, bank):
        slot, sub = self._claim()
        self.count += 1
        if sub == self.every - 1:
            if hasattr(bank, "join"):
                bank.join()             # a bank in queue mode: its second stream's launches are behind the collective too
            require_current_stream(bank, self.send[slot])
            self._start(slot)

    def digits(self):
        got = self.latest()
        if got is None:
            return None
        b = got.contiguous().view(torch.uint8).view(self.world, self.every, self.n*4)[:, :, :self.nbytes]
        return b.reshape(self.world, self.every, self.max_blocks, self.n_ch)


class FloatGather:
    """Gather of one float per channel (e.g. the ERLE of every echo canceller, spangpu_echo_erle()) from every rank
    to rank 0: the fixed-size per-channel result of a reporting interval, the only exchange of the multi-GPU echo
    configuration (BASELINE configs[4]).  `send` is the device buffer the producer writes (for the echo bank:
    bank.erle_device(g.send.data_ptr())); gather() starts the collective on the current stream and returns the
    handle, result() waits for it and returns [world, n_ch] on rank 0."""

    def __init__(self, world, rank, n_ch, device):
        self.world = world
        self.rank = rank
        self.n = n_ch
        self.send = torch.zeros(n_ch, dtype=torch.float32, device=device)
        self.recv = [torch.zeros(n_ch, dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None
        self.handle = None

    def gather(self, bank=None):
        """bank: the producer of `send` (its stream must be the current one: require_current_stream())"""
        if bank is not None:
            require_current_stream(bank, self.send)
        self.handle = dist.gather(self.send, gather_list=self.recv, dst=0, async_op=True)
        return self.handle

    def result(self):
        if self.handle is not None:
            self.handle.wait()
            self.handle = None
        if self.rank != 0:
            return None
        return torch.stack(self.recv)


class BitsGather:
    """Gather of the modem receivers' event streams of a step (the put_bit bits and status reports of every channel:
    SURVEY 8(e), 24 bytes per channel and 160-sample frame for V.29 9600) from every rank to rank 0.  What travels per
    rank and step: int32 counts[n_ch] and int8 events[n_ch][per_channel], written device to device by
    bank.copy_events() (spangpu_modem_copy_events) behind the receiver kernel, so the collective reads what the kernel
    wrote without a host round trip.  Two buffers: submit(bank) after a step's launch starts that step's gather;
    events() waits for the gather started last and, on rank 0, returns it as (counts [world, n_ch] int32,
    events [world, n_ch, per_channel] int8) -- called after the NEXT step's launch it overlaps the transfer with that
    kernel.  (The views are good until the submit after next.)"""

    def __init__(self, world, rank, n_ch, per_channel, device):
        self.world = world
        self.rank = rank
        self.n_ch = n_ch
        self.per = per_channel
        self.nbytes = n_ch*(4 + per_channel)
        words = (self.nbytes + 3)//4
        self.send = [torch.zeros(words, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = None
        if rank == 0:
            self.recv = [[torch.zeros(words, dtype=torch.int32, device=device) for _ in range(world)] for _ in range(2)]
        self.handles = [None, None]
        self.count = 0

    def _wait(self, slot):
        if self.handles[slot] is not None:
            self.handles[slot].wait()
            self.handles[slot] = None

    def submit(self, bank):
        slot = self.count & 1
        self._wait(slot)
        bank.copy_events(self.send[slot].data_ptr(), self.nbytes, self.per)
        if self.send[slot].is_cuda:
            bank.sync()                 # the copy runs on the bank's stream, the collective on the process group's
        self.handles[slot] = dist.gather(self.send[slot], gather_list=self.recv[slot] if self.rank == 0 else None, dst=0,
                                         async_op=True)
        self.count += 1

    def drain(self):
        self._wait(0)
        self._wait(1)

    def events(self):
        if self.count == 0:
            return None
        slot = (self.count - 1) & 1
        self._wait(slot)
        if self.rank != 0:
            return None
        raw = torch.stack(self.recv[slot]).contiguous().view(torch.uint8).view(self.world, -1)[:, :self.nbytes]
        counts = raw[:, :4*self.n_ch].contiguous().view(torch.int32).view(self.world, self.n_ch)
        ev = raw[:, 4*self.n_ch:].contiguous().view(torch.int8).view(self.world, self.n_ch, self.per)
        return counts, ev
