"""Multi-GPU plumbing for the channel banks.

Channels are fully independent (no cross-channel state anywhere on the path), so
they shard across GPUs as contiguous ranges with NO data-path collective; the only
exchange is a gather of the small per-channel result records to rank 0 after a
step (RCCL over xGMI when the process group backend is "nccl"; gloo in the CPU
tests).  One process per GPU, torch.distributed only.
"""
import torch
import torch.distributed as dist


def shard_range(total_channels, world, rank):
    """Contiguous [lo, hi) channel range of `rank`; sizes differ by at most one."""
    base, extra = divmod(total_channels, world)
    lo = rank*base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


class ResultGather:
    """Double-buffered asynchronous gather of a bank's block-record words to rank 0.

    submit(bank) queues a device-to-device copy of the bank's records into a send
    buffer (spangpu_bank_copy_records) and starts a gather of it; the previous use of
    that buffer is waited for first, so a step's gather overlaps the next step's kernel.
    On rank 0, `latest()` returns the [world, max_blocks*n_ch] int32 words of the
    most recently completed gather.
    """

    def __init__(self, world, rank, n_ch, max_blocks, device):
        self.world = world
        self.rank = rank
        self.n = max_blocks*n_ch
        self.send = [torch.zeros(self.n, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = None
        if rank == 0:
            self.recv = [[torch.zeros(self.n, dtype=torch.int32, device=device) for _ in range(world)]
                         for _ in range(2)]
        self.handles = [None, None]
        self.count = 0
        self.done_slot = None

    def submit(self, bank):
        slot = self.count & 1
        if self.handles[slot] is not None:
            self.handles[slot].wait()
            self.done_slot = slot
        buf = self.send[slot]
        bank.copy_records(buf.data_ptr(), buf.numel()*4)
        self.handles[slot] = dist.gather(buf, gather_list=self.recv[slot] if self.rank == 0 else None,
                                         dst=0, async_op=True)
        self.count += 1

    def drain(self):
        for k in range(2):
            slot = (self.count + k) & 1
            if self.handles[slot] is not None:
                self.handles[slot].wait()
                self.handles[slot] = None
                self.done_slot = slot

    def latest(self):
        if self.rank != 0 or self.done_slot is None:
            return None
        return torch.stack(self.recv[self.done_slot])
