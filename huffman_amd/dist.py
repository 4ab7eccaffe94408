"""Multi-GPU sharding of one logical stream: one process per GPU,
torch.distributed (RCCL over xGMI on the GPU box, gloo in CPU tests).

The input is split into contiguous shards, one per rank.  Two tiny exchanges
place the shards in the global stream; an optional bulk one moves them:
  1. all-reduce (sum) of the 256-bin histograms -> every rank builds the same
     codebook on its host (2 KiB: latency bound, topology irrelevant);
  2. all-gather of one 64-bit bit length per rank + local exclusive prefix
     = the cross-GPU scan: shard r starts at bit start_bit[r] of the stream and
     is encoded with lead_bits = start_bit[r] % 8 leading zero bits, so shards
     concatenate bytewise and only the boundary byte needs an OR;
     (gather_histograms + plan_from_histograms fold 1 and 2 into a single
     all-gather of the histograms: one collective and one host sync per stream);
  3. (optional) the shards are sent to rank 0 and merged into one stream.
     Root-inbound xGMI is 7 links x ~153 GB/s, below one GPU's encode rate, so
     keeping the stream sharded (this plan is its descriptor) is the default.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class ShardPlan:
    rank: int
    world: int
    bits: list[int]        # bit length of every shard
    start_bit: list[int]   # first bit of every shard in the global stream

    @property
    def lead_bits(self) -> int:
        return self.start_bit[self.rank] % 8

    @property
    def total_bits(self) -> int:
        return self.start_bit[-1] + self.bits[-1]

    @property
    def total_bytes(self) -> int:
        return (self.total_bits + 7) // 8

    def byte_range(self, r: int) -> tuple[int, int]:
        """[first, last) byte of shard r's local stream inside the global stream."""
        first = self.start_bit[r] // 8
        last = (self.start_bit[r] + self.bits[r] + 7) // 8
        return first, last

    def gather(self, local_stream: torch.Tensor, dist, dst: int = 0):
        """Physically concatenate the shards on rank `dst`.  local_stream is
        this rank's encoded shard (lead bits included).  Returns the whole
        stream on dst, None elsewhere."""
        dev = local_stream.device
        first, last = self.byte_range(self.rank)
        assert local_stream.numel() == last - first
        # boundary bytes first: every rank's leading byte, OR-merged on dst
        head = torch.zeros(1, dtype=torch.uint8, device=dev)
        if local_stream.numel():
            head[0] = local_stream[0]
        heads = [torch.zeros(1, dtype=torch.uint8, device=dev) for _ in range(self.world)]
        dist.all_gather(heads, head)
        # bulk: everything after a shared leading byte goes to a disjoint range
        skip, shared = [], []
        for r in range(self.world):
            f, l = self.byte_range(r)
            shared.append(bool(self.start_bit[r] % 8) and self.bits[r] > 0)
            skip.append(l - f if self.bits[r] == 0 else int(shared[r]))
        if self.rank == dst:
            whole = torch.zeros(self.total_bytes, dtype=torch.uint8, device=dev)
            reqs = []
            for r in range(self.world):
                f, l = self.byte_range(r)
                if l - f - skip[r] <= 0:
                    continue
                view = whole[f + skip[r]: l]
                if r == dst:
                    view.copy_(local_stream[skip[r]:])
                else:
                    reqs.append(dist.irecv(view, src=r))
            for q in reqs:
                q.wait()
            for r in range(self.world):
                f, l = self.byte_range(r)
                if shared[r]:
                    whole[f] |= heads[r][0]
            return whole
        n_send = local_stream.numel() - skip[self.rank]
        if n_send > 0:
            dist.isend(local_stream[skip[self.rank]:].contiguous(), dst=dst).wait()
        return None


def plan_shards(my_bits: int, dist, device) -> ShardPlan:
    """Cross-rank exclusive scan of the shards' bit lengths."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    mine = torch.tensor([my_bits], dtype=torch.int64, device=device)
    every = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(every, mine)
    bits = [int(t.item()) for t in every]
    start, acc = [], 0
    for b in bits:
        start.append(acc)
        acc += b
    return ShardPlan(rank, world, bits, start)


def gather_histograms(local_hist: torch.Tensor, dist):
    """All ranks' 256-bin histograms on every rank's host, shape [world, 256]:
    one all-gather of 2 KiB per rank and one device-to-host copy carry both
    exchanges of the plan -- the sum over ranks is the global histogram, and
    since every rank derives the same codebook from it, row r times the code
    lengths is shard r's bit length (plan_from_histograms)."""
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and local_hist.is_cuda:
        local_hist = local_hist.cpu()  # gloo gathers host tensors
    every = torch.empty(world * 256, dtype=local_hist.dtype, device=local_hist.device)
    dist.all_gather_into_tensor(every, local_hist.contiguous())
    return every.view(world, 256).cpu().numpy()


def plan_from_histograms(hists, lens, rank: int) -> ShardPlan:
    """The cross-rank exclusive scan of the shards' bit lengths, computed
    identically on every rank from the gathered histograms: row r times the
    code lengths is shard r's bit length.  Plain host arithmetic (the C ABI's
    hf_shard_plan does the same sums for bindings in other languages and the
    tests hold the two against each other), so a rank needs neither the GPU
    library nor a GPU to plan."""
    import numpy as np

    h = np.ascontiguousarray(np.asarray(hists, dtype=np.uint64))
    if h.ndim != 2 or h.shape[1] != 256:
        raise ValueError("expected [world, 256] histograms")
    ln = np.asarray(lens, dtype=np.uint64).reshape(256)
    if int(ln.max(initial=0)) > 64:
        raise ValueError("code length above 64")
    world = h.shape[0]
    if float(h.sum(dtype=np.float64)) * 64.0 < 2.0 ** 62:
        bits = [int(b) for b in (h * ln[None, :]).sum(axis=1, dtype=np.uint64)]
    else:  # (Python integers: shards of 2^58 symbols times 64 bits do not fit 64 bits)
        bits = [sum(int(c) * int(l) for c, l in zip(h[r], ln) if c) for r in range(world)]
    start, acc = [], 0
    for b in bits:
        start.append(acc)
        acc += b
    return ShardPlan(rank, world, bits, start)


def allreduce_histogram(hist: torch.Tensor, dist) -> torch.Tensor:
    """Sum of the 256-bin histograms of all ranks (in place)."""
    dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist
