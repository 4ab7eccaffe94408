"""world_size-2 and -3 runs of the sharding logic on CPU (gloo): the cross-rank
bit-offset scan and the byte-wise shard merge, checked against the oracle's
stream for the whole input."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from huffman_amd import synth
from huffman_amd.dist import (allreduce_histogram, gather_histograms, plan_from_histograms,
                              plan_shards)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def shifted_stream(orc, shard, codes, lens, lead):
    """What hf_encode(lead_bits=lead) produces, from the oracle's stream."""
    if shard.size == 0:
        return np.zeros((lead + 7) // 8, dtype=np.uint8), 0
    stream, _ = orc.encode_slice(shard, codes, lens)
    nbits = int(np.asarray(lens, dtype=np.uint64)[shard].sum())
    bits = np.unpackbits(stream)[:nbits]
    return np.packbits(np.concatenate([np.zeros(lead, dtype=np.uint8), bits])), nbits


def worker(rank, world, port, cuts, q):
    import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = synth.generate_host("zipf", cuts[-1], seed=77)
        shard = data[cuts[rank]:cuts[rank + 1]]
        hist = torch.from_numpy(orc.histogram(shard).astype(np.int64))
        allreduce_histogram(hist, dist)
        assert np.array_equal(hist.numpy().astype(np.uint64), orc.histogram(data))
        codes, lens = orc.generate_extended_codewords(hist.numpy().astype(np.uint64))
        my_bits = int(lens[shard].sum())
        plan = plan_shards(my_bits, dist, torch.device("cpu"))
        # the single-collective form used by bench.py gives the same plan
        hists = gather_histograms(torch.from_numpy(orc.histogram(shard).astype(np.int64)), dist)
        assert np.array_equal(hists.sum(axis=0).astype(np.uint64), orc.histogram(data))
        plan2 = plan_from_histograms(hists, lens, rank)
        assert plan2.bits == plan.bits and plan2.start_bit == plan.start_bit
        assert plan2.lead_bits == plan.lead_bits
        local, nbits = shifted_stream(orc, shard, codes, lens, plan.lead_bits)
        assert nbits == my_bits
        f, l = plan.byte_range(rank)
        assert l - f == local.size
        whole = plan.gather(torch.from_numpy(local), dist, dst=0)
        if rank == 0:
            want, _ = orc.encode_slice(data, codes, lens)
            assert plan.total_bits == int(lens[data].sum())
            assert np.array_equal(whole.numpy(), want)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover - surfaced through the queue
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cuts", [[0, 5000, 12345], [0, 4099, 4099, 9001], [0, 1, 2, 70000]])
def test_sharded_streams_concatenate_to_the_reference_stream(cuts):
    import oracle

    oracle.build()
    world = len(cuts) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, cuts, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


@pytest.mark.parametrize("cuts", [[0, 5000, 12345], [0, 4099, 4099, 9001], [0, 1, 2, 70000],
                                  [0, 3, 3, 3, 3000, 3001, 40000]])
def test_shard_plan_and_placement_through_the_c_abi(cuts):
    """hf_shard_plan / hf_place_shard (include/hfgpu.h): what a binding calls on
    every rank between the histogram exchange and its encode, and to assemble
    the stream -- against plain arithmetic and the oracle's stream of the whole
    input, shards placed in reverse order."""
    import ctypes as C

    import oracle as orc
    from huffman_amd._lib import HF_ERR_CAPACITY, HF_OK, lib, u8p, u64p

    orc.build()
    data = synth.generate_host("zipf", cuts[-1], seed=77)
    world = len(cuts) - 1
    shards = [data[cuts[r]:cuts[r + 1]] for r in range(world)]
    hists = np.stack([orc.histogram(s) if s.size else np.zeros(256, dtype=np.uint64)
                      for s in shards]).astype(np.uint64)
    codes, lens = orc.generate_extended_codewords(hists.sum(axis=0))
    lens64 = np.ascontiguousarray(lens, dtype=np.uint64)
    bits = np.zeros(world, dtype=np.uint64)
    start = np.zeros(world, dtype=np.uint64)
    lead = np.zeros(world, dtype=np.uint32)
    first = np.zeros(world, dtype=np.uint64)
    last = np.zeros(world, dtype=np.uint64)
    total = C.c_uint64()
    rc = lib().hf_shard_plan(np.ascontiguousarray(hists).ctypes.data_as(u64p),
                             lens64.ctypes.data_as(u64p), world, bits.ctypes.data_as(u64p),
                             start.ctypes.data_as(u64p), lead.ctypes.data_as(C.POINTER(C.c_uint32)),
                             first.ctypes.data_as(u64p), last.ctypes.data_as(u64p), C.byref(total))
    assert rc == HF_OK
    want_bits = [int(lens64[s].sum()) for s in shards]
    assert bits.tolist() == want_bits
    assert start.tolist() == [sum(want_bits[:r]) for r in range(world)]
    assert lead.tolist() == [s % 8 for s in start.tolist()]
    assert first.tolist() == [s // 8 for s in start.tolist()]
    assert last.tolist() == [(s + b + 7) // 8 for s, b in zip(start.tolist(), want_bits)]
    assert total.value == sum(want_bits)
    # the Python plan (plain host arithmetic) agrees with the C ABI's
    plan = plan_from_histograms(hists, lens, 0)
    assert plan.bits == want_bits and plan.start_bit == start.tolist()
    want, _ = orc.encode_slice(data, codes, lens)
    whole = np.zeros(want.size, dtype=np.uint8)
    for r in reversed(range(world)):
        local, nbits = shifted_stream(orc, shards[r], codes, lens, int(lead[r]))
        assert nbits == want_bits[r]
        local = np.ascontiguousarray(local)
        rc = lib().hf_place_shard(whole.ctypes.data_as(u8p), whole.size,
                                  local.ctypes.data_as(u8p), int(start[r]), nbits)
        assert rc == HF_OK
    assert np.array_equal(whole, want)
    if want.size > 1:
        rc = lib().hf_place_shard(whole.ctypes.data_as(u8p), want.size - 1,
                                  np.ascontiguousarray(want).ctypes.data_as(u8p), 0, total.value)
        assert rc == HF_ERR_CAPACITY
