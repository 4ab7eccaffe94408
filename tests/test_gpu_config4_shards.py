"""The 16 GiB Zipf stream at the shard sizes the multi-GPU runs use (8 x 2 GiB and
2 x 8 GiB), one shard after the other on one GPU: hf_shard_plan from the shards'
histograms, every shard encoded by the planned encoder with its lead bits and
compared -- boundary bytes OR-merged -- with the encode of the whole stream;
shards decoded from their own buffers and, with bit_base, from their byte range
of the whole stream and the whole stream's gap array (start bits above 2^36,
2^31-symbol shards).  What the ranks of `bench.py --gpus N` do, without the
collectives (reference: the one stream of src/huffman/encode.rs:91-136)."""
import numpy as np
import pytest
import torch

from huffman_amd import synth
from huffman_amd.dist import plan_from_histograms

pytestmark = pytest.mark.gpu

GIB = 1 << 30
CS = 64
SEG_SYMS = 256 * CS


@pytest.fixture(scope="module")
def dev():
    from huffman_amd import device
    return device


@pytest.fixture(scope="module")
def stream(dev):
    """The whole 16 GiB stream: data, model, its encode (planned encoder)."""
    n = 16 * GIB
    data = synth.generate_device("zipf", n, seed=2024)
    hist = dev.histogram(data)
    codes, lens = dev.build_codebook(hist)
    model = dev.DeviceModel(codes, lens)
    bh = dev.new_batch_hist(n, data.device)
    dev.histogram_async(data, batch_hist=bh)
    bufs = dev.EncodeBuffers(model, n, CS, out_cap=(n + 8 + 15) & ~15)
    dev.encode_async(data, model, bufs, plan=bh)
    torch.cuda.synchronize()
    total_bits = int((hist * lens).sum())
    assert bufs.result.tolist()[:2] == [total_bits, 0]
    del bh
    yield {"n": n, "data": data, "lens": lens, "model": model, "whole": bufs, "total_bits": total_bits}
    del data, bufs
    torch.cuda.empty_cache()


@pytest.mark.parametrize("world", [8, 2])
def test_shards_of_the_16_gib_stream(dev, stream, world):
    n, data, lens, model, whole = (stream[k] for k in ("n", "data", "lens", "model", "whole"))
    total_bits = stream["total_bits"]
    C = (total_bits + 7) // 8
    per = n // world
    assert per % SEG_SYMS == 0 and (world != 8 or per == 1 << 31)
    hists = np.stack([dev.histogram(data[r * per:(r + 1) * per]) for r in range(world)]).astype(np.uint64)
    merged = torch.zeros(C, dtype=torch.uint8, device=data.device)
    so_whole = whole.seg_off
    decoded_with_bit_base = 0
    for r in range(world):
        plan = plan_from_histograms(hists, lens, r)
        assert plan.total_bits == total_bits and plan.total_bytes == C
        assert plan.bits[r] == int((hists[r] * lens).sum())
        # the scan of the bit lengths is the whole stream's own segment offset there
        assert plan.start_bit[r] == int(so_whole[r * per // SEG_SYMS].item())
        shard = data[r * per:(r + 1) * per]
        bh = dev.new_batch_hist(per, data.device)
        dev.histogram_async(shard, batch_hist=bh)
        bufs = dev.EncodeBuffers(model, per, CS, lead_bits=plan.lead_bits, out_cap=(per + 24) & ~15)
        dev.encode_async(shard, model, bufs, lead_bits=plan.lead_bits, plan=bh)
        torch.cuda.synchronize()
        assert bufs.result.tolist()[:2] == [plan.bits[r], 0], f"rank {r}"
        first, last = plan.byte_range(r)
        assert first == plan.start_bit[r] // 8 and last == (plan.start_bit[r] + plan.bits[r] + 7) // 8
        mine = bufs.data[: last - first]
        # interior bytes are the whole stream's; the two boundary bytes hold only this
        # shard's bits (the neighbours' are zero), so OR-ing the shards gives the stream
        assert torch.equal(mine[1:-1], whole.data[first + 1:last - 1]), f"rank {r}"
        lead = plan.lead_bits
        assert int(mine[0].item()) == int(whole.data[first].item()) & (0xff >> lead)
        tail_bits = (plan.start_bit[r] + plan.bits[r]) % 8
        if tail_bits:
            assert int(mine[-1].item()) == int(whole.data[last - 1].item()) & (0xff00 >> tail_bits) & 0xff
        merged[first:last] |= mine
        # the shard's own gap array starts at its lead bits and sums to its length
        assert int(bufs.seg_off[0].item()) == lead
        if r in (0, world - 2, world - 1):
            # decode from the shard's own buffer and gap array, as its rank does
            out = torch.empty(per, dtype=torch.uint8, device=data.device)
            res = torch.zeros(4, dtype=torch.int64, device=data.device)
            dev.decode_async(bufs.data, last - first, model, per, CS, bufs.chunk_bits, bufs.seg_off, out, res)
            torch.cuda.synchronize()
            assert res.tolist()[1] == 0 and torch.equal(out, shard), f"rank {r}"
            # ... and from its byte range of the whole stream with the whole stream's
            # gap array: absolute segment offsets, bit_base = where the copy starts
            s0 = r * per // SEG_SYMS
            start = ((plan.start_bit[r] // 8) & ~3)
            copy = whole.data[start:last].clone()
            out.zero_()
            res.zero_()
            dev.decode_async(copy, last - start, model, per, CS, whole.chunk_bits[s0 * 256:],
                             whole.seg_off[s0:], out, res, bit_base=8 * start)
            torch.cuda.synchronize()
            assert res.tolist()[1] == 0 and torch.equal(out, shard), f"rank {r} (bit_base)"
            if 8 * start > 1 << 36:
                decoded_with_bit_base += 1
            del out, copy
        del bufs, bh
    assert torch.equal(merged, whole.data[:C])
    if world == 8:
        assert decoded_with_bit_base >= 2  # ranks 6 and 7 start above bit 2^36
