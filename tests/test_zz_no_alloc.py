"""hf_encode_noalloc_async (the entry point that allocates nothing: its scratch
is a fixed block inside the library) against hf_encode_async /
hf_encode_planned_async on the same inputs, through the C ABI: identical
stream bytes, gap arrays and result blocks, on two calls in a row, the second
with a smaller input (the block keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from huffman_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (there is no CPU fallback)")
    from huffman_amd import device

    return device


def _one(dev, orc, kind, n, seed, planned, lead, cs):
    data = synth.generate_host(kind, n, seed=seed)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    t = torch.from_numpy(data).cuda()
    model = dev.DeviceModel(codes, lens)
    ref = dev.EncodeBuffers(model, n, cs, lead_bits=lead)
    got = dev.EncodeBuffers(model, n, cs, lead_bits=lead)
    ref.data.fill_(0x5A)
    got.data.fill_(0xA5)
    bh = None
    if planned:
        bh = dev.new_batch_hist(n, t.device)
        dev.histogram_async(t, batch_hist=bh)
    dev.encode_async(t, model, ref, lead_bits=lead, plan=bh)
    dev.encode_noalloc_async(t, model, got, lead_bits=lead, plan=bh)
    torch.cuda.synchronize()
    assert got.result.tolist() == ref.result.tolist()
    bits, flags = got.result.tolist()[:2]
    assert flags == 0
    nbytes = (lead + bits + 7) // 8
    assert torch.equal(got.data[:nbytes], ref.data[:nbytes])
    assert torch.equal(got.chunk_bits, ref.chunk_bits)
    assert torch.equal(got.seg_off, ref.seg_off)
    # and both are the reference's stream (lead bits are zeros in front of it)
    if lead == 0:
        want, _ = orc.encode_slice(data, codes, lens)
        assert np.array_equal(got.data[:nbytes].cpu().numpy(), want)


@pytest.mark.parametrize("planned", [False, True])
def test_noalloc_encode_equals_the_allocating_entry_point(dev, orc, planned):
    # two calls in a row on the same GPU: the second, smaller one finds the
    # first one's descriptors / plan arrays in the block
    _one(dev, orc, "zipf", (5 << 20) + 4321, 11, planned, 0, 64)
    _one(dev, orc, "geometric", (1 << 20) + 17, 12, planned, 0, 64)
    _one(dev, orc, "uniform", 70001, 13, planned, 3, 128)
    _one(dev, orc, "testfile", 1, 14, planned, 0, 256)


def test_noalloc_encode_rejects_inputs_beyond_its_block(dev, orc):
    from huffman_amd._lib import HF_ERR_UNSUPPORTED, HF_NOALLOC_MAX_SYMBOLS, HfError, lib

    data = synth.generate_host("zipf", 4096, seed=1)
    codes, lens = orc.generate_extended_codewords(orc.histogram(data))
    t = torch.from_numpy(data).cuda()
    model = dev.DeviceModel(codes, lens)
    bufs = dev.EncodeBuffers(model, 4096)
    ctx = model.ctx
    rc = lib().hf_encode_noalloc_async(ctx.handle, model.handle, t.data_ptr(),
                                       HF_NOALLOC_MAX_SYMBOLS + 1, 0, None, bufs.data.data_ptr(),
                                       bufs.cap, 64, bufs.chunk_bits.data_ptr(),
                                       bufs.seg_off.data_ptr(), bufs.result.data_ptr())
    assert rc == HF_ERR_UNSUPPORTED
    assert b"limit" in lib().hf_last_error(ctx.handle)
    with pytest.raises(HfError):
        ctx.check(rc)


def test_noalloc_encode_on_two_streams_shares_the_block_in_order(dev, orc):
    """Two streams of one GPU, a call on each without a wait in between: both use
    the one fixed block, so the second call has to start behind the first one's
    kernels (the library orders them with a per-GPU event).  Several rounds, the
    longer input first, so that the first call is still running when the second
    is queued."""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cases = []
    for kind, n, seed in (("zipf", (96 << 20) + 12345, 21), ("geometric", (3 << 20) + 77, 22)):
        data = synth.generate_host(kind, n, seed=seed)
        codes, lens = orc.generate_extended_codewords(orc.histogram(data))
        t = torch.from_numpy(data).cuda()
        model = dev.DeviceModel(codes, lens)
        ref = dev.EncodeBuffers(model, n, 64)
        dev.encode_async(t, model, ref)
        cases.append((t, model, ref, dev.EncodeBuffers(model, n, 64)))
    torch.cuda.synchronize()
    for _ in range(4):
        for (t, model, ref, got), stream in zip(cases, (s1, s2)):
            got.data.zero_()
        torch.cuda.synchronize()
        for (t, model, ref, got), stream in zip(cases, (s1, s2)):
            with torch.cuda.stream(stream):
                dev.encode_noalloc_async(t, model, got)
        torch.cuda.synchronize()
        for t, model, ref, got in cases:
            assert got.result.tolist() == ref.result.tolist()
            nbytes = (ref.result.tolist()[0] + 7) // 8
            assert torch.equal(got.data[:nbytes], ref.data[:nbytes])
            assert torch.equal(got.chunk_bits, ref.chunk_bits)
            assert torch.equal(got.seg_off, ref.seg_off)
