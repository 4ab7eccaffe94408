#!/usr/bin/env python3
"""Throughput of the Huffman hot path on MI355X.

One step = one full pass of the path over a synthetic byte stream that is
already resident in HBM:  histogram -> host codebook -> device tables ->
encode (+ gap array) -> decode.  `value` is uncompressed GB/s of that round
trip (bytes of all ranks / max-over-ranks time); encode-only and decode-only
rates and per-kernel durations (HIP events on the launch stream, inside the
timed region) are reported alongside.

  python bench.py [--gpus N] [--steps K] [--warmup W]
N=1: 1 GiB Zipf(s=1.1) stream.  N>1 (launched by torch.distributed.run, one
rank per GPU): every rank owns a contiguous 1 GiB shard of one logical stream
(weak scaling; --gib 2 at N=8 is the 16 GiB configuration); the 256-bin
histograms are all-gathered over RCCL, every rank sums them and builds the same
codebook, and the exclusive scan of the shards' bit lengths (histogram row x
code lengths) places each shard at its own bit phase of the global stream.
The compressed shards stay where they are unless --gather.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from huffman_amd import device as hd  # noqa: E402
from huffman_amd import synth  # noqa: E402
from huffman_amd.dist import gather_histograms, plan_from_histograms  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kind", default="zipf", choices=["zipf", "uniform", "geometric", "testfile"])
    ap.add_argument("--gib", type=float, default=None, help="GiB per GPU (default 1)")
    ap.add_argument("--chunk-syms", type=int, default=hd.DEFAULT_CHUNK_SYMS)
    ap.add_argument("--enc-mode", type=int, default=1)
    ap.add_argument("--gather", action="store_true", help="also gather the shards to rank 0")
    ap.add_argument("--no-plan", action="store_true",
                    help="encode without the batch histograms of the counting pass "
                         "(single-pass chained scan instead)")
    ap.add_argument("--hist-grid", type=int, default=0,
                    help="workgroups of the counting pass (0 = the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--seed", type=int, default=2024)
    return ap.parse_args()


class Stage:
    """Accumulates HIP-event durations of one kernel group."""

    def __init__(self):
        self.pairs = []

    def start(self):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pairs.append((e0, e1))

    def stop(self):
        self.pairs[-1][1].record()

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(1, len(self.pairs))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # HF_BENCH_FORCE_DIST=1 runs the collectives with a single rank too (a
    # rehearsal of the multi-GPU code path on a one-GPU box)
    use_dist = world > 1 or os.environ.get("HF_BENCH_FORCE_DIST") == "1"
    json_fd = None
    if use_dist:
        import torch.distributed as dist  # noqa: F811

        # RCCL prints a version banner on stdout when its first communicator
        # comes up: everything but the result line goes to stderr instead
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=dev)

    gib = args.gib if args.gib is not None else 1.0
    n = int(gib * (1 << 30))
    data = synth.generate_device(args.kind, n, seed=args.seed + rank, device=dev)
    ctx = hd.Context.get(dev)
    ctx.set_option("enc_mode", args.enc_mode)
    ctx.set_option("hist_grid", args.hist_grid)
    cs = args.chunk_syms

    d_hist = torch.empty(256, dtype=torch.int64, device=dev)
    # the counting pass also records a histogram per 64 Ki symbols: with the
    # code lengths they place every batch of the stream before the encoder runs
    batch_hist = None if args.no_plan else hd.new_batch_hist(n, dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    dres = torch.zeros(4, dtype=torch.int64, device=dev)
    st_hist, st_enc, st_dec = Stage(), Stage(), Stage()
    bufs = None
    state = {}

    def step(timed: bool):
        nonlocal bufs
        # pass 1: histogram
        if timed:
            st_hist.start()
        hd.histogram_async(data, d_hist, batch_hist=batch_hist)
        if timed:
            st_hist.stop()
        # host: codebook, then device tables.  With several ranks one all-gather
        # of the 2 KiB histograms merges them and, times the code lengths, gives
        # every shard's bit length: the cross-GPU exclusive scan is then local
        lead = 0
        plan = None
        if use_dist:
            hists = gather_histograms(d_hist, dist)
            hist = hists.sum(axis=0).astype(np.uint64)
        else:
            hist = d_hist.cpu().numpy().astype(np.uint64)
        codes, lens = hd.build_codebook(hist)
        model = hd.DeviceModel(codes, lens, device=dev)
        if use_dist:
            plan = plan_from_histograms(hists, lens, rank)
            my_bits = plan.bits[rank]
            lead = plan.lead_bits
        if bufs is None:
            cap = hd.DeviceModel.encode_bound(model, n, 7)
            bufs = hd.EncodeBuffers(model, n, cs, out_cap=cap)
        # pass 2: encode
        if timed:
            st_enc.start()
        hd.encode_async(data, model, bufs, lead_bits=lead, plan=batch_hist)
        if timed:
            st_enc.stop()
        # the stream length is needed on the host (and is known from the histogram)
        total_bits = my_bits if use_dist else int((hist * lens).sum())
        nbytes = (lead + total_bits + 7) // 8
        if args.gather and use_dist:
            plan.gather(bufs.data[:nbytes], dist)
        # decode from the gap array
        if timed:
            st_dec.start()
        hd.decode_async(bufs.data, nbytes, model, n, cs, bufs.chunk_bits, bufs.seg_off, out, dres)
        if timed:
            st_dec.stop()
        state.update(model=model, nbytes=nbytes, total_bits=total_bits, lens=lens, hist=hist)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ctx.set_option("kernel_timing", 1)  # HIP events around the three dominant kernels
    for _ in range(args.warmup):
        step(False)
    sync_all()
    for kind_id in range(3):
        ctx.kernel_ms(kind_id)  # drop the warm-up launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: flags clear, round trip exact
    flags = int(bufs.result[1].item()) | int(dres[1].item())
    ok = flags == 0 and int(bufs.result[0].item()) == state["total_bits"] and torch.equal(out, data)
    if use_dist:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())

    ms_step = elapsed / args.steps * 1e3
    total_bytes = n * world
    value = total_bytes / (elapsed / args.steps) / 1e9
    C = state["nbytes"]
    # per stage: HIP events around the whole call (helper kernels and launch gaps
    # included); per kernel: the library's events around the dominant kernel alone,
    # which is what a rocprofv3 kernel trace of this command reports as well
    stage_ms = {"histogram": st_hist.mean_ms(), "encode": st_enc.mean_ms(),
                "decode": st_dec.mean_ms()}
    kern_ms = {}
    for kind_id, name in enumerate(("histogram", "encode", "decode")):
        total, launches = ctx.kernel_ms(kind_id)
        kern_ms[name] = total / launches if launches else stage_ms[name]
    ctx.set_option("kernel_timing", 0)
    hist_ms, enc_ms, dec_ms = kern_ms["histogram"], kern_ms["encode"], kern_ms["decode"]
    kernels = {
        "histogram": {"ms": hist_ms, "alg_bytes": n},
        "encode": {"ms": enc_ms, "alg_bytes": n + C},
        "decode": {"ms": dec_ms, "alg_bytes": C + n},
    }
    for k in kernels.values():
        k["GBps"] = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
    dom = max(("encode", "decode"), key=lambda k: kernels[k]["ms"])
    roofline = {
        "bound": "hbm",
        "kernel": {"encode": "enc_pack_kernel",
                   # (larger chunks of a model whose table needs a second level take dec_kernel)
                   "decode": "dec_kernel_reg" if cs == 64 or int(state["lens"].max()) <= 11
                   else "dec_kernel"}[dom],
        "achieved": round(kernels[dom]["GBps"], 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBPS, 4),
        "traffic": pmc_traffic(args.kind, n, cs, dom),
        "alg_bytes": kernels[dom]["alg_bytes"],
    }

    result = {
        "metric": "encode+decode GB/s (uncompressed)",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": f"synthetic {args.kind} bytes, seed {args.seed}+rank, generated in HBM",
        "config": {
            "workload": (f"{gib:g} GiB {args.kind} stream per GPU x {world} GPU: histogram + host "
                         "codebook + encode + gap-array decode"),
            "bytes_per_gpu": n,
            "chunk_syms": cs,
            "enc_mode": "planned" if batch_hist is not None else args.enc_mode,
            "compressed_ratio": round(C / n, 4),
            "max_code_len": int(state["lens"].max()),
            "gather_to_root": bool(args.gather),
        },
        "roundtrip_ok": ok,
        "encode_GBps_uncompressed": round(n / (enc_ms * 1e-3) / 1e9, 1),
        "decode_GBps_uncompressed": round(n / (dec_ms * 1e-3) / 1e9, 1),
        "histogram_GBps": round(kernels["histogram"]["GBps"], 1),
        "kernel_ms": {k: round(v["ms"], 4) for k, v in kernels.items()},
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "hbm_frac": {k: round(v["GBps"] / HBM_PEAK_GBPS, 4) for k, v in kernels.items()},
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(data, args.cpu_sample_mib)
    if rank == 0:
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(result) + "\n").encode())
        else:
            print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


def pmc_traffic(kind: str, n: int, cs: int, kernel: str):
    """HBM bytes per launch of `kernel` from the PMC counters (FETCH_SIZE doubled
    as gfx950 needs, plus WRITE_SIZE), collected in separate rocprofv3 --pmc
    passes on this workload and recorded in huffman_amd/pmc_traffic.json by
    tools/make_traffic.py; None when this workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "huffman_amd", "pmc_traffic.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    entry = table.get(f"{kind}:{n}:{cs}")
    return entry.get(kernel) if entry else None


def cpu_baseline(data: torch.Tensor, sample_mib: int) -> dict:
    """The CPU restatement of the reference (oracle/oracle.c), single thread, on
    the leading sample of the same stream: from_slice + write + flush + read."""
    import oracle

    oracle.build()
    m = min(data.numel(), sample_mib << 20)
    host = data[:m].cpu().numpy()
    t0 = time.perf_counter()
    hist = oracle.histogram(host)
    codes, lens = oracle.generate_extended_codewords(hist)
    t1 = time.perf_counter()
    stream, _ = oracle.encode_slice(host, codes, lens)
    t2 = time.perf_counter()
    back = oracle.decode_read(stream, codes, lens, m)
    t3 = time.perf_counter()
    assert np.array_equal(back, host)
    return {
        "value": round(m / (t3 - t0) / 1e9, 4),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {m >> 20} MiB of the same stream; histogram+codebook {t1 - t0:.2f}s, "
                  f"encode {t2 - t1:.2f}s, decode {t3 - t2:.2f}s",
        "encode_GBps": round(m / (t2 - t0) / 1e9, 4),
        "decode_GBps": round(m / (t3 - t2) / 1e9, 4),
    }


if __name__ == "__main__":
    main()
