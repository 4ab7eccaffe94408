#!/usr/bin/env python3
"""Throughput of the Huffman hot path on MI355X.

One step = one full pass of the path over a synthetic byte stream that is
already resident in HBM:  counting pass (histogram + batch histograms) -> host
codebook -> device tables -> encode (+ gap array) -> decode.  `value` is
uncompressed GB/s of that round trip (bytes of all ranks / max-over-ranks
time); encode-only and decode-only rates and per-kernel durations (HIP events
on the launch stream, inside the timed region) are reported alongside.

  python bench.py [--gpus N] [--steps K] [--warmup W]

Every N measures the same workload: ONE logical 16 GiB Zipf(s=1.1) stream
(--total-gib) cut into N contiguous shards of 16/N GiB, one rank per GPU
(strong scaling; at N = 1 the single shard is the whole stream).  Unless a
single workload is asked for with --kind / --gib, the N = 1 run then measures
the other single-GPU configurations -- 1 GiB Zipf, 1 GiB uniform, 4 GiB
geometric(p=0.5) -- and reports them in the "configs" array of the same JSON
line.  Every record also carries "decode_nogap": the decode of the bare stream
without its gap array (hf_decode_bare, what a file written by the reference
itself needs), timed outside the round trip.  "adaptors" holds the rates of the
streaming Encoder::write / Decoder::read adaptors driven the way the
reference's callers drive them (tools/src/adaptor_rate.cpp, host buffers).

N > 1: the 256-bin histograms
are all-gathered over RCCL, every rank sums them and builds the same codebook,
and the exclusive scan of the shards' bit lengths (histogram row x code
lengths) places each shard at its own bit phase of the global stream.  The
compressed shards stay where they are unless --gather.  The ranks are either
started by `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE
in the environment) or, when bench.py is run plainly with --gpus N, by
bench.py itself: it starts that same launcher as a child process before it
touches a GPU and passes the child's JSON line through.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
GIB = 1 << 30


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps of the headline workload (default 20)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--kind", default=None, choices=["zipf", "uniform", "geometric", "testfile"])
    ap.add_argument("--gib", type=float, default=None,
                    help="GiB per GPU; with --kind or --gib only this one workload is run")
    ap.add_argument("--total-gib", type=float, default=16.0,
                    help="N>1: size of the one logical stream the ranks share")
    ap.add_argument("--chunk-syms", type=int, default=64)
    ap.add_argument("--enc-mode", type=int, default=1)
    ap.add_argument("--gather", action="store_true", help="also gather the shards to rank 0")
    ap.add_argument("--no-plan", action="store_true",
                    help="encode without the batch histograms of the counting pass "
                         "(single-pass chained scan instead)")
    ap.add_argument("--hist-grid", type=int, default=0,
                    help="workgroups of the counting pass (0 = the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N=1: skip the 1 GiB Zipf / uniform / 4 GiB geometric sub-records")
    ap.add_argument("--no-adaptors", action="store_true", help="skip the adaptor rates")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--lib-debug", type=int, default=0,
                    help="the library's \"debug\" option (timing experiments: 64 = the other "
                         "LDS image layout in the decoder, 128 = no two-symbol decode table)")
    ap.add_argument("--seed", type=int, default=2024)
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a
    child job (torch.distributed.run, one process per GPU) and hand its result
    line through.  Nothing in this process has touched a GPU: torch is not even
    imported yet."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


class Stage:
    """Accumulates HIP-event durations of one kernel group."""

    def __init__(self, torch):
        self.torch = torch
        self.pairs = []

    def start(self):
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pairs.append((e0, e1))

    def stop(self):
        self.pairs[-1][1].record()

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(1, len(self.pairs))


def tie_count(hist) -> int:
    """Symbols whose count equals another symbol's: the reference's sort leaves
    their order (and so their codes, not their lengths) to the toolchain
    (src/huffman/mod.rs:138, :222-229); here equal counts keep symbol order."""
    import numpy as np

    nz = np.asarray(hist)[np.asarray(hist) > 0]
    _, counts = np.unique(nz, return_counts=True)
    return int(counts[counts > 1].sum())


def run_workload(args, env, kind: str, n: int, steps: int, warmup: int) -> dict:
    """Times `steps` round trips of one workload (n bytes on this rank) and checks
    what was timed: result flags clear, stream length = sum(hist x len), decoded
    bytes equal the input."""
    import numpy as np
    import torch

    from huffman_amd import device as hd
    from huffman_amd import synth
    from huffman_amd.dist import gather_histograms, plan_from_histograms

    dev, dist, rank, world = env["dev"], env["dist"], env["rank"], env["world"]
    use_dist = dist is not None
    data = synth.generate_device(kind, n, seed=args.seed + rank, device=dev)
    ctx = hd.Context.get(dev)
    ctx.set_option("enc_mode", args.enc_mode)
    ctx.set_option("hist_grid", args.hist_grid)
    ctx.set_option("debug", args.lib_debug)
    cs = args.chunk_syms

    d_hist = torch.empty(256, dtype=torch.int64, device=dev)
    # the counting pass also records a histogram per 64 Ki symbols: with the
    # code lengths they place every batch of the stream before the encoder runs
    batch_hist = None if args.no_plan else hd.new_batch_hist(n, dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    dres = torch.zeros(4, dtype=torch.int64, device=dev)
    st_hist, st_enc, st_dec = Stage(torch), Stage(torch), Stage(torch)
    bufs = None
    state = {}

    def step(timed: bool):
        nonlocal bufs
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
            # a Huffman code never needs more than 8 bits per symbol on average
            cap = (n + 8 + 15) & ~15
            bufs = hd.EncodeBuffers(model, n, cs, out_cap=cap)
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
        if timed:
            st_dec.start()
        hd.decode_async(bufs.data, nbytes, model, n, cs, bufs.chunk_bits, bufs.seg_off, out, dres)
        if timed:
            st_dec.stop()
        state.update(model=model, nbytes=nbytes, total_bits=total_bits, lens=lens, hist=hist,
                     plan=plan)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ctx.set_option("kernel_timing", 1)  # HIP events around the three dominant kernels
    for _ in range(warmup):
        step(False)
    sync_all()
    for kind_id in range(3):
        ctx.kernel_ms(kind_id)  # drop the warm-up launches
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=env["coll_dev"])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: flags clear, round trip exact
    flags = int(bufs.result[1].item()) | int(dres[1].item())
    ok = flags == 0 and int(bufs.result[0].item()) == state["total_bits"] and torch.equal(out, data)
    if use_dist:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=env["coll_dev"])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())

    ms_step = elapsed / steps * 1e3
    C = state["nbytes"]
    nogap = None
    if not use_dist and ok:
        nogap = bare_decode(ctx, hd, torch, bufs, state["model"], C, n, out, data, kind, cs)
        ok = ok and nogap["ok"]
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
    kernels = {
        "histogram": {"ms": kern_ms["histogram"], "alg_bytes": n},
        "encode": {"ms": kern_ms["encode"], "alg_bytes": n + C},
        "decode": {"ms": kern_ms["decode"], "alg_bytes": C + n},
    }
    for k in kernels.values():
        k["GBps"] = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
    # the kernel that runs furthest below the roofline
    dom = min(("encode", "decode"), key=lambda k: kernels[k]["GBps"])
    maxlen = int(state["lens"].max())
    traffic, traffic_source = pmc_traffic(kind, n, cs, dom)
    roofline = {
        "bound": "hbm",
        "kernel": {"encode": ctx.encode_kernel(),
                   "decode": ctx.decode_kernel()}[dom],
        "achieved": round(kernels[dom]["GBps"], 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBPS, 4),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "alg_bytes": kernels[dom]["alg_bytes"],
    }
    rec = {
        "workload": f"{n / GIB:g} GiB {kind} per GPU x {world} GPU",
        "kind": kind,
        "bytes_per_gpu": n,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms_step, 4),
        "value": round(n * world / (elapsed / steps) / 1e9, 2),
        "unit": "GB/s",
        "roundtrip_ok": ok,
        "compressed_ratio": round(C / n, 4),
        "max_code_len": maxlen,
        "ties": tie_count(state["hist"]),
        "encode_GBps_uncompressed": round(n / (kern_ms["encode"] * 1e-3) / 1e9, 1),
        "decode_GBps_uncompressed": round(n / (kern_ms["decode"] * 1e-3) / 1e9, 1),
        "histogram_GBps": round(kernels["histogram"]["GBps"], 1),
        "kernel_ms": {k: round(v["ms"], 4) for k, v in kernels.items()},
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "hbm_frac": {k: round(v["GBps"] / HBM_PEAK_GBPS, 4) for k, v in kernels.items()},
        "roofline": roofline,
    }
    if nogap is not None:
        rec["decode_nogap"] = nogap
        rec["roundtrip_ok"] = ok
    if use_dist and state["plan"] is not None:
        rec["stream_total_bytes"] = state["plan"].total_bytes
    env["last_data"] = data if not use_dist else None
    del out, bufs, batch_hist
    return rec


def bare_decode(ctx, hd, torch, bufs, model, C: int, n: int, out, data, kind: str, cs: int) -> dict:
    """hf_decode_bare on the stream the last step left in `bufs`: the decode of a
    stream that comes without a gap array (the reference's own files,
    src/lib.rs:31-80).  Host clock around the synchronous call, HIP events around
    its two heavy kernels: gap_sync_kernel finds the code boundaries (every lane
    walks its 512 bits of the stream until it meets its neighbour's walk), then
    gap_decode_kernel walks once more and writes the symbols."""
    calls = 5 if n <= (4 << 30) else 3
    out.zero_()
    hd.decode_bare(bufs.data, C, model, n, out)  # warm-up (scratch allocation)
    good = bool(torch.equal(out, data))
    torch.cuda.synchronize()
    for kind_id in (3, 4):
        ctx.kernel_ms(kind_id)
    t0 = time.perf_counter()
    for _ in range(calls):
        hd.decode_bare(bufs.data, C, model, n, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / calls * 1e3
    sync_ms, sync_launches = ctx.kernel_ms(3)
    walk_ms, walk_launches = ctx.kernel_ms(4)
    good = good and bool(torch.equal(out, data))
    alg = C + n
    rec = {
        "entry": "hf_decode_bare",
        "ok": good,
        "calls": calls,
        "ms": round(ms, 4),
        "alg_bytes": alg,
        "GBps": round(alg / (ms * 1e-3) / 1e9, 1),
        "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "decode_GBps_uncompressed": round(n / (ms * 1e-3) / 1e9, 1),
        "kernel_ms": {"gap_sync_kernel": round(sync_ms / calls, 4),
                      "gap_decode_kernel": round(walk_ms / calls, 4)},
        "launches_per_call": {"gap_sync_kernel": sync_launches / calls,
                              "gap_decode_kernel": walk_launches / calls},
    }
    for name, key in (("gap_sync_kernel", "gap_sync"), ("gap_decode_kernel", "gap_decode")):
        traffic, _ = pmc_traffic(kind, n, cs, key)
        if traffic is not None:
            rec.setdefault("traffic", {})[name] = traffic
            rec["traffic_source"] = pmc_traffic(kind, n, cs, "gap_source")[0]
    return rec


def adaptor_rates() -> dict | None:
    """Encoder::write in 4096-byte calls (src/lib.rs:68-77) with and without the
    write() value tracked, Decoder::read in 15-byte calls (examples/decode.rs:55-73)
    with and without the gap array: the C++ mirror of the reference's adaptors over
    host buffers, run as a child process (tools/bin/adaptor_rate --json)."""
    exe = os.path.join(ROOT, "tools", "bin", "adaptor_rate")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "256", "--json"], capture_output=True, text=True, timeout=300)
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"error": str(e)}
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    return {"error": (r.stdout + r.stderr)[-300:]}


def cpu_identity() -> dict:
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 0
    return {"cpu_model": model, "host_cpus": os.cpu_count() or 0, "usable_cpus": usable}


def pmc_traffic(kind: str, n: int, cs: int, kernel: str):
    """HBM bytes per launch of `kernel` from the PMC counters (FETCH_SIZE doubled
    as gfx950 needs, plus WRITE_SIZE), collected in separate rocprofv3 --pmc
    passes over this same workload and recorded in huffman_amd/pmc_traffic.json by
    tools/make_traffic.py -- a recorded figure, not a measurement of this run;
    (None, None) when this workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "huffman_amd", "pmc_traffic.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    entry = table.get(f"{kind}:{n}:{cs}")
    if not entry or kernel not in entry:
        return None, None
    return entry[kernel], entry.get("source")


def cpu_baseline(data, sample_mib: int) -> dict:
    """The CPU restatement of the reference (oracle/oracle.c), single thread, on
    the leading sample of the same stream: from_slice + write + flush + read."""
    import numpy as np

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
        **cpu_identity(),
        "kind": "port",
        "sample": f"first {m >> 20} MiB of the same stream; histogram+codebook {t1 - t0:.2f}s, "
                  f"encode {t2 - t1:.2f}s, decode {t3 - t2:.2f}s",
        "encode_GBps": round(m / (t2 - t0) / 1e9, 4),
        "decode_GBps": round(m / (t3 - t2) / 1e9, 4),
    }


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # HF_BENCH_REHEARSE=1: every rank uses GPU 0 and the collectives run over gloo
    # on host tensors -- a rehearsal of the launcher, the shard plan and the
    # strong-scaling bookkeeping on a one-GPU box (RCCL refuses two ranks on
    # one device); its rates mean nothing
    rehearse = os.environ.get("HF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # HF_BENCH_FORCE_DIST=1 runs the collectives with a single rank too (a
    # rehearsal of the multi-GPU code path on a one-GPU box)
    use_dist = world > 1 or os.environ.get("HF_BENCH_FORCE_DIST") == "1"
    json_fd = None
    rccl_world = None
    if use_dist:
        import torch.distributed as dist  # noqa: F811

        # RCCL prints a version banner on stdout when its first communicator
        # comes up: everything but the result line goes to stderr instead
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        if world == 1 and "RANK" not in os.environ:  # a single rank without a launcher
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                free_port = sk.getsockname()[1]
            for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"),
                             ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(free_port))):
                os.environ.setdefault(key, val)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        rccl_world = dist.get_world_size()
    env = {"dev": dev, "dist": dist, "rank": rank, "world": world,
           "coll_dev": torch.device("cpu") if rehearse else dev}

    single = args.kind is not None or args.gib is not None
    kind = args.kind or "zipf"
    # one logical stream of --total-gib, cut into contiguous shards: the same
    # stream whatever the number of GPUs (--gib instead fixes the bytes per GPU)
    gib = args.gib if args.gib is not None else args.total_gib / world
    steps = args.steps if args.steps is not None else 20
    warmup = args.warmup if args.warmup is not None else 3
    scaling = "strong" if args.gib is None else "weak"
    n = int(gib * GIB)

    head = run_workload(args, env, kind, n, steps, warmup)
    result = {
        "metric": "encode+decode GB/s (uncompressed)",
        "value": head["value"],
        "unit": "GB/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": f"synthetic {kind} bytes, seed {args.seed}+rank, generated in HBM",
        "config": {
            "workload": (f"one {gib * world:g} GiB {kind} stream in {world} contiguous shard(s) of "
                         f"{gib:g} GiB, one per GPU: counting pass + host codebook + encode + "
                         "gap-array decode"),
            "bytes_per_gpu": n,
            "total_bytes": n * world,
            "chunk_syms": args.chunk_syms,
            "enc_mode": "planned" if not args.no_plan else args.enc_mode,
            "compressed_ratio": head["compressed_ratio"],
            "max_code_len": head["max_code_len"],
            "ties": head["ties"],
            "gather_to_root": bool(args.gather),
        },
        "roundtrip_ok": head["roundtrip_ok"],
    }
    if rccl_world is not None:
        result["world_size"] = rccl_world
        result["collectives"] = ("gloo on host tensors (rehearsal)" if rehearse else
                                 "RCCL all-gather of the 256-bin histograms (merge + scan in one)")
    for key in ("encode_GBps_uncompressed", "decode_GBps_uncompressed", "histogram_GBps",
                "kernel_ms", "stage_ms", "hbm_frac", "roofline", "decode_nogap"):
        if key in head:
            result[key] = head[key]
    if "stream_total_bytes" in head:
        result["config"]["stream_total_bytes"] = head["stream_total_bytes"]
    ok = head["roundtrip_ok"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline and env.get("last_data") is not None:
        result["cpu_baseline"] = cpu_baseline(env["last_data"], args.cpu_sample_mib)
    env["last_data"] = None
    torch.cuda.empty_cache()

    if world == 1 and not use_dist and not single and not args.no_extra_configs:
        # the other single-GPU configurations, each with fewer steps as the streams grow
        subs = []
        for sub_kind, sub_gib, sub_steps in (("zipf", 1, 100), ("uniform", 1, 30),
                                             ("geometric", 4, 10)):
            rec = run_workload(args, env, sub_kind, int(sub_gib * GIB), sub_steps, 3)
            env["last_data"] = None
            torch.cuda.empty_cache()
            ok = ok and rec["roundtrip_ok"]
            subs.append(rec)
        result["configs"] = subs

    if rank == 0 and world == 1 and not use_dist and not single and not args.no_adaptors:
        rates = adaptor_rates()
        if rates is not None:
            result["adaptors"] = rates

    if rank == 0:
        line = json.dumps(result)
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (line + "\n").encode())
        else:
            print(line, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


if __name__ == "__main__":
    main()
