"""Seeded synthetic byte streams for tests and benchmarks.

The reference draws its test data from an unseeded thread_rng
(src/stats.rs:36-42, examples/generate_testfiles.rs:7-8), so none of its inputs
can be reproduced; these generators give the same distributions from a stated
seed.  `device=None` generates with numpy on the host, otherwise with torch's
Philox generator directly in HBM (no PCIe traffic).
"""
from __future__ import annotations

import numpy as np

TESTFILE_WEIGHTS = [20, 17, 6, 3, 2, 2, 2, 1, 1, 1]  # examples/generate_testfiles.rs:7


def distribution(kind: str) -> np.ndarray:
    """Probability of each of the 256 symbols (rank order = symbol value)."""
    p = np.zeros(256, dtype=np.float64)
    if kind == "zipf":
        k = np.arange(1, 257, dtype=np.float64)
        p[:] = k ** -1.1
    elif kind == "uniform":
        p[:] = 1.0
    elif kind == "testfile":
        p[: len(TESTFILE_WEIGHTS)] = TESTFILE_WEIGHTS
    elif kind == "geometric":
        k = np.arange(256, dtype=np.float64)
        p[:] = 0.5 ** (k + 1)
        p[255] += 0.5 ** 256
    else:
        raise ValueError(f"unknown distribution {kind!r}")
    return p / p.sum()


def generate_host(kind: str, n: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 256, size=n, dtype=np.uint8)
    if kind == "geometric":
        return np.minimum(rng.geometric(0.5, size=n) - 1, 255).astype(np.uint8)
    cdf = np.cumsum(distribution(kind))
    cdf[-1] = 1.0
    return np.minimum(np.searchsorted(cdf, rng.random(n), side="right"), 255).astype(np.uint8)


def generate_device(kind: str, n: int, seed: int = 0, device="cuda", piece: int = 1 << 26):
    """uint8 tensor of n symbols drawn on the GPU, generated piecewise so the
    temporaries stay small next to multi-GiB outputs."""
    import torch

    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    if kind not in ("uniform", "geometric"):
        cdf = torch.tensor(np.cumsum(distribution(kind)), dtype=torch.float64, device=dev)
        cdf[-1] = 1.0
    for off in range(0, n, piece):
        m = min(piece, n - off)
        if kind == "uniform":
            out[off:off + m] = torch.randint(0, 256, (m,), generator=g, device=dev,
                                             dtype=torch.uint8)
        elif kind == "geometric":
            # k = number of leading zeros of a uniform 62-bit integer: P(k) = 2^-(k+1)
            x = torch.randint(1, 1 << 62, (m,), generator=g, device=dev, dtype=torch.int64)
            k = 61 - torch.floor(torch.log2(x.to(torch.float64))).to(torch.int64)
            out[off:off + m] = k.clamp_(0, 255).to(torch.uint8)
        else:
            u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
            k = torch.bucketize(u, cdf, right=True).clamp_(max=255)
            out[off:off + m] = k.to(torch.uint8)
    return out
