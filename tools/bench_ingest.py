"""Development aid: host -> device ingest rate of the count matrix (row f4) at config c3's size (96 x 10^6, 768 MB as
float64) and c2's, per element type; PCIe Gen5 x16 moves at most ~63 GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salamander_amd import Engine

V, K = 96, 50
for N in (100000, 1000000):
    rng = np.random.default_rng(0)
    counts = rng.poisson(20.0, size=(N, V))
    e = Engine(N, V, K)
    for dtype in ("float64", "float32", "int32", "uint16"):
        X = np.ascontiguousarray(counts.astype(dtype))
        e.upload_X(X, clip=True)  # warm: staging buffers, page faults of the source
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); e.upload_X(X, clip=True); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        print(f"N={N} {dtype:8s}: {X.nbytes / 1e6:8.1f} MB in {t * 1e3:7.2f} ms = {X.nbytes / t / 1e9:6.2f} GB/s of source bytes, "
              f"{N * V * 8 / t / 1e9:6.2f} GB/s of float64-equivalent", flush=True)
    H = rng.uniform(0.5, 2.0, size=(N, K))
    e.upload_H(H); t0 = time.perf_counter(); e.upload_H(H); t = time.perf_counter() - t0
    print(f"N={N} H float64: {H.nbytes / 1e6:.1f} MB in {t * 1e3:.2f} ms = {H.nbytes / t / 1e9:.2f} GB/s", flush=True)
    t0 = time.perf_counter(); Hd = e.download_H(); t = time.perf_counter() - t0
    print(f"N={N} download H into a fresh array (page faults included): {H.nbytes / t / 1e9:.2f} GB/s", flush=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); e.download_H(Hd); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print(f"N={N} download H into an array whose pages are touched (what fit() does): {H.nbytes / 1e6:.1f} MB in {t * 1e3:.2f} ms = {H.nbytes / t / 1e9:.2f} GB/s", flush=True)
    e.close()
