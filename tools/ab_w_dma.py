"""A/B of the W staging of the update passes (Engine.set_w_dma): LDS-DMA against the register path of rounds 1-4,
alternating blocks of steps on ONE resident engine per size (profiles/r05/w_dma.md).

    python tools/ab_w_dma.py            # c2, c3's 125 000-sample shard, 10^6 samples, c4 (MvNMF), smaller cohorts
"""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import salamander_amd as sal  # noqa: E402
from salamander_amd.synthetic import synthetic_problem  # noqa: E402


def ab(N, K, mv=False, steps=200, rounds=9):
    X, W0, H0 = synthetic_problem(96, N, K, seed=0)
    e = sal.Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    run = (lambda n: e.mv_step(n, 0, 1.0, 1.0, 1.0)) if mv else (lambda n: e.kl_step(n))
    run(300)
    e.sync()
    res = {True: [], False: []}
    for r in range(rounds):
        for on in ((True, False) if r % 2 == 0 else (False, True)):
            e.set_w_dma(on)
            run(20)
            e.sync()
            t0 = time.perf_counter()
            run(steps)
            e.sync()
            res[on].append((time.perf_counter() - t0) / steps * 1e6)
    outs = []  # bit-identity of the two forms from the same start
    for on in (True, False):
        e.set_w_dma(on)
        e.upload_W(W0), e.upload_H(H0)
        run(3)
        outs.append((e.download_W(), e.download_H()))
    same = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    e.close()
    a, b = statistics.median(res[True]), statistics.median(res[False])
    print(f"{'MvNMF' if mv else 'KLNMF'} N={N:>8} K={K:>3}: dma {a:8.2f} us/step (min {min(res[True]):.2f})   registers {b:8.2f} (min {min(res[False]):.2f})   "
          f"delta {a - b:+.2f} us   bit-identical: {same}", flush=True)


if __name__ == "__main__":
    ab(100000, 50)
    ab(125000, 50)
    ab(1000000, 50, steps=40, rounds=5)
    ab(100000, 30, mv=True)
    ab(100000, 30)
    ab(20000, 50)
    ab(192, 5)
