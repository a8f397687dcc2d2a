"""A/B at c5: the modalities' signature solves driven side by side (one host thread per modality, the default) against one after
the other.  Two models from the same seed, twelve updates each, alternating; same ELBO expected to the last bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import salamander_amd as sal
from salamander_amd import synthetic as orc
from salamander_amd.models import MultimodalCorrNMF

N, dim = 200000, int(sys.argv[1]) if len(sys.argv) > 1 else 40
Xa, _, _ = orc.synthetic_problem(96, N, 40, seed=1)
Xb, _, _ = orc.synthetic_problem(83, N, 40, seed=2)
models = {}
for side in (True, False):
    mdata = sal.MuData({"sbs": sal.AnnData(Xa.copy()), "indel": sal.AnnData(Xb.copy())})
    np.random.seed(0)
    m = MultimodalCorrNMF(ns_signatures=[40, 40], dim_embeddings=dim, init_method="random")
    m.solve_side_by_side = side
    m._setup_mdata(mdata)
    m._initialize(None, {"seed": 0})
    m._sync_to_device()
    models[side] = m
times = {True: [], False: []}
for step in range(12):
    for side in (True, False):
        m = models[side]
        engines = list(m._engines.values())
        for e in engines: e.sync()
        t0 = time.perf_counter(); m._device_steps(1, None)
        for e in engines: e.sync()
        times[side].append((time.perf_counter() - t0) * 1e3)
for side in (True, False):
    t = times[side][3:]
    print(f"side by side = {side}: updates 4..12 median {np.median(t):.2f} ms (min {min(t):.2f}, max {max(t):.2f}); ELBO {models[side]._device_objective()!r}", flush=True)
