"""A/B of the CorrNMF sample-embedding solves at c5's shape ((96 + 83) x N, 40 + 40 signatures, dim 40): batched
(sixteen solves per wavefront, MFMA; csrc/salnmf_corr_batched.hip) against one wavefront per sample.  Prints ms per call
on a fitted state (three updates first) and the agreement of the two results."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import salamander_amd as sal
from salamander_amd import _lib
from salamander_amd import synthetic as orc
from salamander_amd.models import MultimodalCorrNMF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
Xa, _, _ = orc.synthetic_problem(96, N, 40, seed=1)
Xb, _, _ = orc.synthetic_problem(83, N, 40, seed=2)
mdata = sal.MuData({"sbs": sal.AnnData(Xa), "indel": sal.AnnData(Xb)})
np.random.seed(0)
model = MultimodalCorrNMF(ns_signatures=[40, 40], dim_embeddings=dim, init_method="random")
model._setup_mdata(mdata)
model._initialize(None, {"seed": 0})
model._sync_to_device()
engines = list(model._engines.values())
for upd in range(int(os.environ.get("UPDATES", "3"))):
    model._device_steps(1, None)
for e in engines:
    e.sync()
U0 = engines[0].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
res = {}
for batched in (False, True, False, True):
    for e in engines:
        e.set_batched_sample_solves(batched)
    ts = []
    for rep in range(3):
        for e in engines:
            e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U0)
            e.sync()
        t0 = time.perf_counter()
        sal.Engine.corr_update_sample_embeddings_multi(engines, model.variance, 3)
        for e in engines:
            e.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    res[batched] = engines[0].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    print(f"N={N} dim={dim} batched={batched}: {[round(t, 2) for t in ts]} ms per call", flush=True)
scale = np.maximum(np.abs(res[False]).max(axis=1), 1e-3)
err = np.abs(res[True] - res[False]).max(axis=1) / scale
print(f"agreement: median {np.median(err):.2e}, 99% {np.quantile(err, 0.99):.2e}, max {err.max():.2e}, share < 1e-8: {(err < 1e-8).mean():.4f}", flush=True)
