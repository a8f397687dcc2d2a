"""Development aid: device signature-embedding Newton-CG solves vs scipy (via the oracle)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import corrnmf_oracle as co, klnmf_oracle as ko
from salamander_amd import Engine, _lib

def problem(N, K, dim, seed):
    rng = np.random.default_rng(seed)
    X, W, _ = ko.synthetic_problem(96, N, K, seed=seed)
    beta = rng.normal(0, .3, K); L = rng.normal(0, .5, (K, dim)); U = rng.normal(0, .5, (N, dim))
    alpha = co.update_sample_scalings(X, beta, L, U)
    H = co.compute_exposures(beta, alpha, L, U)
    aux = co.compute_aux(X, W, H)
    beta = co.update_signature_scalings(aux, alpha, L, U)
    return X, W, beta, alpha, L, U, aux

cases = [(10, 1, 1), (10, 2, 2), (300, 3, 2), (1000, 7, 3), (2000, 30, 30), (5000, 50, 8), (1500, 64, 64)]
if len(sys.argv) > 1: cases = [(100000, 40, 40)]
for (N, K, dim) in cases:
    X, W, beta, alpha, L, U, aux = problem(N, K, dim, seed=N + K + dim)
    var = 0.8
    e = Engine(N, 96, K); e.upload_X(X); e.upload_W(W); e.corr_configure(dim)
    e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta); e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, alpha)
    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L); e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    e.sync(); t0 = time.perf_counter()
    st = e.corr_update_signature_embeddings(var, 0, return_status=True)
    dt = time.perf_counter() - t0
    got = e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
    t0 = time.perf_counter()
    want = co.update_signature_embeddings(aux, L, U, beta, alpha, var)
    dts = time.perf_counter() - t0
    err = np.abs(got - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-3)
    print(f"N={N} K={K} dim={dim}: max rel err {err.max():.2e}, median {np.median(err):.2e}; status {np.bincount(st, minlength=4)}; device {dt*1e3:.1f} ms, scipy {dts*1e3:.0f} ms", flush=True)
    e.close()
