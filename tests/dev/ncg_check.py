"""Development aid: device Newton-CG sample-embedding solves vs scipy (via the oracle), error distribution."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import corrnmf_oracle as co, klnmf_oracle as ko
from salamander_amd import Engine, _lib

def problem(N, K, dim, seed):
    rng = np.random.default_rng(seed)
    X, W, _ = ko.synthetic_problem(96, N, K, seed=seed)
    beta = rng.normal(0, .3, K); L = rng.normal(0, .7, (K, dim)); U = rng.normal(0, .7, (N, dim))
    alpha = co.update_sample_scalings(X, beta, L, U)
    H = co.compute_exposures(beta, alpha, L, U)
    aux = co.compute_aux(X, W, H)
    return X, W, beta, alpha, L, U, aux

for (N, K, dim, maxiter) in [(400, 1, 1, 3), (400, 2, 2, 3), (400, 7, 3, 3), (300, 30, 30, 3), (300, 50, 8, 3), (200, 64, 64, 3), (300, 7, 3, 0), (200, 30, 30, 0)]:
    X, W, beta, alpha, L, U, aux = problem(N, K, dim, seed=N + K + dim)
    var = 0.8
    e = Engine(N, 96, K); e.upload_X(X); e.upload_W(W); e.corr_configure(dim)
    e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta); e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, alpha)
    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L); e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    t0 = time.perf_counter()
    st = e.corr_update_sample_embeddings(var, maxiter, return_status=True)
    dt = time.perf_counter() - t0
    got = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    t0 = time.perf_counter()
    want = np.array(U)
    opts = {"options": {"maxiter": maxiter}} if maxiter > 0 else {}
    for n in range(N):
        want[n] = co.update_embedding(U[n], L, alpha[n], beta, var, aux[:, n], **opts)
    dts = time.perf_counter() - t0
    err = np.abs(got - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-3)
    print(f"N={N} K={K} dim={dim} maxiter={maxiter}: max rel err {err.max():.2e}, median {np.median(err):.2e}, >1e-8: {(err>1e-8).sum()}, >1e-5: {(err>1e-5).sum()}; "
          f"status counts {np.bincount(st, minlength=4)}; device {dt*1e3:.1f} ms, scipy {dts*1e3:.0f} ms")
    e.close()
