"""Diagnostic: where do batched and one-wavefront-per-sample solves differ, and which one is closer to SciPy?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scipy import optimize
from oracle import corrnmf_oracle as co
from salamander_amd import _lib
from test_gpu_corrnmf import embedding_problem, engine_from

N, K, dim, maxiter = (int(v) for v in sys.argv[1:5])
X, W, beta, alpha, L, U, aux = embedding_problem(N, K, dim, seed=N + K + dim)
out = []
for batched in (False, True):
    e = engine_from(X, W, beta, alpha, L, U)
    e.set_batched_sample_solves(batched)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    status = e.corr_update_sample_embeddings(0.8, maxiter, return_status=True)
    out.append((e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS), status))
    e.close()
(Ua, sa), (Ub, sb) = out
scale = np.maximum(np.abs(Ua).max(axis=1), 1e-3)
err = np.abs(Ub - Ua).max(axis=1) / scale
print("quantiles of |batched - per-sample|:", {q: float(np.quantile(err, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)})
print("status agreement", (sa == sb).mean())
idx = np.argsort(-err)[:12]
rng = np.random.default_rng(0)
idx = np.concatenate([idx, rng.choice(N, 40, replace=False)])
opts = {"maxiter": maxiter} if maxiter > 0 else {}
ea, eb = [], []
for n in idx:
    sg = (aux[:, n, None] * L).sum(axis=0)
    fun = lambda x: co.embedding_objective(x, L, alpha[n], beta, 0.8, aux[:, n])
    res = optimize.minimize(fun=fun, x0=U[n].copy(), method="Newton-CG", jac=lambda x: co.embedding_gradient(x, L, alpha[n], beta, 0.8, sg),
                            hess=lambda x: co.embedding_hessian(x, L, alpha[n], beta, 0.8), options=opts)
    sc = max(np.abs(res.x).max(), 1e-3)
    da, db = np.abs(Ua[n] - res.x).max() / sc, np.abs(Ub[n] - res.x).max() / sc
    ea.append(da); eb.append(db)
    print(f"n={n:6d} err(b-a)={err[n]:.2e}  per-sample vs scipy {da:.2e} (status {sa[n]})  batched vs scipy {db:.2e} (status {sb[n]})  scipy status {res.status} nit {res.nit}"
          f"  f0={fun(U[n]):.8e} fa={fun(Ua[n]):.8e} fb={fun(Ub[n]):.8e} fs={res.fun:.8e}")
