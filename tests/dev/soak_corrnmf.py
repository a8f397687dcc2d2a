"""Development aid: longer CorrNMFDet / MultimodalCorrNMF fits on the GPU -- the ELBO must stay finite and
(up to the tolerance of the inexact embedding solves) non-decreasing, and the device state must stay finite."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
from salamander_amd.models import CorrNMFDet, MultimodalCorrNMF

def report(name, hist, t):
    hist = np.asarray(hist)
    drops = np.diff(hist)
    print(f"{name}: {len(hist)} ELBO values in {t:.1f} s, first {hist[0]:.6e} last {hist[-1]:.6e}, finite={np.isfinite(hist).all()}, "
          f"largest relative drop {max(0.0, -(drops / np.abs(hist[:-1])).min()) if len(drops) else 0:.2e}", flush=True)

X, _, _ = synthetic_problem(96, 20000, 10, seed=3)
np.random.seed(0)
m = CorrNMFDet(n_signatures=10, dim_embeddings=4, init_method="random", min_iterations=200, max_iterations=200, conv_test_freq=5)
t0 = time.perf_counter(); m.fit(sal.AnnData(X), init_kwargs={"seed": 0}); report("CorrNMFDet 96x20000 K=10 dim=4, 200 it", m.history["objective_function"], time.perf_counter() - t0)
assert np.isfinite(m.asignatures.X).all() and np.isfinite(m.adata.obsm["embeddings"]).all()

Xa, _, _ = synthetic_problem(96, 20000, 12, seed=4)
Xb, _, _ = synthetic_problem(83, 20000, 8, seed=5)
np.random.seed(1)
mm = MultimodalCorrNMF(ns_signatures=[12, 8], dim_embeddings=6, init_method="random", min_iterations=100, max_iterations=100, conv_test_freq=5)
t0 = time.perf_counter(); mm.fit(sal.MuData({"sbs": sal.AnnData(Xa), "indel": sal.AnnData(Xb)}), init_kwargs={"seed": 1})
report("MultimodalCorrNMF (96+83)x20000 K=[12,8] dim=6, 100 it", mm.history["objective_function"], time.perf_counter() - t0)
assert np.isfinite(mm.mdata.obsm["embeddings"]).all()

X2, W0, H0 = synthetic_problem(96, 100000, 50, seed=0)
k = sal.models.KLNMF(50, "custom", min_iterations=3000, max_iterations=3000)
t0 = time.perf_counter(); k.fit(sal.AnnData(X2), init_kwargs={"signatures_mat": W0, "exposures_mat": H0}); t = time.perf_counter() - t0
h = np.asarray(k.history["objective_function"])
print(f"KLNMF c2 3000 it in {t:.2f} s: KL {h[0]:.6e} -> {h[-1]:.6e}, monotone non-increasing: {bool((np.diff(h) <= 1e-9 * np.abs(h[:-1])).all())}", flush=True)
mv = sal.models.MvNMF(30, "custom", min_iterations=1000, max_iterations=1000)
X3, W3, H3 = synthetic_problem(96, 100000, 30, seed=2)
t0 = time.perf_counter(); mv.fit(sal.AnnData(X3), init_kwargs={"signatures_mat": W3, "exposures_mat": H3}); t = time.perf_counter() - t0
h = np.asarray(mv.history["objective_function"])
print(f"MvNMF c4 1000 it in {t:.2f} s: objective {h[0]:.6e} -> {h[-1]:.6e}, finite={np.isfinite(h).all()}, gamma={mv._gamma}", flush=True)
