"""Development probe for builds with SALNMF_DEV_POISON=1 (every device allocation starts as NaNs): which entry points
return NaN, at which sizes?  A NaN here is a read of memory nobody wrote."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import klnmf_oracle as orc
from salamander_amd import Engine

def probe(N, V, K):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=1)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    out = {}
    out["objective0"] = e.objective()
    out["samplewise"] = float(np.nansum(e.samplewise_kl())) if np.isfinite(e.samplewise_kl()).all() else float("nan")
    e.kl_step(1, 0)
    out["W1"] = float(np.abs(e.download_W()).sum()); out["H1"] = float(np.abs(e.download_H()).sum())
    out["objective1"] = e.objective()
    e.kl_step_objective(0, 1, 0)
    out["obj_in_step"] = float(e.objective_read(0, 1)[0])
    out["mv_objective"] = e.mv_objective(1.0, 1.0)
    g = e.mv_step(1, 0, 1.0, 1.0, 1.0)
    out["mv1_W"] = float(np.abs(e.download_W()).sum())
    g = e.mv_step(3, 0, 1.0, 1.0, g)
    out["mv3_W"] = float(np.abs(e.download_W()).sum()); out["mv3_H"] = float(np.abs(e.download_H()).sum())
    e.close()
    bad = [k for k, v in out.items() if not np.isfinite(v)]
    print(f"N={N} V={V} K={K}: NaN in {bad}" if bad else f"N={N} V={V} K={K}: clean", flush=True)

for N, V, K in [(900, 96, 16), (900, 96, 17), (900, 96, 42), (16384, 96, 20), (16390, 96, 20), (16400, 96, 20), (16450, 96, 1), (33000, 96, 20), (5000, 83, 12)]:
    probe(N, V, K)
