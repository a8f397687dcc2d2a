"""Development check (uses the oracle, hence under tests/): 60 random (V, N, K, given) shapes, three KL steps of the fp64
path and of the fp32 fast mode against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import klnmf_oracle as orc
from salamander_amd.engine import Engine
rng = np.random.default_rng(2026)
bad = 0
for case in range(60):
    V = int(rng.choice([1, 7, 16, 33, 83, 96])) if case % 3 else 96
    K = int(rng.integers(1, 65))
    N = int(rng.choice([1, 15, 16, 17, 255, 1024, 4097, 16385, 20000])) if case % 2 else int(rng.integers(1, 30000))
    ng = int(rng.integers(0, K + 1)) if case % 4 == 0 else 0
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=case)
    e = Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
    e.kl_step(3, ng)
    W, H = W0.T, H0.T
    for _ in range(3): W, H = orc.update_WH(X.T, W, H, None, None, ng)
    rw = np.linalg.norm(e.download_W() - W.T) / np.linalg.norm(W); rh = np.linalg.norm(e.download_H() - H.T) / np.linalg.norm(H)
    e.upload_W(W0); e.upload_H(H0); e.set_precision("f32"); e.kl_step(3, ng)
    fw = np.linalg.norm(e.download_W() - W.T) / np.linalg.norm(W); fh = np.linalg.norm(e.download_H() - H.T) / np.linalg.norm(H)
    ok = rw < 1e-11 and rh < 1e-11 and fw < 1e-5 and fh < 1e-5
    bad += not ok
    print(f"case {case}: V={V} N={N} K={K} given={ng}: f64 {rw:.1e} {rh:.1e} | f32 {fw:.1e} {fh:.1e} {'' if ok else '  <-- FAIL'}", flush=True)
    e.close()
print("failures:", bad)
