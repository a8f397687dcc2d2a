"""Development aid: the persistent multi-step KL launch against step-by-step launches -- same bits? how fast?"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salamander_amd import synthetic as syn
from salamander_amd import Engine

V, K = 96, 50
for N in [int(a) for a in sys.argv[1:]] or [100000]:
    X, W0, H0 = syn.synthetic_problem(V, min(N, 250000), K, seed=0)
    if N > X.shape[0]:
        reps = N // X.shape[0]; X = np.tile(X, (reps, 1)); H0 = np.tile(H0, (reps, 1))
    N = X.shape[0]
    engines = []
    for on in (True, False):
        e = Engine(N, V, K); e.set_persistent(on); e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
        engines.append(e)
    ep, es = engines
    for steps in (2, 7, 50):
        ep.kl_step(steps); es.kl_step(steps)
        Wp, Ws, Hp, Hs = ep.download_W(), es.download_W(), ep.download_H(), es.download_H()
        print(f"N={N} after +{steps} steps: W differs in {(Wp != Ws).sum()} entries, H in {(Hp != Hs).sum()} (max rel {np.abs(Hp / Hs - 1).max():.1e}); finite {np.isfinite(Wp).all() and np.isfinite(Hp).all()}", flush=True)
    tp, ts = [], []
    for r in range(7):
        for e, out in ((ep, tp), (es, ts)):
            e.sync(); t0 = time.perf_counter(); e.kl_step(200); e.sync(); out.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"N={N}: persistent {statistics.median(tp):.2f} us/step (min {min(tp):.2f}) | per-step launches {statistics.median(ts):.2f} us/step (min {min(ts):.2f}) | ratio {statistics.median(tp) / statistics.median(ts):.4f}", flush=True)
    for e in engines: e.close()
