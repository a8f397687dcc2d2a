"""Ad-hoc GPU check of the engine against the oracle (development aid, not a test)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle import klnmf_oracle as orc
from salamander_amd.engine import Engine

def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))

def run(V, N, K, steps=3, wkl=False, wlh=False, n_given=0, seed=0):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=seed, mean_mutations=2000.0)
    rng = np.random.default_rng(seed + 1)
    wk = rng.uniform(0.5, 2.0, N) if wkl else None
    wl = rng.uniform(0.0, 3.0, N) if wlh else None
    e = Engine(N, V, K)
    e.upload_X(X); e.upload_W(W0); e.upload_H(H0); e.set_weights(wk, wl)
    obj_g = e.objective()
    obj_o = orc.klnmf_objective(X.T, W0.T, H0.T, wk, wl)
    W, H = W0.T.copy(), H0.T.copy()
    for _ in range(steps):
        W, H = orc.update_WH(X.T, W, H, wk, wl, n_given)
    e.kl_step(steps, n_given)
    Wg, Hg = e.download_W(), e.download_H()
    print(f"V={V} N={N} K={K} steps={steps} wkl={wkl} wlh={wlh} given={n_given}: obj rel={abs(obj_g-obj_o)/abs(obj_o):.2e} "
          f"W rel={rel(Wg, W.T):.2e} H rel={rel(Hg, H.T):.2e}")
    sk = e.samplewise_kl(); so = orc.samplewise_kl_divergence(X.T, Wg.T, Hg.T)
    rec = e.reconstruct()
    print(f"    samplewise rel={rel(sk, so):.2e} recon rel={rel(rec, Hg @ Wg):.2e}")
    e.close()

run(96, 10, 2)
run(96, 10, 1)
run(96, 1000, 5)
run(96, 4099, 50)
run(83, 777, 30)
run(96, 2048, 50, wkl=True)
run(96, 2048, 50, wkl=True, wlh=True, n_given=3)
run(96, 2048, 40, wlh=True, n_given=40)
# mvnmf
def run_mv(V, N, K, steps=3, lam=1.0, delta=1.0, n_given=0):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=3)
    e = Engine(N, V, K)
    e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
    og = e.mv_objective(lam, delta); oo = orc.kl_divergence_penalized(X.T, W0.T, H0.T, lam, delta)
    W, H, g = W0.T.copy(), H0.T.copy(), 1.0
    for _ in range(steps):
        W, H, g = orc.mvnmf_step(X.T, W, H, lam, delta, g, n_given)
    gg = e.mv_step(steps, n_given, lam, delta, 1.0)
    Wg, Hg = e.download_W(), e.download_H()
    print(f"MV V={V} N={N} K={K} steps={steps} lam={lam}: obj rel={abs(og-oo)/abs(oo):.2e} W rel={rel(Wg, W.T):.2e} H rel={rel(Hg, H.T):.2e} gamma {gg} vs {g}")
    e.close()
run_mv(96, 10, 2)
run_mv(96, 1000, 30)
run_mv(96, 1000, 30, lam=500.0, steps=5)
# timing
V, N, K = 96, 100000, 50
X, W0, H0 = orc.synthetic_problem(V, N, K, seed=0)
e = Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
e.kl_step(5); e.sync()
tot, fused, tail = e.profile_kl_steps(50, 0, 1)
print(f"c2: total {tot/50*1e3:.1f} us/step, fused kernel {fused*1e3:.1f} us, tail {tail*1e3:.1f} us; fused MFMA-roofline frac = {6*V*K*N/(fused*1e-3)/78.6e12:.3f}")
print(f"forward+objective kernel: {e.profile_objective(20)*1e3:.1f} us -> W@H frac = {2*V*K*N/(e.profile_objective(20)*1e-3)/78.6e12:.3f}")
