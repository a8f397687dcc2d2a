"""Randomised shapes through the C ABI against the oracle (development aid, not part of the suite): KL steps (joint, H only,
W only; weights, l-half, given signatures, zeros in X), objectives, MvNMF steps, kept blocks and rollback, over
n_samples 1 .. 6000, n_features 1 .. 400, n_signatures 1 .. 200 (feature blocks, signature chunks and both).  `python tests/dev/fuzz_shapes.py [cases] [seed]`."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import klnmf_oracle as orc
from salamander_amd import Engine, _lib


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def one(rng, case):
    N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), rng.integers(700, 6000)]))
    V = int(rng.choice([rng.integers(1, 97), 96, 96, rng.integers(97, 400)]))
    K = int(rng.choice([rng.integers(1, 17), rng.integers(17, 65), rng.integers(1, 65), rng.integers(65, 71), rng.integers(65, 200)]))
    n_given = int(rng.choice([0, 0, rng.integers(0, K + 1)]))
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=int(rng.integers(1 << 30)), mean_mutations=float(rng.choice([30.0, 2000.0, 2e5])))
    if rng.random() < 0.3:
        X = X.copy()
        X[rng.random(X.shape) < 0.1] = 0.0
    weighted = rng.random() < 0.3
    wk = rng.uniform(0.5, 2.0, N) if weighted else None
    wl = rng.uniform(0.0, 0.4, N) if weighted and rng.random() < 0.5 else None
    steps = int(rng.integers(1, 4))
    tag = f"case {case}: N={N} V={V} K={K} n_given={n_given} weighted={weighted} lhalf={wl is not None} steps={steps}"
    e = Engine(N, V, K)
    try:
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        if weighted:
            e.set_weights(wk, wl)
        # (a divergence is a difference of sums of the size of sum x |log x|: near-perfect fits -- one feature, one sample --
        # are compared on that scale)
        scale = float(np.sum(X * np.abs(np.log(np.maximum(X, 1e-300))))) + float(X.sum())
        want = orc.klnmf_objective(X.T, W0.T, H0.T, wk, wl) if weighted else orc.kl_divergence(X.T, W0.T, H0.T)
        got = e.objective()
        assert abs(got - want) <= 1e-10 * abs(want) + 1e-14 * scale, ("objective", got, want)
        W, H = W0.T, H0.T
        for _ in range(steps):
            W, H = orc.update_WH(X.T, W, H, weights_kl=wk, weights_lhalf=wl, n_given_signatures=n_given)
        mode = int(rng.integers(0, 3))
        if mode == 0:
            e.kl_step(steps, n_given)
        elif mode == 1:
            for _ in range(steps):
                e.kl_step(1, n_given)
        else:
            e.kl_step_keep(steps, n_given)
        assert rel(e.download_W(), W.T) < 1e-10 and rel(e.download_H(), H.T) < 1e-10, ("kl_step", mode, rel(e.download_W(), W.T), rel(e.download_H(), H.T))
        if mode == 2:
            e.kl_rollback()
            assert np.array_equal(e.download_W(), W0) and np.array_equal(e.download_H(), H0), "rollback"
            e.kl_step(steps, n_given)
        want = orc.klnmf_objective(X.T, W, H, wk, wl) if weighted else orc.kl_divergence(X.T, W, H)
        got = e.objective()
        assert abs(got - want) <= 1e-10 * abs(want) + 1e-14 * scale, ("objective after", got, want)
        # H alone and W alone from the new state
        e.update_H()
        H2 = orc.update_H(X.T, W, H, weights_kl=wk, weights_lhalf=wl)
        assert rel(e.download_H(), H2.T) < 1e-10, "update_H"
        e.update_W(n_given, _lib.CLIP_NON_GIVEN)
        W2 = orc.update_W(X.T, W, H2, weights_kl=wk, n_given_signatures=n_given)
        assert rel(e.download_W(), W2.T) < 1e-10, "update_W"
        # MvNMF (unweighted), two steps from the start
        if 2 <= K and V >= 2 and not weighted and N >= K and (K <= 64 or V * K <= 30000):  # (one signature has no volume, one feature no freedom: their line searches decide on rounding noise)
            e.set_weights(None, None)
            e.upload_W(W0), e.upload_H(H0)
            Wm, Hm, gam, g2 = W0.T, H0.T, 1.0, 1.0
            ok = True
            for _ in range(2):
                Wm, Hm, gam = orc.mvnmf_step(X.T, Wm, Hm, 1.0, 1.0, gam, n_given)
            g2 = e.mv_step(2, n_given, 1.0, 1.0, 1.0)
            # (the closed-form root differences nearly equal terms: 1e-6 on W after two steps, as the MvNMF tests state)
            assert abs(g2 - gam) <= 1e-12 * gam and rel(e.download_W(), Wm.T) < 1e-6 and rel(e.download_H(), Hm.T) < 1e-6, ("mv_step", g2, gam, rel(e.download_W(), Wm.T), rel(e.download_H(), Hm.T))
        return None
    except Exception as exc:  # noqa: BLE001
        return f"{tag}: {type(exc).__name__}: {exc}\n{traceback.format_exc(limit=2)}"
    finally:
        e.close()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = []
    for c in range(cases):
        r = one(rng, c)
        if r:
            bad.append(r)
            print(r, flush=True)
        if c % 25 == 24:
            print(f"{c + 1} cases, {len(bad)} failures", flush=True)
    print(f"done: {cases} cases, {len(bad)} failures")
    sys.exit(1 if bad else 0)


main()
