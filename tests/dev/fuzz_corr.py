"""Randomised CorrNMF shapes (development aid): the lockstep signature solves against the one-workgroup-per-signature
form, the batched sample solves against one wavefront per sample, and both against SciPy on a few problems.
`python tests/dev/fuzz_corr.py [cases] [seed]`."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import corrnmf_oracle as co
from oracle import klnmf_oracle as ko
from salamander_amd import Engine, _lib


def one(rng, case):
    N = int(rng.choice([rng.integers(2048, 4000), rng.integers(4000, 12000), rng.integers(12000, 30000)]))
    K = int(rng.choice([rng.integers(1, 8), rng.integers(8, 25), rng.integers(25, 65)]))
    dim = int(rng.choice([K, rng.integers(1, 65), rng.integers(1, 49)]))
    var = float(rng.choice([0.5, 1.0, 2.0]))
    tag = f"case {case}: N={N} K={K} dim={dim} var={var}"
    X, W, _ = ko.synthetic_problem(96, N, K, seed=int(rng.integers(1 << 30)))
    beta, L, U = rng.normal(0, 0.3, K), rng.normal(0, 0.4, (K, dim)), rng.normal(0, 0.4, (N, dim))
    try:
        out = []
        for fast in (True, False):
            e = Engine(N, 96, K)
            e.set_lockstep(fast)
            e.set_batched_sample_solves(fast)
            e.upload_X(X), e.upload_W(W)
            e.corr_configure(dim)
            e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
            e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
            e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
            e.corr_update_sample_scalings()
            e.corr_compute_exposures()
            e.corr_compute_aux()
            st = e.corr_update_signature_embeddings(var, 0, return_status=True)
            Ls = e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
            e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)  # (the two sample solves on identical inputs: three Newton steps
            e.corr_update_sample_embeddings(var, 3)           # of an unconverged solve amplify 1e-9 of L beyond any tolerance)
            Us = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
            out.append((Ls, st, Us, e.corr_download(_lib.CORR_SAMPLE_SCALINGS), e.corr_download(_lib.CORR_AUX)))
            e.close()
        (La, sa, Ua, alpha, aux), (Lb, sb, Ub, _, _) = out
        if dim > 1:  # (one component: line searches end on rounding noise)
            assert np.array_equal(sa, sb), ("status", sa, sb)
        assert np.allclose(La, Lb, rtol=1e-7, atol=1e-10), ("signature embeddings", float(np.max(np.abs(La - Lb))))
        close = np.isclose(Ua, Ub, rtol=1e-6, atol=1e-9).all(axis=1)
        assert close.mean() >= 0.99, ("sample embeddings", float(close.mean()))
        k = int(rng.integers(K))
        want = co.update_embedding(L[k], U, beta[k], alpha, var, aux[:, k])
        assert np.allclose(La[k], want, rtol=1e-5, atol=1e-8), ("scipy", float(np.max(np.abs(La[k] - want))))
        return None
    except Exception as exc:  # noqa: BLE001
        return f"{tag}: {type(exc).__name__}: {exc}\n{traceback.format_exc(limit=1)}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for c in range(cases):
        r = one(rng, c)
        if r:
            bad += 1
            print(r, flush=True)
        if c % 10 == 9:
            print(f"{c + 1} cases, {bad} failures", flush=True)
    print(f"done: {cases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


main()
