"""Pins ``oracle/corrnmf_oracle.py`` against the reference's own CorrNMF fixtures.

The data files under ``tests/golden/ref_fixtures/corrnmf`` are the reference's
``tests/test_data/models/corrnmf/*`` (inputs and expected outputs of
``tests/test_corrnmf.py:98-175``); the checks mirror that test module.
"""

import os

import numpy as np
import pandas as pd
import pytest

from oracle import corrnmf_oracle as co
from oracle import klnmf_oracle as ko

FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures", "corrnmf")


def load_case(k):
    sfx = f"nsigs{k}_dim{k}.npy"
    g = lambda name: np.load(os.path.join(FIX, f"{name}_{sfx}"))  # noqa: E731
    X = pd.read_csv(os.path.join(FIX, "counts.csv"), index_col=0).T.to_numpy(dtype=np.float64)
    return {
        "X": X,  # (N, V)
        "W": g("signatures_mat_init").T.copy(),  # (K, V)
        "beta": g("signature_scalings_init"),
        "alpha": g("sample_scalings_init"),
        "L": g("signature_embeddings_init").T.copy(),  # (K, dim)
        "U": g("sample_embeddings_init").T.copy(),  # (N, dim)
        "variance": float(g("variance_init")),
        "aux": g("aux"),
        "objective": float(g("objective_init")),
        "W_updated": g("signatures_mat_updated").T,
        "beta_updated": g("signature_scalings_updated"),
        "alpha_updated": g("sample_scalings_updated"),
        "L_updated": g("signature_embeddings_updated").T,
        "U_updated": g("sample_embeddings_updated").T,
        "variance_updated": float(g("variance_updated")),
    }


@pytest.fixture(params=[1, 2])
def case(request):
    return load_case(request.param)


def exposures(c):
    return co.compute_exposures(c["beta"], c["alpha"], c["L"], c["U"])


def test_objective(case):
    got = co.elbo_corrnmf(case["X"], case["W"], exposures(case), case["L"], case["U"], case["variance"])
    assert np.allclose(got, case["objective"])


def test_aux_fixture_is_consistent(case):
    # the stored aux was produced from the initial state: compute_aux must reproduce it
    got = co.compute_aux(case["X"], case["W"], exposures(case))
    assert np.allclose(got, case["aux"])


def test_update_signatures(case):
    H = exposures(case)
    got = ko.update_W(case["X"].T, case["W"].T, H.T).T
    assert np.allclose(got, case["W_updated"])


def test_update_signature_scalings(case):
    got = co.update_signature_scalings(case["aux"], case["alpha"], case["L"], case["U"])
    assert np.allclose(got, case["beta_updated"])


def test_update_sample_scalings(case):
    got = co.update_sample_scalings(case["X"], case["beta"], case["L"], case["U"])
    assert np.allclose(got, case["alpha_updated"])


def test_update_signature_embeddings(case):
    got = co.update_signature_embeddings(case["aux"], case["L"], case["U"], case["beta"], case["alpha"], case["variance"])
    assert np.allclose(got, case["L_updated"])


def test_update_sample_embeddings(case):
    got = co.update_sample_embeddings(case["aux"], case["L"], case["U"], case["beta"], case["alpha"], case["variance"])
    assert np.allclose(got, case["U_updated"])


def test_update_variance(case):
    assert np.allclose(co.update_variance(case["L"], case["U"]), case["variance_updated"])


def test_full_step_runs_and_improves_elbo(case):
    c = case
    H0 = exposures(c)
    before = co.elbo_corrnmf(c["X"], c["W"], H0, c["L"], c["U"], c["variance"])
    W, beta, alpha, L, U, var, _ = co.corrnmf_det_step(c["X"], c["W"], c["beta"], c["alpha"], c["L"], c["U"], c["variance"])
    for _ in range(5):
        W, beta, alpha, L, U, var, _ = co.corrnmf_det_step(c["X"], W, beta, alpha, L, U, var)
    H = co.compute_exposures(beta, alpha, L, U)
    after = co.elbo_corrnmf(c["X"], W, H, L, U, var)
    assert np.isfinite(after) and after > before


# ------------------------------------------------------------------ multimodal (tests/test_mmcorrnmf.py:16-255 data)

MMFIX = os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures", "multimodal_corrnmf")


def load_mm_case():
    g = lambda name: np.load(os.path.join(MMFIX, name + ".npy"))  # noqa: E731
    c = {"Xs": [], "Ws": [], "betas": [], "alphas": [], "Ls": [], "auxs": [], "W_updated": [], "beta_updated": [], "alpha_updated": [], "L_updated": []}
    for m in range(2):
        counts = pd.read_csv(os.path.join(MMFIX, f"model{m}_counts.csv"), index_col=0)  # (V, N)
        X = counts.T.to_numpy(dtype=np.float64)
        c["Xs"].append(X)
        c["Ws"].append(g(f"model{m}_signatures_mat_init").T.copy())
        c["betas"].append(g(f"model{m}_signature_scalings_init"))
        c["alphas"].append(g(f"model{m}_sample_scalings_init"))
        c["Ls"].append(g(f"model{m}_signature_embeddings_init").T.copy())
        # aux_kd = sum_v x_vd p_vkd with the stored p (tests/test_mmcorrnmf.py:101-106)
        c["auxs"].append(np.einsum("vd,vkd->kd", counts.to_numpy(dtype=np.float64), g(f"model{m}_p")))
        c["W_updated"].append(g(f"model{m}_signatures_mat_updated").T)
        c["beta_updated"].append(g(f"model{m}_signature_scalings_updated"))
        c["alpha_updated"].append(g(f"model{m}_sample_scalings_updated"))
        c["L_updated"].append(g(f"model{m}_signature_embeddings_updated").T)
    c["U"] = g("sample_embeddings_init").T.copy()
    c["U_updated"] = g("sample_embeddings_updated").T
    c["variance"] = float(g("variance_init"))
    c["variance_updated"] = float(g("variance_updated"))
    c["objective"] = float(g("objective_init"))
    return c


@pytest.fixture
def mm():
    return load_mm_case()


def mm_exposures(c):
    return [co.compute_exposures(b, a, L, c["U"]) for b, a, L in zip(c["betas"], c["alphas"], c["Ls"])]


def test_mm_objective(mm):
    assert np.allclose(co.mm_elbo(mm["Xs"], mm["Ws"], mm_exposures(mm), mm["Ls"], mm["U"], mm["variance"]), mm["objective"])


def test_mm_aux(mm):
    for X, W, H, aux in zip(mm["Xs"], mm["Ws"], mm_exposures(mm), mm["auxs"]):
        assert np.allclose(co.compute_aux(X, W, H), aux)


def test_mm_signatures_and_scalings(mm):
    Hs = mm_exposures(mm)
    for m in range(2):
        assert np.allclose(ko.update_W(mm["Xs"][m].T, mm["Ws"][m].T, Hs[m].T).T, mm["W_updated"][m])
        assert np.allclose(co.update_sample_scalings(mm["Xs"][m], mm["betas"][m], mm["Ls"][m], mm["U"]), mm["alpha_updated"][m])
        got = co.update_signature_scalings(mm["auxs"][m], mm["alphas"][m], mm["Ls"][m], mm["U"])
        assert np.allclose(got, mm["beta_updated"][m])


def test_mm_signature_embeddings(mm):
    for m in range(2):
        got = co.update_signature_embeddings(mm["auxs"][m], mm["Ls"][m], mm["U"], mm["betas"][m], mm["alphas"][m], mm["variance"])
        assert np.allclose(got, mm["L_updated"][m])


def test_mm_sample_embeddings(mm):
    got = co.mm_update_sample_embeddings(mm["auxs"], mm["Ls"], mm["U"], mm["betas"], mm["alphas"], mm["variance"])
    assert np.allclose(got, mm["U_updated"])


def test_mm_variance(mm):
    assert np.allclose(co.mm_update_variance(mm["Ls"], mm["U"]), mm["variance_updated"])


def test_mm_step_improves_elbo(mm):
    c = mm
    before = co.mm_elbo(c["Xs"], c["Ws"], mm_exposures(c), c["Ls"], c["U"], c["variance"])
    Ws, betas, alphas, Ls, U, var = c["Ws"], c["betas"], c["alphas"], c["Ls"], c["U"], c["variance"]
    for _ in range(5):
        Ws, betas, alphas, Ls, U, var, Hs = co.mm_step(c["Xs"], Ws, betas, alphas, Ls, U, var)
    Hs = [co.compute_exposures(b, a, L, U) for b, a, L in zip(betas, alphas, Ls)]
    assert co.mm_elbo(c["Xs"], Ws, Hs, Ls, U, var) > before


# ------------------------------------------------------------------ vectors produced by executing the reference (corr_synth.npz)

CORR_SYNTH = os.path.join(os.path.dirname(__file__), "golden", "corr_synth.npz")


def load_corr_synth(tag):
    g = np.load(CORR_SYNTH)
    return {k[len(tag) + 1 :]: g[k] for k in g.files if k.startswith(tag + "_")}


@pytest.fixture(params=["a", "b"])
def ref(request):
    return load_corr_synth(request.param)


def test_reference_executed_dense_pieces(ref):
    r = ref
    var = float(r["var"])
    H = co.compute_exposures(r["beta"], r["alpha"], r["L"], r["U"])
    assert np.allclose(H, r["H"], rtol=1e-13, atol=0)
    aux = co.compute_aux(r["X"], r["W"], H)
    assert np.allclose(aux, r["aux"], rtol=1e-12, atol=0)
    assert np.isclose(co.elbo_corrnmf(r["X"], r["W"], H, r["L"], r["U"], var), float(r["elbo"]), rtol=1e-13)
    assert np.isclose(
        co.elbo_corrnmf(r["X"], r["W"], H, r["L"], r["U"], var, penalize_sample_embeddings=False), float(r["elbo_nopen"]), rtol=1e-13
    )
    assert np.allclose(co.update_signature_scalings(aux, r["alpha"], r["L"], r["U"]), r["beta_upd"], rtol=0, atol=1e-13)
    assert np.allclose(co.update_sample_scalings(r["X"], r["beta"], r["L"], r["U"]), r["alpha_upd"], rtol=0, atol=1e-13)


def test_reference_executed_embedding_solves(ref):
    r = ref
    var = float(r["var"])
    L1 = co.update_signature_embeddings(r["aux"], r["L"], r["U"], r["beta"], r["alpha"], var)
    U1 = co.update_sample_embeddings(r["aux"], r["L"], r["U"], r["beta"], r["alpha"], var)
    assert np.allclose(L1, r["L_upd"], rtol=1e-6, atol=1e-9)
    assert np.allclose(U1, r["U_upd"], rtol=1e-6, atol=1e-9)


def test_reference_executed_three_updates(ref):
    r = ref
    W, beta, alpha, L, U, var = r["W"], r["beta"], r["alpha"], r["L"], r["U"], float(r["var"])
    for _ in range(3):
        W, beta, alpha, L, U, var, H = co.corrnmf_det_step(r["X"], W, beta, alpha, L, U, var)
    assert np.allclose(W, r["W3"], rtol=1e-6, atol=1e-12)
    assert np.allclose(H, r["H3"], rtol=1e-6)
    assert np.allclose(beta, r["beta3"], rtol=1e-6, atol=1e-9) and np.allclose(alpha, r["alpha3"], rtol=1e-6, atol=1e-9)
    assert np.allclose(L, r["L3"], rtol=1e-5, atol=1e-7) and np.allclose(U, r["U3"], rtol=1e-5, atol=1e-7)
    assert np.isclose(var, float(r["var3"]), rtol=1e-6)
    assert np.isclose(co.elbo_corrnmf(r["X"], W, H, L, U, var), float(r["elbos"][-1]), rtol=1e-9)


def load_mm_synth():
    g = np.load(CORR_SYNTH)
    mods = [{k: g[f"mm{m}_{k}"] for k in ("X", "W", "beta", "alpha", "L", "aux")} for m in range(2)]
    return mods, g["mm_U"], g["mm_U_upd"], float(g["mm_var"])


def test_reference_executed_joint_sample_solve():
    mods, U, U_upd, var = load_mm_synth()
    got = co.mm_update_sample_embeddings([m["aux"] for m in mods], [m["L"] for m in mods], U, [m["beta"] for m in mods], [m["alpha"] for m in mods], var)
    assert np.allclose(got, U_upd, rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------ the instantiations config c5 uses (corr_c5.npz)
CORR_C5 = os.path.join(os.path.dirname(__file__), "golden", "corr_c5.npz")


def load_c5_sample_solve(it):
    """Inputs and the reference-executed result of the joint sample solve of update ``it`` (1 or 3) of MultimodalCorrNMF at
    c5's shape: ns_signatures [40, 40], dim 40, 256 samples (``tests/golden/make_golden.py --corr-c5``)."""
    g = np.load(CORR_C5)
    mods = [{k: g[f"s{it}_mm{m}_{k}"] for k in ("beta", "alpha", "L", "aux")} for m in range(2)]
    return mods, g[f"s{it}_U"], g[f"s{it}_U_upd"], float(g[f"s{it}_var"])


def load_c5_signature_solve():
    g = np.load(CORR_C5)
    return {k: g["g_" + k] for k in ("U", "L", "beta", "alpha", "aux", "L_upd")}, float(g["g_var"])


def load_c5_trajectory():
    g = np.load(CORR_C5)
    start = {"Xs": [g["mm0_X"], g["mm1_X"]], "Ws": [g["mm0_W0"], g["mm1_W0"]], "betas": [g["mm0_beta0"], g["mm1_beta0"]],
             "Ls": [g["mm0_L0"], g["mm1_L0"]], "U": g["mm_U0"], "var": float(g["mm_var0"])}
    end = {"Ws": [g["mm0_W3"], g["mm1_W3"]], "betas": [g["mm0_beta3"], g["mm1_beta3"]], "alphas": [g["mm0_alpha3"], g["mm1_alpha3"]],
           "Ls": [g["mm0_L3"], g["mm1_L3"]], "Hs": [g["mm0_H3"], g["mm1_H3"]], "U": g["mm_U3"], "var": float(g["mm_var3"]),
           "llh_plus_signature_priors": g["mm_llh_plus_signature_priors"]}
    return start, end


@pytest.mark.parametrize("it", [1, 3])
def test_c5_shape_joint_sample_solves_match_reference_executed_vectors(it):
    """80 terms x dim 40, maxiter 3 (``mmcorrnmf.py:398-428``): the restated solve against what the reference's
    ``update_embedding`` returned for the same 256 problems -- at the random start and on the state two updates later."""
    mods, U, U_upd, var = load_c5_sample_solve(it)
    got = co.mm_update_sample_embeddings([m["aux"] for m in mods], [m["L"] for m in mods], U, [m["beta"] for m in mods], [m["alpha"] for m in mods], var)
    assert np.allclose(got, U_upd, rtol=1e-9, atol=1e-12)


def test_c5_shape_signature_solves_over_2304_samples_match_reference_executed_vectors():
    """dim 40 signature solves (``corrnmf_det.py:88-141``, ``mmcorrnmf.py:319-334``) over 2 304 samples."""
    r, var = load_c5_signature_solve()
    got = co.update_signature_embeddings(r["aux"], r["L"], r["U"], r["beta"], r["alpha"], var)
    assert np.allclose(got, r["L_upd"], rtol=1e-8, atol=1e-11)


def test_c5_shape_three_updates_match_the_reference_executed_trajectory():
    """Three ``MultimodalCorrNMF._update_parameters`` (``mmcorrnmf.py:443-453``) at c5's shape on 256 samples."""
    s, e = load_c5_trajectory()
    Ws, betas, Ls, U, var = s["Ws"], s["betas"], s["Ls"], s["U"], s["var"]
    alphas = [None, None]
    for _ in range(3):
        Ws, betas, alphas, Ls, U, var, Hs = co.mm_step(s["Xs"], Ws, betas, alphas, Ls, U, var)
    for m in range(2):
        assert np.allclose(Ws[m], e["Ws"][m], rtol=1e-6, atol=1e-12) and np.allclose(Hs[m], e["Hs"][m], rtol=1e-6)
        assert np.allclose(betas[m], e["betas"][m], rtol=0, atol=1e-8) and np.allclose(alphas[m], e["alphas"][m], rtol=0, atol=1e-8)
        assert np.allclose(Ls[m], e["Ls"][m], rtol=1e-5, atol=1e-7)
    assert np.allclose(U, e["U"], rtol=1e-5, atol=1e-7) and np.isclose(var, e["var"], rtol=1e-8)
    llh = sum(co.elbo_corrnmf(s["Xs"][m], Ws[m], Hs[m], Ls[m], U, var, penalize_sample_embeddings=False) for m in range(2))
    assert np.isclose(llh, float(e["llh_plus_signature_priors"][-1]), rtol=1e-9)
