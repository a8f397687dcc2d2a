"""The opt-in fp32 fast mode of the KL step (``salnmf_kernels_f32.h``, ``salnmf_set_precision``) against the oracle.

The tolerances here are the mode's own (``include/salnmf.h``): 1e-5 rel-L2 after 20 steps, 1e-3 after 500 -- the fp64
default path is held to 1e-10 in ``test_gpu_parity.py``.
"""

import numpy as np
import pytest

from conftest import rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _oracle_steps(X, W0, H0, steps, n_given=0):
    W, H = W0.T, H0.T
    for _ in range(steps):
        W, H = orc.update_WH(X.T, W, H, None, None, n_given)
    return W.T, H.T


@pytest.mark.parametrize("V,N,K", [(96, 1000, 50), (96, 333, 5), (83, 2049, 17), (96, 4096, 64), (40, 100, 8), (96, 16, 1)])
def test_fast_mode_steps_within_its_tolerance(V, N, K):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=V + N + K)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_precision("f32")
    e.kl_step(20, 0)
    W, H = _oracle_steps(X, W0, H0, 20)
    assert rel_l2(e.download_W(), W) < 1e-5
    assert rel_l2(e.download_H(), H) < 1e-5
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W.T, H.T), rtol=1e-6)  # fp64 objective of the fp32 iterate


def test_fast_mode_many_steps_and_back_to_fp64():
    V, N, K = 96, 3000, 50
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=9)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_precision("f32")
    objs = []
    for _ in range(10):  # 500 steps in calls of 50, as a fit loop with an objective in between would issue them
        e.kl_step(50, 0)
        objs.append(e.objective())
    assert all(b <= a * (1 + 1e-7) for a, b in zip(objs, objs[1:]))  # the KL keeps decreasing
    W, H = _oracle_steps(X, W0, H0, 500)
    assert rel_l2(e.download_W(), W) < 1e-3
    assert rel_l2(e.download_H(), H) < 1e-3
    assert np.isclose(objs[-1], orc.kl_divergence(X.T, W.T, H.T), rtol=1e-6)
    # back to the default: from the same state the fp64 step is the fp64 step
    Wd, Hd = e.download_W(), e.download_H()
    e.set_precision("f64")
    e.kl_step(3, 0)
    ref = Engine(N, V, K)
    ref.upload_X(X), ref.upload_W(Wd), ref.upload_H(Hd)
    ref.kl_step(3, 0)
    assert np.array_equal(e.download_W(), ref.download_W()) and np.array_equal(e.download_H(), ref.download_H())


def test_fast_mode_given_signatures_new_data_and_refusals():
    V, N, K = 96, 700, 12
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=3)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_precision("f32")
    e.kl_step(10, 4)  # the first four signatures are given: their rows stay
    W, H = _oracle_steps(X, W0, H0, 10, n_given=4)
    Wd = e.download_W()
    assert rel_l2(Wd[:4], W[:4]) < 1e-14  # given rows do not pass through fp32
    assert rel_l2(Wd, W) < 1e-5 and rel_l2(e.download_H(), H) < 1e-5
    e.kl_step(5, K)  # all given: only the exposures move
    W2, H2 = _oracle_steps(X, W, H, 5, n_given=K)
    assert np.array_equal(e.download_W(), Wd)
    assert rel_l2(e.download_H(), H2) < 2e-5
    # new counts: the fp32 copy of X is rebuilt
    X2, _, _ = orc.synthetic_problem(V, N, K, seed=4)
    e.upload_X(X2), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(10, 0)
    W3, H3 = _oracle_steps(X2, W0, H0, 10)
    assert rel_l2(e.download_W(), W3) < 1e-5 and rel_l2(e.download_H(), H3) < 1e-5
    # weights are not part of the mode
    e.set_weights(np.ones(N), None)
    with pytest.raises(RuntimeError, match="no weighted step"):
        e.kl_step(1, 0)
    e.set_weights(None, None)
    with pytest.raises(ValueError):
        e.set_precision("bf16")


def test_klnmf_model_with_fast_precision_reaches_the_fp64_objective():
    import salamander_amd as sal

    V, N, K = 96, 2000, 6
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=21)
    fits = {}
    for precision in ("f64", "f32"):
        model = sal.models.KLNMF(K, "custom", min_iterations=200, max_iterations=200, precision=precision)
        model.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0, "exposures_mat": H0})
        fits[precision] = (np.asarray(model.asignatures.X), float(model.history["objective_function"][-1]))
    assert rel_l2(fits["f32"][0], fits["f64"][0]) < 1e-4
    assert np.isclose(fits["f32"][1], fits["f64"][1], rtol=1e-7)
