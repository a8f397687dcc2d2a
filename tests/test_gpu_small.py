"""Small cohorts: ``kl_step`` as one workgroup, all steps of a call in one launch (``csrc/salnmf_small.hip``).

The kernel reproduces the per-step path's summation orders, so the comparison is bit for bit: same W, H and reduced
numerator after any number of steps, at every workgroup size (1-4 groups of four waves), with one and with several tiles
per wave, ragged N, V < 96, given signatures, and through the kept-block / rollback protocol of the fit loop."""
import numpy as np
import pytest

import salamander_amd as sal
from conftest import rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import Engine

pytestmark = pytest.mark.gpu


def pair(N, V, K, seed=0):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=seed)
    engines = []
    for tiles in (0, 64):  # per-step path, one-workgroup path
        e = Engine(N, V, K)
        e.set_small_cohort_tiles(tiles)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        engines.append(e)
    return X, W0, H0, engines


@pytest.mark.parametrize(
    "N,V,K,n_given",
    [(16, 96, 1, 0), (50, 96, 5, 0), (64, 96, 16, 0), (100, 83, 3, 0), (128, 96, 8, 2), (192, 96, 5, 0), (200, 96, 13, 0), (256, 96, 4, 3),
     (257, 96, 5, 0), (300, 96, 16, 0), (511, 70, 2, 1), (700, 96, 9, 0), (1000, 96, 5, 0), (1024, 96, 16, 5)],
)
def test_one_workgroup_steps_equal_the_per_step_path_bit_for_bit(N, V, K, n_given):
    X, W0, H0, (a, b) = pair(N, V, K, seed=N + K)
    done = 0
    for steps in (1, 2, 7):
        a.kl_step(steps, n_given)
        b.kl_step(steps, n_given)
        done += steps
        assert np.array_equal(a.download_W(), b.download_W()), (steps, "W")
        assert np.array_equal(a.download_H(), b.download_H()), (steps, "H")
        assert a.objective() == b.objective()
    # and both follow the oracle
    W, H = W0.T, H0.T
    for _ in range(done):
        W, H = orc.update_WH(X.T, W, H, n_given_signatures=n_given)
    assert rel_l2(b.download_W(), W.T) < 1e-10 and rel_l2(b.download_H(), H.T) < 1e-10
    # the single-parameter updates read the state the kernel left (W in global memory, the reduced numerator)
    a.update_H(), b.update_H()
    a.update_W(n_given), b.update_W(n_given)
    assert np.array_equal(a.download_W(), b.download_W()) and np.array_equal(a.download_H(), b.download_H())
    a.close(), b.close()


@pytest.mark.parametrize("N,K", [(192, 5), (600, 12)])
def test_kept_blocks_rollback_and_queued_objectives_on_the_small_path(N, K):
    X, W0, H0, (a, b) = pair(N, 96, K, seed=3)
    for e in (a, b):
        e.kl_step(3, 0)
        e.kl_step_objective(0, 10, 0, keep=True)
    assert a.objective_read(0, 1)[0] == b.objective_read(0, 1)[0]
    assert np.array_equal(a.download_W(), b.download_W()) and np.array_equal(a.download_H(), b.download_H())
    for e in (a, b):
        e.kl_rollback()
        e.kl_step_objective(1, 4, 0, keep=False)
    assert np.array_equal(a.download_W(), b.download_W()) and np.array_equal(a.download_H(), b.download_H())
    assert a.objective_read(1, 1)[0] == b.objective_read(1, 1)[0]
    a.close(), b.close()


def test_weights_and_wide_problems_stay_on_the_per_step_path():
    X, W0, H0 = orc.synthetic_problem(96, 192, 5, seed=1)
    wk = np.random.default_rng(0).uniform(0.5, 2.0, 192)
    out = []
    for tiles in (0, 64):
        e = Engine(192, 96, 5)
        e.set_small_cohort_tiles(tiles)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.set_weights(wk, None)
        e.kl_step(5, 0)
        out.append((e.download_W(), e.download_H()))
        e.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    e = Engine(32, 96, 3)
    with pytest.raises(RuntimeError):
        e.set_small_cohort_tiles(-1)
    e.close()
