"""The resumable Newton-CG (``csrc/salnmf_ncg_machine.h``) against SciPy, on the CPU.

The batched sample-embedding kernel advances sixteen solves per wavefront, one evaluation request per solve and
round; the solver it runs is the state-machine form of ``salnmf_newtoncg.h``.  That header is plain C++ on the host
as well: ``tests/native/ncg_machine_host.cpp`` drives it with a loop evaluator, and this test compares its iterates and
status codes with ``scipy.optimize.minimize(method="Newton-CG")`` -- the reference's call (``_utils_corrnmf.py:400-407``)
-- on CorrNMF sample-embedding problems, without a GPU.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import corrnmf_oracle as co
from oracle import klnmf_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "ncg_machine_host.cpp")


@pytest.fixture(scope="module")
def host_solver(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("ncg") / "ncg_machine_host")
    subprocess.run([gxx, "-O2", "-std=c++17", "-o", exe, SRC], check=True)
    return exe


def embedding_problem(N, K, dim, seed):
    rng = np.random.default_rng(seed)
    X, W, _ = ko.synthetic_problem(96, N, K, seed=seed)
    beta = rng.normal(0, 0.3, K)
    L = rng.normal(0, 0.7, (K, dim))
    U = rng.normal(0, 0.7, (N, dim))
    alpha = co.update_sample_scalings(X, beta, L, U)
    aux = co.compute_aux(X, W, co.compute_exposures(beta, alpha, L, U))
    return beta, alpha, L, U, aux


def run_host(exe, beta, alpha, L, U, aux, var, maxiter):
    K, dim = L.shape
    N = U.shape[0]
    rows = [f"{K} {dim} {N} {maxiter} {var!r}", " ".join(repr(float(v)) for v in L.ravel()), " ".join(repr(float(v)) for v in beta)]
    for n in range(N):
        rows.append(" ".join([repr(float(alpha[n]))] * K))
        rows.append(" ".join(repr(float(v)) for v in aux[:, n]))
        rows.append(" ".join(repr(float(v)) for v in U[n]))
    out = subprocess.run([exe], input="\n".join(rows) + "\n", capture_output=True, text=True, check=True).stdout
    table = np.array([[float(t) for t in line.split()] for line in out.strip().splitlines()])
    return table[:, 2:], table[:, 0].astype(int), table[:, 1].astype(int)


@pytest.mark.parametrize("N,K,dim,maxiter", [(60, 2, 2, 3), (60, 7, 3, 3), (40, 30, 30, 3), (30, 64, 48, 3), (40, 7, 3, 0), (20, 30, 30, 0)])
def test_machine_matches_scipy(host_solver, N, K, dim, maxiter):
    from scipy import optimize

    beta, alpha, L, U, aux = embedding_problem(N, K, dim, seed=N + K + dim)
    var = 0.8
    got, status, rounds = run_host(host_solver, beta, alpha, L, U, aux, var, maxiter)
    want = np.empty_like(U)
    want_status = np.empty(N, dtype=int)
    opts = {"maxiter": maxiter} if maxiter > 0 else {}
    for n in range(N):
        sg = (aux[:, n, None] * L).sum(axis=0)
        res = optimize.minimize(
            fun=lambda x: co.embedding_objective(x, L, alpha[n], beta, var, aux[:, n]),
            x0=U[n].copy(),
            method="Newton-CG",
            jac=lambda x: co.embedding_gradient(x, L, alpha[n], beta, var, sg),
            hess=lambda x: co.embedding_hessian(x, L, alpha[n], beta, var),
            options=opts,
        )
        want[n], want_status[n] = res.x, res.status
    scale = np.maximum(np.abs(want).max(axis=1), 1e-3)
    err = np.abs(got - want).max(axis=1) / scale
    if maxiter > 0:  # the reference's setting: the three iterates agree to rounding
        assert np.median(err) < 1e-12
        assert (err < 1e-8).mean() >= 0.9
    assert err.max() < (2e-4 if maxiter > 0 else 10 * dim * 1e-5)
    assert (status == want_status).mean() >= (0.97 if maxiter > 0 else 0.8)
    assert rounds.min() >= 1


def test_machine_terminates_on_non_finite_input(host_solver):
    beta, alpha, L, U, aux = embedding_problem(8, 3, 2, seed=3)
    U[5, 0] = np.nan
    aux[1, 7] = np.inf
    got, status, rounds = run_host(host_solver, beta, alpha, L, U, aux, 1.0, 3)
    clean = np.ones(8, dtype=bool)
    clean[[5, 7]] = False
    assert np.isfinite(got[clean]).all()
    assert rounds.max() < 5000
