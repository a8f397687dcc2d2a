"""Development aid: time the CorrNMF dense pieces on the GPU.

Shapes follow config c5's per-GPU share: 96 x 50 000 and 96 x 200 000 samples, 40 signatures,
embedding dimension 40 (the reference's default dim_embeddings = n_signatures) and 8.
Each piece is timed as wall clock over a batch of asynchronous launches + one sync.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import klnmf_oracle as orc
from salamander_amd import Engine, _lib

def timed(fn, e, reps=50):
    fn(); e.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    e.sync()
    return (time.perf_counter() - t0) / reps * 1e6

V, K = 96, 40
for N, dim in [(50000, 40), (200000, 40), (200000, 8)]:
    rng = np.random.default_rng(0)
    X, W0, _ = orc.synthetic_problem(V, N, K, seed=1)
    e = Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.corr_configure(dim)
    e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, rng.normal(0, .3, K))
    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, rng.normal(0, .3, (K, dim)))
    e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, rng.normal(0, .3, (N, dim)))
    e.corr_update_sample_scalings(); e.corr_compute_exposures(); e.corr_compute_aux(); e.sync()
    rows = [("update_sample_scalings", e.corr_update_sample_scalings), ("compute_exposures", e.corr_compute_exposures),
            ("compute_aux (+G reduce)", e.corr_compute_aux), ("update_signature_scalings", e.corr_update_signature_scalings),
            ("update_signatures (tail)", lambda: e.corr_update_signatures(0))]
    print(f"--- N={N} K={K} dim={dim}")
    tot = 0.0
    for name, fn in rows:
        us = timed(fn, e); tot += us
        print(f"{name:28s} {us:8.1f} us")
    t0 = time.perf_counter()
    for _ in range(20): e.corr_poisson_llh()
    print(f"{'poisson_llh (host sync)':28s} {(time.perf_counter()-t0)/20*1e6:8.1f} us")
    print(f"{'dense pieces of one update':28s} {tot:8.1f} us   (6*V*K*N flop of aux at fp64 MFMA peak: {6*V*K*N/78.6e12*1e6:.1f} us)")
    t0 = time.perf_counter(); aux = e.corr_download(_lib.CORR_AUX); dt = time.perf_counter() - t0
    print(f"{'download aux to host':28s} {dt*1e6:8.1f} us ({aux.nbytes/dt/1e9:.1f} GB/s)")
    e.close()

# ---- one full CorrNMFDet update, device resident, vs the NumPy/SciPy oracle on the host cores
import json
from oracle import corrnmf_oracle as co
results = []
for N, dim in [(50000, 40), (200000, 40)]:
    rng = np.random.default_rng(1)
    X, W0, _ = orc.synthetic_problem(V, N, K, seed=2)
    beta = rng.normal(0, .3, K); L = rng.normal(0, .3, (K, dim)); U = rng.normal(0, .3, (N, dim))
    e = Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.corr_configure(dim)
    e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta); e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L); e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
    var = 1.0
    def step():
        global var
        e.corr_update_sample_scalings(); e.corr_compute_exposures(); e.corr_compute_aux(); e.corr_update_signature_scalings()
        e.corr_update_signature_embeddings(var, 0); e.corr_update_sample_embeddings(var, 3)
        a, b = e.corr_embedding_sumsq(); var = max((a + b) / ((K + N) * dim), 1.2e-7)
        e.corr_update_signatures(0)
    times = []
    for i in range(6):
        e.sync(); t0 = time.perf_counter(); step(); e.sync(); times.append(time.perf_counter() - t0)
    # pieces of the last state
    e.sync(); t0 = time.perf_counter(); e.corr_update_signature_embeddings(var, 0); e.sync(); t_sig = time.perf_counter() - t0
    t0 = time.perf_counter(); e.corr_update_sample_embeddings(var, 3); e.sync(); t_smp = time.perf_counter() - t0
    print(f"--- full CorrNMFDet update N={N} K={K} dim={dim}: steps {[round(t*1e3,1) for t in times]} ms; signature solves {t_sig*1e3:.1f} ms, sample solves {t_smp*1e3:.1f} ms", flush=True)
    # host baseline on a bounded sample: the dense pieces in full, the SciPy solves on a subset
    alpha = co.update_sample_scalings(X, beta, L, U)
    t0 = time.perf_counter()
    alpha = co.update_sample_scalings(X, beta, L, U); H = co.compute_exposures(beta, alpha, L, U); aux = co.compute_aux(X, W0, H)
    beta1 = co.update_signature_scalings(aux, alpha, L, U)
    t_dense = time.perf_counter() - t0
    ks = 4
    t0 = time.perf_counter()
    for k in range(ks): co.update_embedding(L[k], U, beta1[k], alpha, 1.0, aux[k])
    t_sig_cpu = (time.perf_counter() - t0) / ks * K
    ns = 1500
    t0 = time.perf_counter()
    for n in range(ns): co.update_embedding(U[n], L, alpha[n], beta1, 1.0, aux[:, n], options={"maxiter": 3})
    t_smp_cpu = (time.perf_counter() - t0) / ns * N
    print(f"    host oracle (extrapolated from {ks} signature and {ns} sample solves): dense {t_dense:.2f} s, signature solves {t_sig_cpu:.1f} s, sample solves {t_smp_cpu:.1f} s", flush=True)
    results.append({"N": N, "K": K, "dim": dim, "device_step_ms": times, "device_signature_solves_ms": t_sig * 1e3, "device_sample_solves_ms": t_smp * 1e3,
                    "host_dense_s": t_dense, "host_signature_solves_s_extrapolated": t_sig_cpu, "host_sample_solves_s_extrapolated": t_smp_cpu,
                    "host_sample": f"{ks} of {K} signature solves, {ns} of {N} sample solves, SciPy Newton-CG via the oracle"})
    e.close()
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out"), exist_ok=True)
json.dump(results, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "corrnmf_step.json"), "w"), indent=1)
