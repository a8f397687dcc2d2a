"""Development aid: time the CorrNMF dense pieces on the GPU.

Shapes follow config c5's per-GPU share: 96 x 50 000 and 96 x 200 000 samples, 40 signatures,
embedding dimension 40 (the reference's default dim_embeddings = n_signatures) and 8.
Each piece is timed as wall clock over a batch of asynchronous launches + one sync.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
