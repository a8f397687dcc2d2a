"""Config c5 (BASELINE.json): MultimodalCorrNMF on synthetic (96 + 83) x N counts, n_signatures 40 per modality
(ns_signatures = [40, 40], as SURVEY.md 8f reads the config), device resident on ONE GPU.

Times whole `_device_steps(1)` updates (everything on the device) and the pieces; N = 50 000 is the per-GPU
share of c5 on 4 GPUs, N = 200 000 the whole problem on one GPU.  dim_embeddings = 40 (the reference's default
max(ns_signatures)) and 8.
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import salamander_amd as sal
from salamander_amd import synthetic as orc
from salamander_amd.models import MultimodalCorrNMF

out = []
CONFIGS = [(200000, 40)] if __import__("os").environ.get("C5_ONLY_FULL") else [(50000, 40), (200000, 40), (200000, 8)]
for N, dim in CONFIGS:
    Xa, _, _ = orc.synthetic_problem(96, N, 40, seed=1)
    Xb, _, _ = orc.synthetic_problem(83, N, 40, seed=2)
    mdata = sal.MuData({"sbs": sal.AnnData(Xa), "indel": sal.AnnData(Xb)})
    np.random.seed(0)
    model = MultimodalCorrNMF(ns_signatures=[40, 40], dim_embeddings=dim, init_method="random")
    model._setup_mdata(mdata)
    t0 = time.perf_counter(); model._initialize(None, {"seed": 0}); t_init = time.perf_counter() - t0
    t0 = time.perf_counter(); model._sync_to_device(); t_up = time.perf_counter() - t0
    engines = list(model._engines.values())
    steps = []
    for i in range(8):
        for e in engines: e.sync()
        t0 = time.perf_counter(); model._device_steps(1, None)
        for e in engines: e.sync()
        steps.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); obj = model._device_objective(); t_obj = time.perf_counter() - t0
    # pieces on the last state
    def timed(fn):
        for e in engines: e.sync()
        t0 = time.perf_counter(); fn()
        for e in engines: e.sync()
        return (time.perf_counter() - t0) * 1e3
    t_sig = timed(lambda: [e.corr_update_signature_embeddings(model.variance, 0) for e in engines])
    t_sig_par = timed(lambda: list(model._solve_pool(len(engines)).map(lambda e: e.corr_update_signature_embeddings(model.variance, 0), engines)))
    t_smp = timed(lambda: sal.Engine.corr_update_sample_embeddings_multi(engines, model.variance, 3))
    t_aux = timed(lambda: [e.corr_compute_aux() for e in engines])
    print(f"c5 N={N} dim={dim}: update steps {[round(s*1e3,1) for s in steps]} ms; signature solves {t_sig:.1f} ms one modality after the other / {t_sig_par:.1f} ms side by side (one host thread each), joint sample solves {t_smp:.1f} ms, "
          f"aux passes {t_aux:.2f} ms, ELBO {t_obj*1e3:.1f} ms (value {obj:.6e}); host init {t_init:.2f} s, upload {t_up:.2f} s", flush=True)
    out.append({"N": N, "dim": dim, "ns_signatures": [40, 40], "features": [96, 83], "update_ms": [s * 1e3 for s in steps], "signature_solves_ms": t_sig, "signature_solves_side_by_side_ms": t_sig_par,
                "joint_sample_solves_ms": t_smp, "aux_passes_ms": t_aux, "elbo_ms": t_obj * 1e3, "elbo": obj})
    for e in engines: e.close()
    model._engines = {}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c5_mmcorrnmf.json"), "w"), indent=1)
