"""Wall-clock to fixed KL: the target is the KL the CPU path (NumPy oracle)
reaches after 500 joint steps from the shared init on config c2; the GPU time is the wall-clock of
KLNMF.fit (device-resident loop, objective every 10 steps) until its objective is <= that target.
Prints one JSON object.  Run on the GPU box: python tests/dev/time_to_kl.py [n_samples] [cpu_steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import salamander_amd as sal
from oracle import klnmf_oracle as orc

V, K = 96, 50
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cpu_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
X, W0, H0 = orc.synthetic_problem(V, N, K, seed=0)

t0 = time.perf_counter()
W, H = W0.T.copy(), H0.T.copy()
Xt = np.asfortranarray(X.T)
for _ in range(cpu_steps):
    W, H = orc.update_WH(Xt, W, H)
target = orc.kl_divergence(Xt, W, H)
cpu_s = time.perf_counter() - t0

adata = sal.AnnData(X.copy())
model = sal.models.KLNMF(K, "custom", min_iterations=cpu_steps, max_iterations=cpu_steps)
t0 = time.perf_counter()
model.fit(adata, init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
gpu_fit_s = time.perf_counter() - t0
hist = np.array(model.history["objective_function"])
first = int(np.argmax(hist <= target * (1 + 1e-12))) if (hist <= target * (1 + 1e-12)).any() else -1
# device-resident loop only (no upload/download): same steps + objective cadence on a warm engine
e = sal.Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.upload_H(H0); e.objective(); e.sync()
t0 = time.perf_counter()
for _ in range(cpu_steps // 10):
    e.kl_step(10); obj = e.objective()
loop_s = time.perf_counter() - t0
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print(json.dumps({
    "workload": f"KLNMF n_signatures={K}, synthetic Poisson counts {V}x{N}, shared custom init",
    "cpu_steps": cpu_steps, "target_kl": target, "cpu_seconds_to_target": cpu_s, "cpu_cores": os.cpu_count(),
    "gpu_first_check_at_or_below_target": (first + 1) * 10, "gpu_objective_there": float(hist[first]),
    "gpu_fit_seconds_end_to_end": gpu_fit_s, "gpu_loop_seconds_device_resident": loop_s,
    "speedup_loop": cpu_s / loop_s, "speedup_end_to_end": cpu_s / gpu_fit_s,
    "rel_l2_W_vs_cpu": rel(model.asignatures.X, W.T), "rel_l2_H_vs_cpu": rel(adata.obsm["exposures"], H.T),
}))
