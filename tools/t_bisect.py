import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.device_init import initialize_on_device
from salamander_amd.synthetic import synthetic_problem
V, N, K = 96, 100000, 50
X, W0, H0 = synthetic_problem(V, N, K, seed=0)
def run(label, xcopy, init, clip, download):
    for rep in range(4):
        Xc = X.copy() if xcopy else X
        t0 = time.perf_counter()
        if clip: Xc = Xc.clip(1e-7, None)
        e = sal.Engine(N, V, K); e.upload_X(Xc)
        if init:
            S = initialize_on_device(e, K, "nndsvd", None, N); e.upload_W(S)
        else:
            e.upload_W(W0); e.upload_H(H0)
        e.set_weights(None, None)
        ts = [time.perf_counter()]
        e.objective(); ts.append(time.perf_counter())
        for it in range(50):
            e.kl_step(10); e.objective(); ts.append(time.perf_counter())
        if download: W = e.download_W(); H = e.download_H()
        t1 = time.perf_counter()
        d = [(b - a) * 1e3 for a, b in zip(ts, ts[1:])]
        print(f"{label} rep {rep}: total {1e3*(t1-t0):.1f} ms, before loop {1e3*(ts[0]-t0):.1f}, loop {1e3*(ts[-1]-ts[0]):.1f} (max interval {max(d):.1f}), after loop {1e3*(t1-ts[-1]):.1f}", flush=True)
        e.close()
run("plain        ", False, False, False, False)
run("+X.copy      ", True, False, False, False)
run("+clip        ", True, False, True, False)
run("+init        ", True, True, True, False)
run("+download    ", True, True, True, True)
