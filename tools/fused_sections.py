"""Section clocks of the fused passes (a SALNMF_DEV_PROFILE build: tools/build_variant.sh p -DSALNMF_DEV_PROFILE, run with
SALNMF_LIB=salamander_amd/lib/libsalnmf_p.so): one engine per workload, the table is printed when the engine closes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, synthetic

def run(N, K, mv, steps=200):
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=2)
    e = Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    t0 = time.perf_counter()
    if mv: e.mv_step(steps, 0, 1.0, 1.0, 1.0)
    else: e.kl_step(steps)
    e.sync()
    print(f"--- {'MvNMF queued (MVJ pass)' if mv else 'KLNMF joint step'} N={N} K={K}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us/step in this build", file=sys.stderr, flush=True)
    e.close()

run(100000, 30, True)
run(100000, 50, False)
run(125000, 50, False)
run(100000, 30, False)
