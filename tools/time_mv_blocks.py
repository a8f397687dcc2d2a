"""Development aid: c4 MvNMF steps in calls of 10 (what MvNMF.fit issues: conv_test_freq = 10, the objective of every tenth
state, more_follows) against calls of 50 and 500 -- what a call boundary costs."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, synthetic
N, K = 100000, 30
X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=2)
e = Engine(N, 96, K)
e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
g = e.mv_step(300, 0, 1.0, 1.0, 1.0); e.sync()
for per_call, follows in ((500, False), (50, False), (10, False), (10, True), (10, True), (10, False), (50, True)):
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(500 // per_call):
            g, f = e.mv_step_objective(per_call, 0, 1.0, 1.0, g, more_follows=follows and (i + 1 < 500 // per_call))
        e.sync()
        reps.append((time.perf_counter() - t0) / 500 * 1e6)
    print(f"calls of {per_call:3d} steps, more_follows={follows}: {statistics.median(reps):7.2f} us/step", flush=True)
e.close()
