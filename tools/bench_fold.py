"""Development aid: what does the objective cost when it is folded into the step that follows it?
us per call of kl_step(1), kl_step_objective(slot, 1) plain / kept, objective_async + kl_step(1), at c2."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem

N, V, K = 100000, 96, 50
X, W0, H0 = synthetic_problem(V, N, K, seed=0)
e = sal.Engine(N, V, K)
e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
e.kl_step(20)
e.objective_async(0); e.objective_read(0, 1)

def timed(name, fn, reps=200):
    fn(0); e.sync()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(reps):
            fn(i)
        e.sync()
        best.append((time.perf_counter() - t0) / reps * 1e6)
    print(f"{name:58s} {np.median(best):7.1f} us per call (min {min(best):.1f})")

timed("kl_step(1)", lambda i: e.kl_step(1))
timed("kl_step_keep(1)", lambda i: e.kl_step_keep(1))
timed("kl_step_objective(slot, 1)", lambda i: e.kl_step_objective(i % 250, 1))
timed("kl_step_objective(slot, 1, keep)", lambda i: e.kl_step_objective(i % 250, 1, 0, True))
timed("objective_async(slot) + kl_step(1)", lambda i: (e.objective_async(i % 250), e.kl_step(1)))
timed("objective_async(slot)", lambda i: e.objective_async(i % 250))
timed("kl_step_objective(slot, 10, keep)", lambda i: e.kl_step_objective(i % 250, 10, 0, True), reps=50)
timed("kl_step(10)", lambda i: e.kl_step(10), reps=50)
