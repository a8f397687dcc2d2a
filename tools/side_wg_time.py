"""Development aid: duration of the MvNMF update_H pass when it has (almost) no tiles -- i.e. of its side workgroup (the
W-only algebra of the next step) -- to be read from a rocprofv3 kernel trace of this script."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
for K in (30, 50):
    X, W0, H0 = synthetic_problem(96, 64, K, seed=1)
    e = sal.Engine(64, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    g = e.mv_step(60, 0, 1.0, 1.0, 1.0)
    e.sync()
    e.close()
