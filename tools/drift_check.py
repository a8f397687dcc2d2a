"""Development aid: drift of the GPU trajectory from the reference-generated golden trajectory."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from salamander_amd import Engine
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
g = np.load("tests/golden/kl_pcawg.npz")
X, W0, H0 = g["X"], g["W0"], g["H0"]
e = Engine(X.shape[1], X.shape[0], W0.shape[1]); e.upload_X(X.T); e.upload_W(W0.T); e.upload_H(H0.T)
t = 0; objs = []
for mark in range(10, 501, 10):
    e.kl_step(mark - t); t = mark; objs.append(e.objective())
    if mark in (10, 100, 500):
        print(mark, "W", rel(e.download_W(), g[f"W{mark}"].T), "H", rel(e.download_H(), g[f"H{mark}"].T))
print("obj max rel", np.max(np.abs(np.array(objs) - g["obj"][1:]) / g["obj"][1:]))
