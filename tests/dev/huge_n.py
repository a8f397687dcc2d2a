"""A shard sized for the 288 GB of one MI355X: N x 96 and N x 64 both beyond 2^31 elements (default N = 34 000 000: X 26 GB,
H 17 GB on the device).  The problem is R copies of a small one stacked along the sample axis, so the small problem's results
say what the big one must give: the numerator is R times the small one's (W' equal after normalisation, up to summation
order), every sample's new exposures are its template's, the objective is R times the small one's.  Checks the index
arithmetic of every kernel of the KLNMF path beyond 32 bits.  `python tests/dev/huge_n.py [N_big] [N_small]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import klnmf_oracle as orc
from salamander_amd import Engine

N_big = int(sys.argv[1]) if len(sys.argv) > 1 else 34_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
R = N_big // n
N = R * n
V, K = 96, 50
X, W0, H0 = orc.synthetic_problem(V, n, K, seed=3)
Xs = X.astype(np.uint16)  # counts: exact in 16 bits; the clip of the zeros happens on the device
assert np.array_equal(Xs.astype(np.float64), X) or X.max() < 65536
small = Engine(n, V, K)
small.upload_X(Xs, clip=True), small.upload_W(W0), small.upload_H(H0)
obj_s = small.objective()
kl_s = small.samplewise_kl()
small.kl_step(1)
W1s, H1s = small.download_W(), small.download_H()
small.kl_step(2)
W3s, H3s, obj3_s = small.download_W(), small.download_H(), small.objective()
small.close()

t0 = time.perf_counter()
Xb = np.tile(Xs, (R, 1))
Hb = np.tile(H0, (R, 1))
print(f"N = {N:,} = {R} x {n:,}: host arrays {Xb.nbytes / 1e9:.1f} + {Hb.nbytes / 1e9:.1f} GB in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
e = Engine(N, V, K)
e.upload_X(Xb, clip=True), e.upload_W(W0), e.upload_H(Hb)
e.sync()
print(f"engine + uploads {time.perf_counter() - t0:.1f} s (X {N * V * 8 / 1e9:.1f} GB, H {N * 64 * 8 / 1e9:.1f} GB on the device)", flush=True)
del Xb
obj = e.objective()
assert np.isclose(obj, R * obj_s, rtol=1e-12), (obj, R * obj_s)
kl = e.samplewise_kl()
for r in (0, R // 2, R - 1):
    assert np.array_equal(kl[r * n:(r + 1) * n], kl_s), f"samplewise_kl, copy {r}"
t0 = time.perf_counter()
e.kl_step(1)
e.sync()
print(f"one step {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
W1 = e.download_W()
rel = np.linalg.norm(W1 - W1s) / np.linalg.norm(W1s)
assert rel < 1e-13, rel
e.download_H(Hb)
# the H half of the first step uses the OLD W: a sample's new exposures are its template's -- bit for bit where both engines
# ran the sample through the same kind of tile (the leftover round's cooperative tiles sum the remainder columns of U in
# another order: rounding level)
for r in (0, 1, R // 3, R // 2, R - 2, R - 1):
    blk = Hb[r * n:(r + 1) * n]
    same = np.all(blk == H1s, axis=1)
    dev = np.max(np.abs(blk - H1s) / np.maximum(np.abs(H1s), 1e-300))
    print(f"H after one step, copy {r}: {same.mean():.4f} of the rows bit-equal, largest relative deviation {dev:.1e}", flush=True)
    assert same.mean() > 0.95 and dev < 1e-13, f"H after one step, copy {r}"
t0 = time.perf_counter()
e.kl_step(2)
e.sync()
dt = (time.perf_counter() - t0) / 2
print(f"{1e3 * dt:.1f} ms per step = {6.0 * V * K * N / dt / 1e12 / 78.6:.3f} of the fp64 MFMA peak on 6 V K N", flush=True)
W3 = e.download_W()
e.download_H(Hb)
assert np.linalg.norm(W3 - W3s) / np.linalg.norm(W3s) < 1e-12
for r in (0, R // 2, R - 1):
    assert np.linalg.norm(Hb[r * n:(r + 1) * n] - H3s) / np.linalg.norm(H3s) < 1e-12, f"H after three steps, copy {r}"
assert np.isclose(e.objective(), R * obj3_s, rtol=1e-11)
e.update_H(), e.update_W(0)
assert np.all(np.isfinite(e.download_W()))
e.close()
print("huge-N checks passed", flush=True)
