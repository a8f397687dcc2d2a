"""Headline benchmark: KL-NMF update-steps/sec (and wall-clock to fixed KL) at 96 x N, k = 50 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one joint KLNMF update (W and H both updated = one ``update_WH`` call of the reference,
``_utils_klnmf.py:281-361``) of the device-resident state.  fp64, synthetic Poisson counts.

Workload
  * ``--gpus 1``: config c2 of BASELINE.json, 96 x 100 000, k = 50 -- the configuration the metric is quoted on.
  * ``--gpus N`` (N > 1): config c3, 96 x 1 000 000 **in total**, the sample axis split over the ranks with
    ``distributed.shard_bounds`` (**strong scaling**); every step contains one RCCL all-reduce of the 50 x 96
    numerator.  ``value`` = global update-steps/s of the whole job (NOT multiplied by N).  Rank 0 also runs the
    same 10^6-sample problem alone on its GPU and reports that figure (``config.one_gpu_same_problem``) --
    the number an N-GPU value has to be compared with; c2 at N = 1 is a different (10x smaller) problem.
    ``--weak`` restores round 1's weak-scaling reading (100 000 samples per GPU, value = N x global steps/s).

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier +
stream/torch synchronisation on both sides; a rank's clock runs from the opening barrier to the completion of its own
K steps, and the block's time is the maximum over the ranks (the closing barrier's own latency is not counted).  One block of 20 steps is 1.6 ms,
too short to carry a number, so the block is repeated until the GPU has been busy for ~2.5 s and the line
reports the MEDIAN block (``ms_per_step`` = median / K, with min / max / first block beside it).  Events and
buffers are created by an untimed pass first.  Inputs are resident in HBM before any timed region.

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel (the fused update pass) on algorithmic
flops 6*V*K*N_local against the fp64 MFMA peak; its duration is the mean over >= 100 launches sampled with HIP
events on the engine's stream in a separate, untimed pass.  ``cpu_baseline`` times the NumPy oracle (the
restated reference arithmetic) on this box's host cores (rank 0, N = 1 only); the same CPU run defines the
target of ``time_to_kl``.  ``extra`` carries the other single-GPU configurations (c4 MvNMF, default-init fit).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V, K = 96, 50
N_C2 = 100000        # config c2 (1 GPU)
N_C3 = 1000000       # config c3 (sharded)
BLOCK_ROWS = 125000  # generation block of the 10^6 problem (seed = block index)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (spec); 77.8 measured by tools/mfma_f64_probe.hip
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X fp32 matrix peak (MI355X_MICROARCH.md); only the opt-in fast mode is priced against it


def cpu_baseline(X, W0, H0, max_steps=500, budget_s=150.0):
    """The oracle's update_WH on the host cores from the shared init: 3 untimed warm-up steps on copies, then up
    to ``max_steps`` timed steps (stopping early at a multiple of 10 once ``budget_s`` is spent).  Returns the
    baseline record and (steps, objective after those steps, seconds, W, H after those steps -- sample-major, as the engine
    holds them) for time_to_kl and the north star's parity gate."""
    from oracle import klnmf_oracle as orc

    Xt = np.asfortranarray(X.T)  # the layouts the reference computes on
    W, H = W0.T.copy(), H0.T.copy()
    for _ in range(3):
        W, H = orc.update_WH(Xt, W, H)
    W, H = W0.T.copy(), H0.T.copy()
    n, t0 = 0, time.perf_counter()
    while n < max_steps:
        W, H = orc.update_WH(Xt, W, H)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s and n % 10 == 0:
            break
    dt = time.perf_counter() - t0
    target = float(orc.kl_divergence(Xt, W, H))
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    rec = {
        "value": n / dt,
        "unit": "update-steps/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"{n} timed update_WH steps from the shared init (after 3 warm-up steps) of the NumPy oracle on the full "
        f"96x{X.shape[0]} k={K} workload, numpy {np.__version__}, {dt:.1f} s",
    }
    return rec, (n, target, dt, np.ascontiguousarray(W.T), np.ascontiguousarray(H.T))


def device_loop_to_target(e, target, limit):
    """Steps on a resident engine until the objective (every 10 steps, as fit evaluates it) is at or below ``target``.
    The host stays out of the way as ``KLNMF.fit`` does past ``min_iterations``: the objective is queued TOGETHER with the
    next block of 10 steps (``kl_step_objective``: evaluated inside that block's first update where the engine can; a
    kept block), read while the block runs, and the block is rolled back once the target is met -- the clock stops when
    the host knows.  Returns (steps, objective there, seconds)."""
    e.objective_async(0)
    e.objective_read(0, 1)
    e.sync()
    t0 = time.perf_counter()
    e.kl_step(10)
    steps, slot = 10, 1
    while True:
        e.kl_step_objective(slot, 10, 0, keep=True)
        obj = float(e.objective_read(slot, 1)[0])
        if obj <= target * (1 + 1e-12) or steps >= limit:
            seconds = time.perf_counter() - t0
            e.kl_rollback()
            return steps, obj, seconds
        steps += 10
        slot = 1 + slot % 250


def parity_gate(e, W0, H0, cpu_steps, W_cpu, H_cpu):
    """BASELINE.json's acceptance figure: W, H within 1e-4 rel-L2 of the CPU path after EQUAL iterations from the same init
    (``_utils_klnmf.py:281-361`` restated by the oracle, run by ``cpu_baseline`` on this box)."""
    e.upload_W(W0), e.upload_H(H0)
    e.kl_step(cpu_steps)
    W, H = e.download_W(), e.download_H()
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    rw, rh = rel(W, W_cpu), rel(H, H_cpu)
    return {"parity_steps": cpu_steps, "rel_l2_W": rw, "rel_l2_H": rh, "parity_gate": 1e-4, "parity_ok": bool(max(rw, rh) <= 1e-4)}


def time_to_kl(sal, X, W0, H0, cpu_steps, target, cpu_seconds, device, W_cpu=None, H_cpu=None):
    """Wall-clock until the device-resident loop (objective every 10 steps, as fit does) is at or below the KL
    the CPU path reached after ``cpu_steps`` steps from the same init; then the same through KLNMF.fit.  With the CPU
    path's factors: the north star's parity gate after the same number of steps (``parity_gate``)."""
    N = X.shape[0]
    e = sal.Engine(N, V, K, device=device)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(300)  # (20 ms of work: the GPU's clocks are back up after the idle minutes of the CPU baseline) ...
    e.upload_W(W0), e.upload_H(H0)  # ... and the loop starts from the shared init
    steps, obj, loop_s = device_loop_to_target(e, target, cpu_steps + 100)
    gate = parity_gate(e, W0, H0, cpu_steps, W_cpu, H_cpu) if W_cpu is not None else {}
    e.close()
    fit_runs = []
    for _ in range(3):  # (the host side of a fit -- 77 MB of fresh pages for X.clip, the pinned staging -- varies by tens of ms)
        adata = sal.AnnData(X.copy())
        model = sal.models.KLNMF(K, "custom", min_iterations=steps, max_iterations=steps, device=device)
        init_kwargs = {"signatures_mat": W0.copy(), "exposures_mat": H0.copy()}  # (the caller's copies are not part of the fit)
        t0 = time.perf_counter()
        model.fit(adata, init_kwargs=init_kwargs)
        fit_runs.append(time.perf_counter() - t0)
    fit_s = statistics.median(fit_runs)
    return {
        "target": f"KL the NumPy oracle reaches after {cpu_steps} update_WH steps from the shared init",
        "cpu_steps": cpu_steps,
        "target_kl": target,
        "cpu_seconds": cpu_seconds,
        "gpu_steps_to_target": steps,
        "gpu_objective_there": obj,
        "reached": bool(obj <= target * (1 + 1e-12)),
        "gpu_loop_seconds": loop_s,
        "gpu_loop_protocol": "objective every 10 steps, evaluated inside the first update of the next block, which is queued before the deciding objective is read (kept block, rolled back at the target)",
        "gpu_fit_seconds_end_to_end": fit_s,
        "gpu_fit_seconds_runs": fit_runs,
        "fit_objective_last": float(model.history["objective_function"][-1]) if model.history["objective_function"] else None,
        **gate,
    }


def extra_c4(sal, device):
    """Config c4: MvNMF n_signatures=30 on 96 x 100 000 (lam = delta = 1), device-resident steps."""
    from salamander_amd.synthetic import synthetic_problem

    Kc = 30
    X, W0, H0 = synthetic_problem(V, N_C2, Kc, seed=2)
    e = sal.Engine(N_C2, V, Kc, device=device)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    g = e.mv_step(300, 0, 1.0, 1.0, 1.0)  # (30 ms: the clocks are back up after the CPU baseline's idle minutes)
    e.sync()
    blocks = []
    for _ in range(9):
        t0 = time.perf_counter()
        g = e.mv_step(50, 0, 1.0, 1.0, g)
        e.sync()
        blocks.append((time.perf_counter() - t0) / 50)
    e.close()
    t = statistics.median(blocks)
    flops = 12.0 * V * Kc * N_C2  # >= 12 V K N per MvNMF iteration
    # the model-level loop (mvnmf.py:197-210 inside signature_nmf.py:358-385): 500 iterations, a convergence test every 10
    fits = []
    for _ in range(2):
        model = sal.models.MvNMF(Kc, "custom", lam=1.0, delta=1.0, min_iterations=500, max_iterations=500, device=device)
        adata = sal.AnnData(X.copy())
        init_kwargs = {"signatures_mat": W0.copy(), "exposures_mat": H0.copy()}
        t0 = time.perf_counter()
        model.fit(adata, init_kwargs=init_kwargs)
        fits.append(time.perf_counter() - t0)
    return {
        "fit_seconds_500_iterations": min(fits),
        "fit_objective_last": float(model.history["objective_function"][-1]),
        "workload": f"c4: MvNMF n_signatures={Kc}, {V}x{N_C2}, lam=delta=1, 9 blocks of 50 device-resident steps (median) after 300 warm-up steps",
        "us_per_step": t * 1e6,
        "steps_per_s": 1.0 / t,
        "us_per_step_min": min(blocks) * 1e6,
        "us_per_step_max": max(blocks) * 1e6,
        "algorithmic_flops_per_step": flops,
        "frac_of_fp64_mfma_peak": flops / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        # the reference's algorithm forms 6 products of 2 V K N per iteration (update_H 2, numerator 2, f0 and the first trial's
        # f1 one each); the one-pass step shares P between update_H and the trial's KL and between the numerator and f0, so
        # the kernel EXECUTES 4 products = 8 V K N on the matrix cores (SQ_INSTS_MFMA agrees: profiles/r05/mvj_sections.md)
        "executed_mfma_flops_per_step": 8.0 * V * Kc * N_C2,
        "frac_of_fp64_mfma_peak_executed": 8.0 * V * Kc * N_C2 / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        "gamma_after": g,
    }


def extra_default_init_fit(sal, device):
    """KLNMF(50).fit(adata) with the DEFAULT init_method on c2: seconds of initialisation next to the loop."""
    from salamander_amd.synthetic import synthetic_problem

    X, _, _ = synthetic_problem(V, N_C2, K, seed=0)
    # once per process, not per fit (pinned staging buffers' first allocation, LAPACK's first eigh): a 2 000-sample fit first
    warm = sal.models.KLNMF(K, min_iterations=1, max_iterations=1, device=device)
    warm.fit(sal.AnnData(X[:2000].copy()))
    adata = sal.AnnData(X.copy())
    model = sal.models.KLNMF(K, min_iterations=500, max_iterations=500, device=device)
    t0 = time.perf_counter()
    model._setup_adata(adata)
    model._initialize(None, None)
    init_s = time.perf_counter() - t0
    adata = sal.AnnData(X.copy())
    model = sal.models.KLNMF(K, min_iterations=500, max_iterations=500, device=device)
    t0 = time.perf_counter()
    model.fit(adata)
    fit_s = time.perf_counter() - t0
    return {
        "workload": f"KLNMF({K}).fit(adata), init_method='{model.init_method}' (the default), {V}x{N_C2}, 500 iterations; "
        "after one small warm-up fit in the process",
        "init_seconds": init_s,
        "fit_seconds_end_to_end": fit_s,
        "objective_last": float(model.history["objective_function"][-1]),
    }


def extra_fast_mode(sal, device):
    """The opt-in fp32 fast mode of the KL step at c2 (NOT the headline: `value` and `roofline` are the fp64 path's)."""
    from salamander_amd.synthetic import synthetic_problem

    X, W0, H0 = synthetic_problem(V, N_C2, K, seed=0)
    res = {}
    state = {}
    for mode in ("f64", "f32"):
        e = sal.Engine(N_C2, V, K, device=device)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.set_precision(mode)
        e.kl_step(100)
        state[mode] = (e.download_W(), e.download_H(), e.objective())
        e.sync()
        blocks = []
        for _ in range(7):
            t0 = time.perf_counter()
            e.kl_step(200)
            e.sync()
            blocks.append((time.perf_counter() - t0) / 200)
        res[mode] = statistics.median(blocks)
        e.close()
    flops = 6.0 * V * K * N_C2
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    return {
        "workload": f"c2 through Engine.set_precision('f32'): {V}x{N_C2}, k={K}, 7 blocks of 200 steps per kl_step call (median), "
        "fp32 copies of X and H made / written back inside every call",
        "us_per_step_f32": res["f32"] * 1e6,
        "us_per_step_f64_same_protocol": res["f64"] * 1e6,
        "speedup": res["f64"] / res["f32"],
        "step_frac_of_fp32_mfma_peak": flops / res["f32"] / 1e12 / FP32_MFMA_PEAK_TFLOPS,
        "after_100_steps_rel_l2_vs_f64": {"W": rel(state["f32"][0], state["f64"][0]), "H": rel(state["f32"][1], state["f64"][1])},
        "after_100_steps_objective": {"f32": state["f32"][2], "f64": state["f64"][2]},
    }


def extra_c5(sal, device):
    """Config c5 on ONE GPU: MultimodalCorrNMF, (96 + 83) x 200 000 counts, 40 + 40 signatures, dim_embeddings 40 --
    whole device-resident updates (mmcorrnmf.py:430-470: scalings, exposures, aux, signature and sample embedding solves,
    variance, signatures), the first two (allocations) left out."""
    from salamander_amd.models import MultimodalCorrNMF
    from salamander_amd.synthetic import synthetic_problem

    N5 = 200000
    Xa, _, _ = synthetic_problem(96, N5, 40, seed=1)
    Xb, _, _ = synthetic_problem(83, N5, 40, seed=2)
    mdata = sal.MuData({"sbs": sal.AnnData(Xa), "indel": sal.AnnData(Xb)})
    np.random.seed(0)
    model = MultimodalCorrNMF(ns_signatures=[40, 40], dim_embeddings=40, init_method="random", device=device)
    model._setup_mdata(mdata)
    model._initialize(None, {"seed": 0})
    model._sync_to_device()
    engines = list(model._engines.values())
    steps = []
    for _ in range(8):
        for e in engines:
            e.sync()
        t0 = time.perf_counter()
        model._device_steps(1, None)
        for e in engines:
            e.sync()
        steps.append((time.perf_counter() - t0) * 1e3)
    elbo = float(model._device_objective())
    for e in engines:
        e.close()
    model._engines = {}
    return {
        "workload": "c5 on one GPU: MultimodalCorrNMF (96 + 83) x 200000, ns_signatures [40, 40], dim_embeddings 40, 8 device-resident updates",
        "update_ms": steps,
        "update_ms_median_after_warmup": statistics.median(steps[2:]),
        "elbo_after": elbo,
    }


def extra_weighted_step(sal, device):
    """The weighted instantiation of the joint step at c2 (KLNMF(fitting_kwargs={"weights_kl": ..., "weights_lhalf": ...})):
    per-sample weights select fused_kernel<..., WTS = true> (no cooperative leftover tile, conditional loads in the tile loop)."""
    from salamander_amd.synthetic import synthetic_problem

    X, W0, H0 = synthetic_problem(V, N_C2, K, seed=0)
    rng = np.random.default_rng(1)
    res = {}
    for name, (wk, wl) in (("weights_kl", (rng.uniform(0.5, 2.0, N_C2), None)), ("weights_kl_and_lhalf", (rng.uniform(0.5, 2.0, N_C2), rng.uniform(0.0, 0.2, N_C2)))):
        e = sal.Engine(N_C2, V, K, device=device)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.set_weights(wk, wl)
        e.kl_step(50)
        e.sync()
        blocks = []
        for _ in range(7):
            t0 = time.perf_counter()
            e.kl_step(200)
            e.sync()
            blocks.append((time.perf_counter() - t0) / 200)
        e.close()
        res[name] = statistics.median(blocks) * 1e6
    return {"workload": f"c2 with per-sample weights: {V}x{N_C2}, k={K}, 7 blocks of 200 steps (median)", "us_per_step": res}


def extra_wide_catalogue(sal, device):
    """A wide catalogue (SBS-288 contexts) and more than 64 signatures at c2's number of samples: the joint step of engines with
    feature blocks (one pass per 96-feature block) and signature chunks (the reference has no limit on either size)."""
    from salamander_amd.synthetic import synthetic_problem

    res = {}
    for name, (Vw, Kw) in (("288x50", (288, 50)), ("96x100", (96, 100)), ("288x100", (288, 100))):
        X, W0, H0 = synthetic_problem(Vw, N_C2, Kw, seed=0)
        e = sal.Engine(N_C2, Vw, Kw, device=device)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        del X, H0
        e.kl_step(20)
        e.sync()
        blocks = []
        for _ in range(5):
            t0 = time.perf_counter()
            e.kl_step(50)
            e.sync()
            blocks.append((time.perf_counter() - t0) / 50)
        e.close()
        t = statistics.median(blocks)
        res[name] = {"us_per_step": t * 1e6, "frac_of_fp64_mfma_peak_on_6VKN": 6.0 * Vw * Kw * N_C2 / t / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    return {"workload": f"KLNMF joint steps at {N_C2} samples, n_features x n_signatures beyond one 96-feature block / one 64-signature chunk, 5 blocks of 50 steps (median)",
            "engines": res}


def problem_rows(start, stop):
    """Rows [start, stop) of the 10^6-sample problem: 125 000-row blocks, block b drawn with seed b
    (per-shard seeds, no 8 GB host temporary).  W0 is block 0's."""
    from salamander_amd.synthetic import synthetic_problem

    Xs, Hs = [], []
    for b in range(start // BLOCK_ROWS, (stop - 1) // BLOCK_ROWS + 1):
        Xb, _, Hb = synthetic_problem(V, BLOCK_ROWS, K, seed=b)
        lo, hi = max(start, b * BLOCK_ROWS) - b * BLOCK_ROWS, min(stop, (b + 1) * BLOCK_ROWS) - b * BLOCK_ROWS
        Xs.append(Xb[lo:hi])
        Hs.append(Hb[lo:hi])
    return np.concatenate(Xs), np.concatenate(Hs)


def validate_exchange(engine, dist, W0, H0, ctrl, steps=3, reference="rccl"):
    """A few steps from the same start through the reference collective and through the peer-to-peer exchange: W must
    agree to rounding (the two add the ranks' numerators in different orders) and, with the exchange, be bit-identical on
    all ranks.  The reference is the engine's RCCL communicator, or -- where that could not be created -- the split step
    with ``torch.distributed.all_reduce`` between its halves (``reference="host"``)."""
    import torch

    out = {"p2p_valid": False, "validation_steps": steps, "validated_against": reference}
    rel, digest, good = float("inf"), 0.0, 0.0
    W, failure = {}, None
    # (the host-collective reference calls torch.distributed: it runs outside the try block, on every rank alike)
    if reference == "host":
        from salamander_amd.distributed import HostCollectiveAdapter, host_collective_steps

        engine.upload_W(W0)  # (the split step exchanges nothing itself: the collective between its halves is torch's)
        engine.upload_H(H0)
        host_collective_steps(HostCollectiveAdapter(engine), steps)
        W["ref"] = engine.download_W()
    try:  # (no torch.distributed call inside: a rank that fails must still meet the others in the all-reduce below)
        for mode in (("rccl", "p2p") if reference == "rccl" else ("p2p",)):
            engine.set_p2p(mode == "p2p")
            engine.upload_W(W0)
            engine.upload_H(H0)
            engine.kl_step(steps)
            W["ref" if mode == "rccl" else mode] = engine.download_W()
        rel = float(np.linalg.norm(W["p2p"] - W["ref"]) / np.linalg.norm(W["ref"]))
        digest = float(np.frombuffer(W["p2p"].tobytes(), dtype=np.uint32).astype(np.uint64).sum() % (1 << 52))
        good = 1.0 if rel < 1e-11 else 0.0
    except RuntimeError as exc:  # e.g. an exchange that gave up waiting for a peer
        out["error"] = str(exc)
    t = torch.tensor([digest, -digest, good], dtype=torch.float64, device=ctrl)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    same = bool(t[0].item() == -t[1].item())
    out.update({"p2p_valid": bool(t[2].item() == 1.0 and same), "p2p_vs_rccl_rel_l2_W": rel, "p2p_W_identical_on_all_ranks": same})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle run (and time_to_kl)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra single-GPU configurations")
    ap.add_argument("--cpu-steps", type=int, default=500, help="CPU oracle steps that define the time-to-KL target")
    ap.add_argument("--cpu-budget", type=float, default=150.0, help="seconds after which the CPU run stops early")
    ap.add_argument("--cpu-steps-sharded", type=int, default=20, help="N > 1: CPU oracle steps on the whole sharded problem (each takes seconds)")
    ap.add_argument("--samples-total", type=int, default=0, help="override the total number of samples")
    ap.add_argument("--weak", action="store_true", help="N > 1: 100 000 samples per GPU instead of 10^6 in total")
    ap.add_argument("--no-one-gpu-reference", action="store_true", help="N > 1: skip rank 0's run of the whole problem on one GPU")
    ap.add_argument("--busy-seconds", type=float, default=2.5, help="repeat the K-step block until the GPU was busy this long")
    ap.add_argument("--no-p2p", action="store_true", help="N > 1: RCCL all-reduce only, do not try the peer-to-peer exchange")
    ap.add_argument("--simulate-no-rccl", action="store_true", help="rehearsal aid: behave as if the engine's RCCL communicator could not be created")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank's engine on device 0, control plane over gloo, the engine's RCCL communicator "
                    "treated as absent (two RCCL ranks cannot share a GPU), the K x V exchange by the peer-to-peer protocol between the "
                    "processes -- the whole --gpus N flow of this file except the xGMI hop and RCCL.  The line says so "
                    "(config.rehearsal); its numbers are N processes time-sharing one GPU, NOT a scaling measurement")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="--gpus 1 only: run the N > 1 code path (process group, RCCL communicator in the engine, c3's row blocks, "
                    "one-GPU reference) at world size 1 -- a rehearsal of what the driver launches on a multi-GPU node")
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL's version banner at
    # communicator creation, for one) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch  # before salamander_amd: the engine must bind torch's HIP runtime and RCCL (salamander_amd/_lib.py)

    import salamander_amd as sal
    from salamander_amd.distributed import shard_bounds
    from salamander_amd.synthetic import synthetic_problem

    dist = None
    sharded = world > 1 or args.rehearse_sharded
    if sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if args.rehearse_one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # the HIP device of this rank's engine, and where the control plane's small tensors live (gloo reduces host tensors)
    device = 0 if args.rehearse_one_device else local_rank
    ctrl = "cpu" if args.rehearse_one_device else f"cuda:{device}"
    if args.rehearse_one_device:
        args.simulate_no_rccl = True

    strong = sharded and not args.weak
    if not sharded:
        n_total = args.samples_total or N_C2
        X, W0, H0 = synthetic_problem(V, n_total, K, seed=0)
        n_local = n_total
    elif strong:
        n_total = args.samples_total or N_C3
        lo, hi = shard_bounds(n_total, world, rank)
        X, H0 = problem_rows(lo, hi)
        _, W0, _ = synthetic_problem(V, 16, K, seed=0)  # any common start: rank 0's is broadcast below
        n_local = hi - lo
    else:
        n_local = N_C2
        n_total = n_local * world
        X, W0, H0 = synthetic_problem(V, n_local, K, seed=rank)

    engine = sal.Engine(n_local, V, K, device=device)
    rccl_ok = True
    if sharded:
        from salamander_amd.distributed import attach_communicator, broadcast_from_rank0

        # the engine's own RCCL communicator; if it cannot be created (on any rank) the run goes on with the peer-to-peer
        # exchange alone, cross-checked against torch.distributed instead
        err = None
        try:
            if args.simulate_no_rccl:
                raise RuntimeError("simulated: no RCCL communicator (--simulate-no-rccl)")
            attach_communicator(engine)
        except RuntimeError as exc:
            err = str(exc)
        flags = [None] * world
        dist.all_gather_object(flags, err)
        rccl_ok = all(f is None for f in flags)
        if not rccl_ok and rank == 0:
            print(f"[bench] no RCCL communicator in the engine ({next(f for f in flags if f)}); peer-to-peer exchange only", file=sys.stderr)
        W0 = broadcast_from_rank0(W0)
    engine.upload_X(X)
    engine.upload_W(W0)
    engine.upload_H(H0)
    # N > 1: the K x V all-reduce of every step goes either through RCCL or through the engine's peer-to-peer exchange
    # (salnmf_p2p_kernels.h).  Both are run; the exchange's result is checked against the RCCL result before it is timed,
    # and the line reports the faster VALID mode as `value` with both timings under config.exchange.
    exchange = {"p2p_connected": False, "rccl_communicator": rccl_ok} if sharded else None
    if sharded and not (args.no_p2p and rccl_ok):
        from salamander_amd.distributed import attach_peer_exchange

        exchange["p2p_connected"] = bool(attach_peer_exchange(engine, n_total, required=False))
        if exchange["p2p_connected"]:
            exchange.update(validate_exchange(engine, dist, W0, H0, ctrl, reference="rccl" if rccl_ok else "host"))
            if not exchange["p2p_valid"] and rccl_ok:
                engine.set_p2p(False)
            engine.upload_W(W0)
            engine.upload_H(H0)
        if not rccl_ok and not (exchange["p2p_connected"] and exchange.get("p2p_valid")):
            raise SystemExit("bench: neither the engine's RCCL communicator nor a valid peer-to-peer exchange is available on this node")
        if not rccl_ok:
            engine.set_p2p(True)

    failures = []  # engine errors inside timed blocks (an exchange that gave up waiting for a peer): see timed_block

    def barrier():
        try:
            engine.sync()
        except RuntimeError as exc:  # (reported through the block's time, below: every rank still joins the collective)
            failures.append(str(exc))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=ctrl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(obj):
        """Every rank's object, in rank order (a list of one without a process group)."""
        if dist is None or world == 1:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    # untimed: warm-up steps, and one profiling pass so that every event / lazily sized buffer exists
    engine.kl_step(args.warmup)
    engine.profile_kl_steps(8, 0, 2)

    def local_sync():
        engine.sync()
        torch.cuda.synchronize()

    def timed_block():
        # barrier + synchronisation on both sides of EXACTLY K steps.  Every rank stops its clock when ITS K steps have
        # finished (each step's exchange waits for every peer's contribution to that step, so no rank can finish the
        # last step before all ranks have computed theirs) and the block's time is the MAXIMUM over the ranks: the
        # moment the whole job is done.  The closing barrier still runs, but its own latency (a collective launch of
        # 30-50 us: 2 us per step at K = 20) is a cost of measuring, not of the K steps, and is not counted.
        # An engine error inside the block (a peer-to-peer exchange that gave up) does not leave this rank's collectives
        # unmatched: the block's time becomes infinite, which the MAX over the ranks hands to everybody, and every rank
        # leaves the series of blocks at the same point (the caller falls back to RCCL).
        barrier()
        t0 = time.perf_counter()
        try:
            engine.kl_step(args.steps)
            local_sync()
            dt = time.perf_counter() - t0
        except RuntimeError as exc:
            failures.append(str(exc))
            dt = float("inf")
        barrier()
        return max_over_ranks(dt)

    def timed_blocks():
        first = timed_block()
        if math.isinf(first):
            return first, 0, [], first
        n_blocks = int(min(1000, max(5, math.ceil(args.busy_seconds / max(first, 1e-6)))))
        blocks = []
        for _ in range(n_blocks):
            blocks.append(timed_block())
            if math.isinf(blocks[-1]):
                return first, len(blocks), blocks, float("inf")
        return first, n_blocks, blocks, statistics.median(blocks)

    gpu_sections = {}  # wall-clock seconds of the sections in which the GPU is the one that works (launch gaps included)
    t_sec = time.perf_counter()
    first, n_blocks, blocks, median = timed_blocks()
    if math.isinf(median) and exchange is not None and exchange.get("p2p_valid") and rccl_ok:
        # the peer-to-peer exchange failed under load although it had passed its validation: every rank saw the infinite
        # block (MAX over ranks), so every rank comes here; the run goes on through RCCL and the line says so
        exchange["p2p_failed_during_timing"] = failures[-1] if failures else "on another rank"
        exchange["p2p_valid"] = False
        engine.set_p2p(False)
        engine.upload_W(W0)
        engine.upload_H(H0)
        engine.kl_step(args.warmup)
        first, n_blocks, blocks, median = timed_blocks()
    if math.isinf(median):
        raise SystemExit(f"bench: the timed steps failed: {failures[-1] if failures else 'on another rank'}")
    gpu_sections["timed_blocks"] = time.perf_counter() - t_sec
    exchange_mode = "rccl" if sharded else None
    if exchange is not None and exchange.get("p2p_valid") and not rccl_ok:
        exchange["p2p_ms_per_step"] = median / args.steps * 1e3
        exchange_mode = "p2p"
    elif exchange is not None and exchange.get("p2p_valid"):
        # the blocks above ran with the peer-to-peer exchange; the same protocol once more through RCCL
        exchange["p2p_ms_per_step"] = median / args.steps * 1e3
        engine.set_p2p(False)
        engine.kl_step(args.warmup)
        r_first, r_n, r_blocks, r_median = timed_blocks()
        exchange["rccl_ms_per_step"] = None if math.isinf(r_median) else r_median / args.steps * 1e3
        if math.isinf(r_median):  # (RCCL itself failed: the peer-to-peer figure stands; the engine's state is restored)
            exchange["rccl_failed_during_timing"] = failures[-1] if failures else "on another rank"
            engine.set_p2p(True)
            engine.upload_W(W0)
            engine.upload_H(H0)
            engine.kl_step(args.warmup)
            exchange_mode = "p2p"
        elif r_median < median:
            first, n_blocks, blocks, median = r_first, r_n, r_blocks, r_median
        else:
            engine.set_p2p(True)
            exchange_mode = "p2p"
    if exchange is not None:
        exchange["used_for_value"] = exchange_mode
        # where a sharded step's microseconds go, rank by rank (untimed; same launches, HIP events on the two dispatches
        # and s_memrealtime stamps around the exchange inside the tail): printed with the line, so that the first run on a
        # node says whether compute, the exchange's wire time or the ranks' skew (wait_for_peer_flags) is the long pole
        if exchange.get("p2p_valid"):
            was_p2p = exchange_mode == "p2p"
            # (every collective below is entered by every rank whatever happened on it: a rank's failure is a row of the
            # table, not a missing participant -- the timeline must never cost the headline line, nor hang it)
            err = None
            try:
                if not was_p2p:
                    engine.set_p2p(True)
            except Exception as exc:
                err = f"{type(exc).__name__}: {exc}"
            errs = gather(err)
            if all(x is None for x in errs):
                barrier()
                try:
                    mine = engine.profile_sharded_steps(args.steps)
                except Exception as exc:
                    mine = {"error": f"{type(exc).__name__}: {exc}"}
                mine["rank"] = rank
                exchange["timeline_us_per_rank"] = gather(mine)
                exchange["timeline_protocol"] = (
                    f"{args.steps} steps per rank through the peer-to-peer tail; step / fused_pass / tail_exchange_launch from HIP events "
                    "on the dispatches, the rest averaged over the tail's row workgroups from in-kernel 100 MHz stamps"
                )
            else:
                exchange["timeline_error"] = errs
            if not was_p2p:
                try:
                    engine.set_p2p(False)
                except Exception as exc:
                    exchange["timeline_error"] = f"{type(exc).__name__}: {exc}"
            barrier()
        # what the exchange layers themselves say about this job: the RCCL communicator's own rank count (ncclCommCount, not the
        # number this script passed), the peer-to-peer exchange's mapped inboxes, and each rank's device / PCI bus id
        try:
            seen = engine.comm_observed()
        except Exception as exc:
            seen = {"error": f"{type(exc).__name__}: {exc}"}
        seen["rank"], seen["local_rank"], seen["pid"] = rank, local_rank, os.getpid()
        rows = gather(seen)
        exchange["ranks"] = rows
        exchange["rccl_nranks"] = min((r.get("rccl_nranks", -1) for r in rows), default=-1)
        exchange["p2p_nranks"] = min((r.get("p2p_nranks", 0) for r in rows), default=0)
        exchange["distinct_devices"] = len({r.get("pci_bus_id") for r in rows if r.get("pci_bus_id")})

    # untimed: kernel durations from HIP events bound to the dispatches of every 2nd step of 200 (100 samples each)
    barrier()
    t_sec = time.perf_counter()
    _, fused_ms, tail_ms = engine.profile_kl_steps(200, 0, 2)
    objective = engine.objective()
    fwd_ms = engine.profile_objective(20)
    wh_ms = engine.profile_reconstruct(20)
    gpu_sections["kernel_samples"] = time.perf_counter() - t_sec
    barrier()

    # From here on nothing may cost the headline line (the timed blocks above are the measurement): what only rank 0 does is
    # wrapped on rank 0, and what all ranks do together is wrapped with the collectives OUTSIDE the guarded part, so that a
    # rank that failed still meets the others.
    one_gpu = None
    whole = []  # rank 0: the whole problem, drawn once for the one-GPU reference and the CPU baseline (36 s of NumPy at 10^6)

    def whole_problem():
        if not whole:
            whole.extend(problem_rows(0, n_total))
        return whole[0], whole[1]

    if strong and not args.no_one_gpu_reference:
        # the same problem on ONE GPU (rank 0's), the figure the N-GPU value is to be compared with
        if rank == 0:
            try:
                Xa, Ha = whole_problem()
                e1 = sal.Engine(n_total, V, K, device=device)
                e1.upload_X(Xa), e1.upload_W(W0), e1.upload_H(Ha)
                del Xa, Ha
                # the protocol of the timed blocks above: the same warm-up, blocks of the same K steps timed the same way, as
                # many of them as fill the same busy time (a handful of blocks right after the upload ran 9 % slow at the
                # 125 000-sample shard: the clocks had not settled, and the comparison flattered the sharded run)
                e1.kl_step(max(args.warmup, 5))
                e1.sync()
                ts = []
                busy = 0.0
                while len(ts) < 5 or (busy < args.busy_seconds and len(ts) < 1000):
                    t0 = time.perf_counter()
                    e1.kl_step(args.steps)
                    e1.sync()
                    ts.append((time.perf_counter() - t0) / args.steps)
                    busy += ts[-1] * args.steps
                e1.close()
                t1 = statistics.median(ts)
                one_gpu = {
                    "n_gpus": 1,
                    "n_samples": n_total,
                    "ms_per_step": t1 * 1e3,
                    "steps_per_s": 1.0 / t1,
                    "speedup_of_this_run": (args.steps / median) * t1,
                    "how": f"rank 0 alone, median of {len(ts)} blocks of {args.steps} steps after {max(args.warmup, 5)} warm-up steps (the timed blocks' protocol), "
                           f"same {n_total}-sample problem"
                    + (" (the other ranks' engines idle on the SAME device: rehearsal)" if args.rehearse_one_device else ""),
                }
            except Exception as exc:
                one_gpu = {"error": f"{type(exc).__name__}: {exc}"}
        barrier()

    # N > 1 (and the rehearsal): the other half of the metric for the sharded problem.  Rank 0 times the NumPy oracle on the
    # WHOLE problem (a few steps: one step of 96 x 10^6 takes seconds) and broadcasts the KL it reached; then all ranks
    # run the sharded device loop to that target together (the objectives are all-reduced: every rank takes the same
    # decision), and once more for exactly the CPU path's number of steps: the north star's parity gate on the sharded run
    # (W, replicated, on every rank; H on rank 0's rows).
    sharded_cpu, sharded_ttk = None, None
    if strong and not args.no_cpu_baseline:
        box = [None]
        W_cpu = H_cpu = None
        if rank == 0:
            try:
                Xa, Ha = whole_problem()
                rec, (n_cpu, target, cpu_s, W_cpu, H_cpu) = cpu_baseline(Xa, W0, Ha, args.cpu_steps_sharded, args.cpu_budget)
                del Xa, Ha
                whole.clear()
                H_cpu = H_cpu[:n_local].copy()
                box = [(rec, n_cpu, target, cpu_s)]
            except Exception as exc:
                box = [{"error": f"{type(exc).__name__}: {exc}"}]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], dict):
            sharded_cpu = box[0]
        else:
            sharded_cpu, n_cpu, target, cpu_s = box[0]
            steps = obj = None
            loop_s, err, gate = float("nan"), None, None
            barrier()
            try:
                engine.upload_W(W0)
                engine.upload_H(H0)
                steps, obj, loop_s = device_loop_to_target(engine, target, n_cpu + 100)
                engine.upload_W(W0)
                engine.upload_H(H0)
                engine.kl_step(n_cpu)
                Wg = engine.download_W()
                digest = float(np.frombuffer(Wg.tobytes(), dtype=np.uint32).astype(np.uint64).sum() % (1 << 52))
                gate = {"W_digest": digest}
                if rank == 0:
                    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
                    gate.update({"rel_l2_W": rel(Wg, W_cpu), "rel_l2_H_rank0_rows": rel(engine.download_H(), H_cpu)})
            except Exception as exc:
                err = f"{type(exc).__name__}: {exc}"
            loop_s = max_over_ranks(loop_s)
            rows = gather((err, gate))
            errs = [r[0] for r in rows if r[0] is not None]
            if errs or steps is None:
                sharded_ttk = {"error": errs or ["failed on another rank"]}
            else:
                same_W = len({r[1]["W_digest"] for r in rows}) == 1
                g0 = rows[0][1]
                sharded_ttk = {
                    "target": f"KL the NumPy oracle reaches after {n_cpu} update_WH steps on the whole {V}x{n_total} problem from the shared init",
                    "cpu_steps": n_cpu,
                    "target_kl": target,
                    "cpu_seconds": cpu_s,
                    "gpu_steps_to_target": steps,
                    "gpu_objective_there": obj,
                    "reached": bool(obj <= target * (1 + 1e-12)),
                    "gpu_loop_seconds": loop_s,
                    "gpu_loop_protocol": "sharded device loop on all ranks, objective (all-reduced) every 10 steps; maximum over ranks",
                    "parity_steps": n_cpu,
                    "rel_l2_W": g0["rel_l2_W"],
                    "rel_l2_H_rank0_rows": g0["rel_l2_H_rank0_rows"],
                    "W_identical_on_all_ranks": same_W,
                    "parity_gate": 1e-4,
                    "parity_ok": bool(max(g0["rel_l2_W"], g0["rel_l2_H_rank0_rows"]) <= 1e-4 and same_W),
                }
        barrier()

    if rank == 0:
        flops_step = 6.0 * V * K * n_local  # algorithmic flops of one launch of the fused kernel
        achieved = flops_step / (fused_ms * 1e-3) / 1e12
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("fused_kernel_hbm_bytes_per_launch") if n_local == N_C2 else None  # measured at c2 only
                traffic_source = (
                    f"NOT measured in this run: profiles/pmc_summary.json ({rec.get('date', 'round 1')}), rocprofv3 --pmc "
                    "FETCH_SIZE / WRITE_SIZE passes of this command at c2 (FETCH x2 gfx950 correction applied)"
                )
            except Exception:
                traffic = None
        steps_per_s = args.steps / median
        line = {
            "metric": "KL-NMF update-steps/sec (96xN, k=50)",
            "value": steps_per_s * (world if (sharded and not strong) else 1),
            "unit": "update-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": median / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": ("strong" if strong else "weak") if sharded else "n/a",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (
                    f"c3: KLNMF n_signatures={K} on synthetic Poisson counts {V}x{n_total} in total, sample-sharded over {world} GPUs "
                    f"({n_local} rows on rank 0), joint update_WH steps, device resident"
                    if strong
                    else f"c2: KLNMF n_signatures={K} on synthetic Poisson counts {V}x{n_local} per GPU (joint update_WH steps, device resident)"
                ),
                "n_features": V,
                "n_samples_per_gpu": n_local,
                "n_samples_total": n_total,
                "n_signatures": K,
                "parallelism": (
                    f"sample-sharded x{world}; one all-reduce of {K}x{V} f64 per step "
                    + ("by peer-to-peer stores over xGMI (salnmf_p2p_kernels.h)" if exchange_mode == "p2p" else "by RCCL")
                    if sharded
                    else "single GPU"
                ),
                "exchange": exchange,
                "global_steps_per_s": steps_per_s,
                "objective_after_run": objective,
                "one_gpu_same_problem": one_gpu,
            },
            "timing": {
                "blocks": n_blocks,
                "steps_per_block": args.steps,
                "block_ms_median": median * 1e3,
                "block_ms_min": min(blocks) * 1e3,
                "block_ms_max": max(blocks) * 1e3,
                "first_block_ms": first * 1e3,
                "ms_per_step_best_block": min(blocks) / args.steps * 1e3,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "fused_kernel<13,3,2,G,U> (P=H.W, R=X/P, G+=H^T.R, U=R.W^T, H update)",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_flops_per_launch": flops_step,
                "kernel_avg_ms": fused_ms,
                "kernel_samples": 100,
                "tail_avg_ms": tail_ms,
                "step_frac_of_peak": flops_step / (median / args.steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "forward_objective_kernel_ms": fwd_ms,
                "forward_WH_frac": 2.0 * V * K * n_local / (fwd_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "WH_only_kernel_ms": wh_ms,
                "WH_only_frac": 2.0 * V * K * n_local / (wh_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            },
        }
        engine.close()
        if sharded_cpu is not None:
            line["cpu_baseline"] = sharded_cpu
            line["time_to_kl"] = sharded_ttk
            if sharded_ttk and "rel_l2_W" in sharded_ttk:
                line["parity"] = {k: sharded_ttk[k] for k in ("parity_steps", "rel_l2_W", "rel_l2_H_rank0_rows", "W_identical_on_all_ranks", "parity_gate", "parity_ok")}
        if one_gpu and "steps_per_s" in one_gpu:
            # strong scaling: the N-GPU value is to be compared with ONE GPU on the SAME 10^6-sample problem (measured in this very
            # run, on rank 0's GPU) -- not with the N = 1 line of this script, which is c2, a ten times smaller problem
            line["one_gpu_same_problem_value"] = one_gpu["steps_per_s"]
            line["speedup_vs_one_gpu_same_problem"] = one_gpu["speedup_of_this_run"]
        if args.rehearse_one_device:
            line["config"]["rehearsal"] = (
                f"{world} ranks as processes on ONE GPU (device 0), control plane gloo, no RCCL, peer-to-peer exchange between the "
                "processes: a rehearsal of the --gpus N flow, NOT a scaling measurement"
            )
        if not sharded:
            # the other single-GPU configurations FIRST, the CPU baseline (100 s of NumPy on the host cores, the GPU idle)
            # last: a utilisation trace of this process shows the device at work for the first part of the run, then nothing
            if not args.no_extra:
                extra = {}
                for name, fn in (("c4_mvnmf", extra_c4), ("c2_default_init_fit", extra_default_init_fit), ("c2_fp32_fast_mode", extra_fast_mode), ("c2_weighted_step", extra_weighted_step), ("c5_mmcorrnmf", extra_c5), ("wide_catalogues", extra_wide_catalogue)):
                    t_sec = time.perf_counter()
                    try:
                        extra[name] = fn(sal, local_rank)
                    except Exception as exc:  # an extra must never cost the headline line
                        extra[name] = {"error": f"{type(exc).__name__}: {exc}"}
                    gpu_sections["extra." + name] = time.perf_counter() - t_sec  # (includes the host's synthetic data and uploads)
                line["extra"] = extra
            if not args.no_cpu_baseline:
                t_cpu = time.perf_counter()
                rec, (n_cpu, target, cpu_s, W_cpu, H_cpu) = cpu_baseline(X, W0, H0, args.cpu_steps, args.cpu_budget)
                line["cpu_baseline"] = rec
                cpu_wall = time.perf_counter() - t_cpu
                t_sec = time.perf_counter()
                line["time_to_kl"] = time_to_kl(sal, X, W0, H0, n_cpu, target, cpu_s, local_rank, W_cpu, H_cpu)
                # the north star's gate, also at the top level of the line: W, H after the SAME number of steps as the CPU path
                line["parity"] = {k: line["time_to_kl"][k] for k in ("parity_steps", "rel_l2_W", "rel_l2_H", "parity_gate", "parity_ok")}
                gpu_sections["time_to_kl"] = time.perf_counter() - t_sec
                line["cpu_baseline_wall_seconds"] = cpu_wall
        # where this process's wall clock went: the GPU works in the sections listed (the timed blocks are pure device
        # time; the extras include host-side data generation), it idles during the CPU baseline
        line["gpu_seconds_total"] = sum(gpu_sections.values())
        line["gpu_sections_seconds"] = gpu_sections
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    else:
        engine.close()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
