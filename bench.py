"""Headline benchmark: KL-NMF update-steps/sec at 96 x N, k = 50 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one joint KLNMF update (W and H both updated = one ``update_WH`` call of the
reference, ``_utils_klnmf.py:281-361``) of the device-resident state.  Per GPU the workload
is config c2 of BASELINE.json -- synthetic Poisson counts 96 x 100 000, k = 50, fp64; with
N GPUs the sample axis is sharded (N x 100 000 samples in total, weak scaling) and every
step contains one RCCL all-reduce of the 50 x 96 numerator.  ``value`` = shard-steps all
ranks completed per second = N x (global steps / s); at N = 1 that is plain update-steps/s
on c2.  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel (the fused update
pass) on algorithmic flops 6*V*K*N against the fp64 MFMA peak; its duration is measured
with HIP events on the engine's stream around the launches of every 25th step inside the
timed region (an event record costs a few microseconds of stream time, so denser sampling would
slow the very loop it measures).
``cpu_baseline`` times the NumPy oracle (the restated reference arithmetic) on this box's
host cores, rank 0, N = 1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V, N_PER_GPU, K = 96, 100000, 50
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (spec); 77.8 measured by tools/mfma_f64_probe.hip


def cpu_baseline(X, W0, H0, budget_s=15.0):
    """The oracle's update_WH on the host cores: 3 warm-up steps, then timed steps for ~budget_s."""
    from oracle import klnmf_oracle as orc

    Xt, W, H = np.asfortranarray(X.T), W0.T.copy(), H0.T.copy()  # the layouts the reference computes on
    for _ in range(3):
        W, H = orc.update_WH(Xt, W, H)
    n, t0 = 0, time.perf_counter()
    while True:
        W, H = orc.update_WH(Xt, W, H)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 200:
            break
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    return {
        "value": n / dt,
        "unit": "update-steps/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"{n} timed update_WH steps (after 3 warm-up) of the NumPy oracle on the full 96x{X.shape[0]} k={K} workload, numpy {np.__version__}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--samples-per-gpu", type=int, default=N_PER_GPU)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch

    import salamander_amd as sal
    from salamander_amd.synthetic import synthetic_problem

    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n_local = args.samples_per_gpu
    # every rank's shard comes from its own seed; W0 is rank 0's (broadcast below)
    X, W0, H0 = synthetic_problem(V, n_local, K, seed=rank)
    engine = sal.Engine(n_local, V, K, device=local_rank)
    if world > 1:
        from salamander_amd.distributed import attach_communicator, broadcast_from_rank0

        attach_communicator(engine)
        W0 = broadcast_from_rank0(W0)
    engine.upload_X(X)
    engine.upload_W(W0)
    engine.upload_H(H0)

    def barrier():
        engine.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    engine.kl_step(args.warmup)
    barrier()
    t0 = time.perf_counter()
    total_ms, fused_ms, tail_ms = engine.profile_kl_steps(args.steps, 0, 25)  # HIP events on every 25th step; syncs the stream
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    objective = engine.objective()
    fwd_ms = engine.profile_objective(10)
    wh_ms = engine.profile_reconstruct(10)

    if rank == 0:
        flops_step = 6.0 * V * K * n_local  # algorithmic flops of one launch of the fused kernel
        achieved = flops_step / (fused_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("fused_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "KL-NMF update-steps/sec (96x100000-sample shard per GPU, k=50)",
            "value": world * args.steps / elapsed,
            "unit": "update-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"c2: KLNMF n_signatures={K} on synthetic Poisson counts {V}x{n_local} per GPU (joint update_WH steps, device resident)",
                "n_features": V,
                "n_samples_per_gpu": n_local,
                "n_samples_total": n_local * world,
                "n_signatures": K,
                "parallelism": f"sample-sharded x{world}; one RCCL all-reduce of {K}x{V} f64 per step" if world > 1 else "single GPU",
                "global_steps_per_s": args.steps / elapsed,
                "objective_after_run": objective,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "fused_kernel<13,G,U> (P=H.W, R=X/P, G+=H^T.R, U=R.W^T, H update)",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "algorithmic_flops_per_launch": flops_step,
                "kernel_avg_ms": fused_ms,
                "tail_avg_ms": tail_ms,
                "event_total_ms_per_step": total_ms / args.steps,
                "forward_objective_kernel_ms": fwd_ms,
                "forward_WH_frac": 2.0 * V * K * n_local / (fwd_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "WH_only_kernel_ms": wh_ms,
                "WH_only_frac": 2.0 * V * K * n_local / (wh_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(X, W0, H0)
        print(json.dumps(line), flush=True)

    engine.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
