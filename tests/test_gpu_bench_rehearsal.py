"""The whole ``bench.py --gpus N`` flow, rehearsed with N = 2 and 3 ranks on ONE GPU.

The driver launches ``bench.py --gpus N`` on an 8-GPU node at round end; nothing but the engine-level protocol
(``tests/test_gpu_p2p.py``) had run at N > 1 before.  ``--rehearse-one-device`` runs the same file, launched the same
way (``python -m torch.distributed.run --nproc-per-node N ...``), with every rank's engine on device 0, the control plane
over gloo and the engine's RCCL communicator treated as absent (two RCCL ranks cannot share a GPU): the row blocks of the
strong-scaling problem, the exchange's validation against the host collective, the timed blocks with their barriers and
the maximum over ranks, the per-rank timeline, rank 0's one-GPU reference of the same problem, the CPU baseline of the
sharded problem, the sharded time-to-KL loop and parity gate, and the one JSON line.  What it cannot cover: the xGMI hop
and RCCL with more than one rank.  The single-rank run below goes through the real RCCL communicator and checks that
the line's rank count is the one RCCL reports (``ncclCommCount``), not the one the script passed.
"""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_distributed_gloo import _free_port

pytestmark = pytest.mark.gpu


def _run(cmd, timeout=420):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]  # ONE JSON line on stdout and nothing else
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharded_flow_with_ranks_as_processes_on_one_gpu(world):
    n_total = 100000
    line = _run([
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
        "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--steps", "20", "--warmup", "5",
        "--rehearse-one-device", "--samples-total", str(n_total), "--busy-seconds", "0.3", "--cpu-steps-sharded", "10",
    ])
    assert line["n_gpus"] == world and line["steps"] == 20 and line["warmup"] == 5 and line["scaling"] == "strong"
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] * 1e-3 - 1.0) < 1e-9
    cfg = line["config"]
    assert "rehearsal" in cfg and cfg["n_samples_total"] == n_total and cfg["workload"].startswith("c3")
    ex = cfg["exchange"]
    assert ex["p2p_connected"] and ex["p2p_valid"] and ex["p2p_W_identical_on_all_ranks"] and ex["validated_against"] == "host"
    assert ex["p2p_vs_rccl_rel_l2_W"] < 1e-11 and ex["used_for_value"] == "p2p"
    # the rank count as the exchange layers report it, one row per rank, every rank with its device
    assert ex["p2p_nranks"] == world and ex["rccl_nranks"] == -1 and not ex["rccl_communicator"]
    assert [r["rank"] for r in ex["ranks"]] == list(range(world))
    assert all(r["p2p_inboxes_mapped"] == world and r["device"] == 0 and r["pci_bus_id"] for r in ex["ranks"])
    assert len({r["pid"] for r in ex["ranks"]}) == world and ex["distinct_devices"] == 1
    # per-rank timeline of the sharded step
    rows = ex["timeline_us_per_rank"]
    assert [r["rank"] for r in rows] == list(range(world)) and all("error" not in r and r["step"] > 0 for r in rows)
    # rank 0's run of the whole problem on one GPU, the sharded CPU baseline, time to KL and the parity gate
    one = cfg["one_gpu_same_problem"]
    assert "error" not in one and one["n_samples"] == n_total and one["steps_per_s"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    ttk = line["time_to_kl"]
    assert "error" not in ttk and ttk["reached"] and ttk["cpu_steps"] == 10 and ttk["gpu_steps_to_target"] in (10, 20)
    par = line["parity"]
    assert par["parity_ok"] and par["W_identical_on_all_ranks"] and max(par["rel_l2_W"], par["rel_l2_H_rank0_rows"]) < 1e-9
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["algorithmic_flops_per_launch"] == 6.0 * 96 * 50 * cfg["n_samples_per_gpu"]


def test_bench_sharded_flow_through_rccl_reports_the_communicators_own_rank_count():
    """World size 1 through the N > 1 code path with the REAL RCCL communicator in the engine: ``rccl_nranks`` comes from
    ``ncclCommCount`` (on an N-GPU node it must equal ``n_gpus``)."""
    line = _run([
        sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--rehearse-sharded", "--samples-total", "100000",
        "--busy-seconds", "0.3", "--cpu-steps-sharded", "10",
    ])
    ex = line["config"]["exchange"]
    assert ex["rccl_communicator"] and ex["rccl_nranks"] == 1 == line["n_gpus"] and ex["p2p_nranks"] == 1
    assert ex["ranks"][0]["rccl_rank"] == 0 and ex["ranks"][0]["rccl_device"] == ex["ranks"][0]["device"] == 0
    assert ex["p2p_valid"] and ex["validated_against"] == "rccl" and "rccl_ms_per_step" in ex and "p2p_ms_per_step" in ex
    assert line["parity"]["parity_ok"] and line["time_to_kl"]["reached"]
