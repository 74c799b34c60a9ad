"""RCCL at world size 1 (one GPU box): the collectives of newmap_amd.parallel and of bench.py run on the "nccl" backend in a
fresh child process under the environment torch.distributed.run gives a rank (the 8-GPU job is the driver's to launch; here
every collective of that job executes once, on device tensors, with a world of one)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _launcher_env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("NEWMAP_AMD_DIST_BACKEND", None)
    return env


def _last_json(stdout: str) -> dict:
    return json.loads([ln for ln in stdout.splitlines() if ln.startswith("{")][-1])


def test_parallel_collectives_on_rccl_world_of_one():
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "rccl_child.py")], env=_launcher_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["backend"] == "nccl" and out["world"] == 1 and out["rccl_ranks"] == 1
    assert out["gather_to_root_equal"] and out["raise_together"]
    assert out.get("unverified_records", [0]) == [0] * len(out.get("unverified_records", [0]))
    for tag in ("range", "list"):
        assert out[f"{tag}_files"] == 4 and out[f"{tag}_gather_equal"] and out[f"{tag}_shard_equal"], out


def test_bench_under_the_launcher_with_one_rank(tmp_path):
    """bench.py as ONE rank of torch.distributed.run: init_process_group("nccl"), the `rccl_ranks` all-reduce, the max / sum
    reductions of the timed region and Run.final_gather all execute"""
    env = _launcher_env()
    env["NEWMAP_AMD_BENCH_DIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--mbp", "30", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-configs1", "--no-end-to-end", "--no-spread"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["collective_backend"].startswith("nccl")
    assert out["final_gather_ms"] > 0 and out["final_gather_ranks_with_results"] == 1 and out["value"] > 0
