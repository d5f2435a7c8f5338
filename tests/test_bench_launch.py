"""`bench.py --gpus N` starts its own ranks when no launcher has set WORLD_SIZE.  CPU part: the refusals.  GPU part: a
two-rank run on one GPU (gloo carries the exchanges; RCCL needs one GPU per rank) whose line reports what the process
group held, not what the flag said."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_more_ranks_than_gpus_is_refused_over_rccl():
    import torch
    have = torch.cuda.device_count()
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 2)], env=_env(ZKHIP_BENCH_BACKEND="nccl"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 2
    assert "GPU(s) are visible" in out.stderr
    assert not out.stdout.strip()


def test_flag_and_launcher_world_size_must_agree():
    import torch
    if torch.cuda.is_available():
        pytest.skip("checked where no GPU is present: the refusal comes before any GPU call either way")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 2
    assert "WORLD_SIZE=2" in out.stderr


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_two_self_launched_ranks_on_one_gpu():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(ZKHIP_BENCH_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=1100)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2
    assert line["launcher"] == "self" and line["backend"] == "gloo"
    assert sorted(r["rank"] for r in line["ranks_seen"]) == [0, 1]
    assert len({r["pid"] for r in line["ranks_seen"]}) == 2
    assert line["msm_base_split_ok"] is True
    split = line["evaluate_h"]["split_by_coset"]
    assert split["ranks"] == 2 and split["ranks_with_rows"] == 2 and split["rows_gathered"] == 1 << 21
    proof = line["create_proof_k19"]
    assert "error" not in proof, proof
    assert proof["ranks"] == 2 and proof["proof_bytes"] > 0
    assert line["value"] > 0 and line["scaling"] == "weak"
