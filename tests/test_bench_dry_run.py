"""bench.py's N > 1 path end to end on CPU: `python bench.py --gpus 2 --dry-run-cpu` goes through its own respawn()
(torch.distributed.run on 127.0.0.1, one rank per "GPU"), the per-rank set-up, warm-up, the barrier-bracketed timed region, the
max-over-ranks reduction and the single JSON line of rank 0 -- with gloo instead of RCCL, the CPU-emulated kernel library instead of
libdgs_hip.so and a tiny model, so that argument plumbing and the collectives of the training step are exercised before the
driver's first multi-GPU run.  Nothing here is a measurement (the line says so: `dry_run`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, gpus=2):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run-cpu", "--steps", "1", "--warmup", "1", *flags],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("exchange", ["fp32", "bf16"])
def test_two_rank_training_step_through_respawn(exchange):
    d = _run("--mode", "train", "--train-batch", "1", "--bucket-mb", "1", "--grad-exchange", exchange)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "dry_run" in d and d["unit"] == "samples/s" and d["value"] > 0
    ts = d["train_step"]
    assert abs(d["value"] - 2 * ts["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 0.006                # whole-job aggregate (2 decimals)
    ar = ts["allreduce"]
    assert ar["world"] == 2 and ar["exchange"] == exchange and ar["buckets"] == len(ar["bucket_mib"]) >= 3
    assert ar["launched_during_backward"] >= ar["buckets"] - 2     # all but the tail overlap the backward
    assert ts["loss"] == ts["loss"]                                # finite


def test_two_rank_inference_step_through_respawn():
    d = _run("--mode", "infer")
    assert d["n_gpus"] == 2 and d["unit"] == "renders/s" and "dry_run" in d and d["value"] > 0
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) <= 0.006                                  # B = 1, 4 views, 2 ranks
    assert d["config"]["parallelism"] == "dp2" and d["cpu_baseline"] is None


def test_one_rank_with_the_process_group_forced():
    """`--force-dist` (the one-GPU RCCL smoke run of profiles/r03_train_rccl_world1.json) on gloo: a process group of one rank, every
    barrier / max-over-ranks reduction / gradient bucket issued."""
    d = _run("--mode", "train", "--train-batch", "1", "--bucket-mb", "1", "--force-dist", gpus=1)
    ar = d["train_step"]["allreduce"]
    assert d["n_gpus"] == 1 and ar["world"] == 1 and ar["collectives_issued"] is True and ar["buckets"] >= 3
    assert d["train_step"]["optimizer"] == "fused"
