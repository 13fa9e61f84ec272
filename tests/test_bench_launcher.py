"""bench.py as a launcher (missing #1): `python bench.py --gpus N` started plainly must become an N-rank job
(the reference's multi-GPU entry is a launcher too: sh/ddp_train.sh:1, train.py:526).  CPU-only: --dry-run runs the
rendezvous / barrier / max-over-ranks plumbing on gloo without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _line(stdout):
    rows = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 1, stdout            # exactly ONE json line, printed by rank 0
    return json.loads(rows[0])


def test_plain_start_with_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"] is True
    assert line["config"]["global_batch"] == 32 and line["config"]["parallelism"].startswith("dp2")
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 one
    assert line["ms_per_step"] >= 1.9


def test_single_rank_dry_run_is_one_rank():
    r = _run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1


def test_under_torch_distributed_run_the_process_is_a_rank():
    """The driver's form: python -m torch.distributed.run ... bench.py --gpus 2 (no second spawn)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


def test_more_ranks_than_devices_fails_loudly():
    """No GPU in the build container: a real (non-dry) 2-rank run must refuse, not quietly run one rank."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run", "--steps", "1"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29654"}, timeout=60)
    assert r.returncode != 0
