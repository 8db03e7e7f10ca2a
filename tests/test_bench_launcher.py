"""`python bench.py --gpus N` without a launcher: the parent starts N child processes with
the torch.distributed environment, relays rank 0's JSON line and fails if any rank fails.  The children here are a
stub (ARTEMIS_BENCH_CHILD_CMD) so the launcher logic runs without a GPU; the stub also proves that the ranks can
rendezvous on the address / port the parent hands out (gloo all-reduce)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, os, sys
result_fd = os.dup(1)   # as bench.py does: the caller's stdout is kept for the line, library chatter goes to stderr
os.dup2(2, 1)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert "ARTEMIS_BENCH_CHILD_CMD" not in os.environ
mode = os.environ.get("STUB_MODE", "ok")
if mode == "fail" and rank == world - 1:
    sys.exit(7)
import torch
import torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t)
print("noise from rank %d" % rank, file=sys.stderr)
if rank == 0:
    os.write(result_fd, (json.dumps({"sum": float(t.item()), "argv": sys.argv[1:], "n_gpus": world}) + "\n").encode())
else:
    os.write(result_fd, b"a line on stdout of a rank > 0 that must NOT reach the parent's stdout\n")
dist.destroy_process_group()
'''


def launch(tmp_path, n, mode="ok", extra=()):
    stub = tmp_path / "stub_child.py"
    stub.write_text(STUB)
    env = dict(os.environ, ARTEMIS_BENCH_CHILD_CMD=json.dumps([sys.executable, str(stub)]), STUB_MODE=mode, ARTEMIS_BENCH_GRACE_S="3")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3"] + list(extra),
                          env=env, capture_output=True, text=True, timeout=300)


def test_bench_starts_its_own_ranks(tmp_path):
    r = launch(tmp_path, 2, extra=["--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line: rank 0's
    out = json.loads(lines[0])
    assert out["sum"] == 3.0 and out["n_gpus"] == 2
    assert out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]  # the children get the parent's arguments
    assert "must NOT reach" in r.stderr  # the other ranks' stdout goes to stderr


def test_bench_launcher_four_ranks(tmp_path):
    r = launch(tmp_path, 4)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout)["sum"] == 10.0


def test_bench_launcher_fails_when_a_rank_fails(tmp_path):
    r = launch(tmp_path, 2, mode="fail")
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank(s) failed" in r.stderr


def test_the_refined_workloads_accept_n_ranks(tmp_path):
    """BASELINE configs[3] / [4] are 8-GPU configurations: `--workload disk_sph_smr|disk_amr --gpus N` reaches the ranks
    (missing 1); the one-block workloads refuse N > 1 before anything is started."""
    for w in ("disk_sph_smr", "disk_amr"):
        r = launch(tmp_path, 2, extra=["--workload", w])
        assert r.returncode == 0, r.stderr[-2000:]
        assert json.loads(r.stdout)["argv"] == ["--gpus", "2", "--steps", "3", "--workload", w]
    for w in ("ssheet_dust", "disk_sph"):
        r = launch(tmp_path, 2, extra=["--workload", w])
        assert r.returncode != 0 and "single-GPU measurement" in r.stderr and r.stdout.strip() == ""
    r = launch(tmp_path, 2, extra=["--workload", "disk_amr", "--remesh-in-timed-region"])
    assert r.returncode != 0 and "single-GPU measurement" in r.stderr


def test_bench_rejects_unsupported_rank_counts(tmp_path):
    r = launch(tmp_path, 3)
    assert r.returncode != 0 and "1, 2, 4 or 8" in r.stderr


def test_single_gpu_run_needs_no_launcher_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr and r.stdout.strip() == ""
