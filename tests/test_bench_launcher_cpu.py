"""`python bench.py --gpus N` with no launcher around it must START N ranks itself (it used to time one GPU and print
"n_gpus": 1).  Exercised here without GPUs: the device count is faked (VMV_BENCH_FAKE_DEVICES), the process group is gloo
(VMV_BENCH_PG_BACKEND) and --pg-dry-run stops every rank after the group has counted its members — the reference starts its own
workers the same way (mp.spawn, inference_text2video_entrance.py:55-61)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("n", [2, 4])
def test_bench_gpus_n_spawns_n_ranks_and_counts_them(n):
    r = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--pg-dry-run"],
             dict(VMV_BENCH_FAKE_DEVICES=str(n), VMV_BENCH_PG_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["rccl_ranks"] == n and j["launcher"] == "self-spawned" and j["steps"] == 3 and j["warmup"] == 1
    # the frame-parallel leg's CHILD processes (one per rank, a process group of their own on a port the parents agreed on) met too
    c = j["fp_child"]
    assert c and "error" not in c and c["n_gpus"] == n and c["rccl_ranks"] == n and c["launcher"] == "frame-parallel child of self-spawned"


def test_frame_parallel_children_rendezvous_under_torch_distributed_run():
    """The driver's own form — `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` — with the frame-parallel leg in
    child processes: the children must not inherit the elastic agent's store (TORCHELASTIC_USE_AGENT_STORE points at the PARENTS' port;
    a child that kept it would wait for ever on a port nobody serves).  --pg-dry-run, gloo: no GPU needed."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VMV_BENCH_FAKE_DEVICES="2", VMV_BENCH_PG_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pg-dry-run"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["launcher"] == "torch.distributed.run"
    c = j["fp_child"]
    assert c and "error" not in c and c["rccl_ranks"] == 2 and c["launcher"] == "frame-parallel child of torch.distributed.run"


def test_no_frame_parallel_spawns_no_children():
    r = _run(["--gpus", "2", "--pg-dry-run", "--no-frame-parallel"], dict(VMV_BENCH_FAKE_DEVICES="2", VMV_BENCH_PG_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["fp_child"] is None and j["rccl_ranks"] == 2


def test_bench_gpus_n_refuses_when_fewer_devices_are_visible():
    r = _run(["--gpus", "8", "--pg-dry-run"], dict(VMV_BENCH_FAKE_DEVICES="2", VMV_BENCH_PG_BACKEND="gloo"))
    assert r.returncode != 0 and "only 2 GPU(s)" in (r.stderr + r.stdout)


def test_bench_main_in_process_reaches_the_process_group(monkeypatch, capsys):
    """`bench.main(["--gpus", "2", ...])` called as a function: the launcher path is taken from the argument list, not from sys.argv."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("VMV_BENCH_FAKE_DEVICES", "2")
    monkeypatch.setenv("VMV_BENCH_PG_BACKEND", "gloo")
    seen = {}
    import subprocess as sp
    orig = sp.Popen

    def spy(cmd, **kw):
        seen.setdefault("ranks", []).append((kw["env"]["RANK"], kw["env"]["WORLD_SIZE"], kw["env"]["MASTER_ADDR"]))
        kw["stdout"] = sp.DEVNULL if kw["env"]["RANK"] != "0" else sp.PIPE
        p = orig(cmd, **kw)
        if kw["env"]["RANK"] == "0":
            seen["p0"] = p
        return p
    monkeypatch.setattr(sp, "Popen", spy)
    bench.main(["--gpus", "2", "--pg-dry-run"])
    assert seen["ranks"] == [("0", "2", "127.0.0.1"), ("1", "2", "127.0.0.1")]
    j = json.loads(seen["p0"].stdout.read().decode().strip().splitlines()[-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2


def test_failing_ranks_fail_the_job():
    """Ranks that die (here: an unknown process-group backend, so every rank raises in init_process_group) give a non-zero exit code and
    no JSON line — never a silent 1-GPU result."""
    r = _run(["--gpus", "2", "--pg-dry-run"], dict(VMV_BENCH_FAKE_DEVICES="2", VMV_BENCH_PG_BACKEND="no_such_backend"), timeout=120)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
