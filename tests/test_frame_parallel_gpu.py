"""Frame-parallel sampling on the GPU (`pytest -m gpu`; BASELINE configs[2]).  The GPU box has ONE device,
so the collectives are exercised two ways: (1) world 1 — the sharded plan (pack / all-to-all / unpack, gathered GroupNorm
sums, plan cut at its collectives) must reproduce the unsharded plan bit for bit; (2) two processes sharing cuda:0 with a
gloo group (device tensors staged through host memory by ``comm.FrameComm``) — every rank's frames against the single-rank
HIP result and the fp32 oracle.  RCCL itself is reached through the same two ``torch.distributed`` calls at N > 1.

Tolerances (rel-L2 on eps): vs the oracle <= 3e-2 (bf16 storage; same bound as tests/test_unet_gpu.py); sharded vs
single-rank <= 3e-2 (the two runs' bf16 rounding noise decorrelates once the GroupNorm fold order differs)."""
import dataclasses
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(in_dim=4, dim=64, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64,
           num_res_blocks=1, attn_scales=[1.0, 0.5], camera_dim=16, use_camera_condition=True,
           use_fps_condition=False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rel_l2(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-12))


def _case(F_, H, W, L=7, seed=5):
    from oracle.unet_ref import UNetCfg
    from oracle.weights import random_state_dict, unet_param_shapes
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 99)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 4, F_, H, W, generator=g)
    t = torch.tensor([501])
    y = torch.randn(2, L, 1024, generator=g)
    cam = torch.randn(1, F_, 16, generator=g)
    return ocfg, sd, x, t, y, cam


@pytest.mark.parametrize("same_gn_form", [True, False])
def test_world1_sharded_plan_equals_the_unsharded_plan(monkeypatch, same_gn_form):
    """N-shard result == 1-GPU result.  With BOTH plans on the statistics -> apply form of the GroupNorms
    (VMV_GN_FUSED=0: the one-launch LDS form is a choice of the unsharded plan only) every launch of the sharded plan computes
    what its unsharded counterpart computes — the layout switches are identity permutations at world 1, the gathered totals
    records are the rank's own — and the result must be BITWISE equal.  With the default forms (unsharded all-frame norms in one
    launch, two-pass; sharded stats -> gather -> apply) the statistics differ in the last bits, which flips isolated 16-bit
    roundings: bounded by twice the storage-rounding noise."""
    if same_gn_form:
        monkeypatch.setenv("VMV_GN_FUSED", "0")
    from videomv_amd.comm import FrameComm
    from videomv_amd.unet_engine import UNetEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        ocfg, sd, x, t, y, cam = _case(4, 16, 16)
        dev = torch.device("cuda", 0)
        outs = []
        for comm in (None, FrameComm()):
            eng = UNetEngine(CFG, sd, 2, 4, 16, 16, y.shape[1], dev, n_t=1, comm=comm)
            eng.set_context(y.to(dev)); eng.set_camera(cam.to(dev))
            eng.forward_rows(x.to(dev), t.to(dev))
            outs.append(eng.eps_ncfhw().cpu())
            if comm is not None:
                assert len(eng.breaks) > 30 and comm.n_all_to_all > 0 and comm.n_all_gather > 0
        assert torch.isfinite(outs[0]).all()
        if same_gn_form:
            assert torch.equal(outs[1], outs[0]), rel_l2(outs[1], outs[0])
        else:
            from videomv_amd import _lib as L
            tol = 5e-3 if L.elem_name() == "fp16" else 3e-2
            assert rel_l2(outs[1], outs[0]) < tol, rel_l2(outs[1], outs[0])
    finally:
        dist.destroy_process_group()


def test_world1_kv_gather_temporal_mode_on_the_gpu(monkeypatch):
    """VMV_FP_TEMPORAL=kv_gather (BASELINE's north-star form: an all-gather of K | V before each temporal attention) through the HIP
    kernels at world 1: B = 1 plan, short-sequence attention with Nq = Nk = F here, the K | V pack copy, the gather through the
    communicator — against the unsharded B = 1 plan."""
    monkeypatch.setenv("VMV_FP_TEMPORAL", "kv_gather")
    from videomv_amd.comm import FrameComm
    from videomv_amd.unet_engine import UNetEngine
    from videomv_amd import _lib as L
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        ocfg, sd, x, t, y, cam = _case(4, 16, 16)
        dev = torch.device("cuda", 0)
        outs = []
        for comm in (None, FrameComm()):
            eng = UNetEngine(CFG, sd, 1, 4, 16, 16, y.shape[1], dev, n_t=1, comm=comm)
            eng.set_context(y[:1].to(dev)); eng.set_camera(cam.to(dev))
            eng.forward_rows(x.to(dev), t.to(dev))
            outs.append(eng.eps_ncfhw().cpu())
            if comm is not None:
                assert sum(1 for l in eng.S.labels if l.endswith(".kv.pack")) > 0
        tol = 5e-3 if L.elem_name() == "fp16" else 3e-2
        assert torch.isfinite(outs[0]).all() and rel_l2(outs[1], outs[0]) < tol, rel_l2(outs[1], outs[0])
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, F_, H, W, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.unet_ref import unet_forward
        from videomv_amd.comm import FrameComm
        from videomv_amd.unet_engine import UNetEngine
        ocfg, sd, x, t, y, cam = _case(F_, H, W)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ref = UNetEngine(CFG, sd, 2, F_, H, W, y.shape[1], dev, n_t=1)
        ref.set_context(y.to(dev)); ref.set_camera(cam.to(dev))
        ref.forward_rows(x.to(dev), t.to(dev))
        eps_single = ref.eps_ncfhw().cpu()
        eps_oracle = torch.cat([unet_forward(sd, ocfg, x, t, y[i:i + 1], cam) for i in range(2)], dim=0)
        comm = FrameComm()
        fl = F_ // world
        sl = slice(rank * fl, (rank + 1) * fl)
        eng = UNetEngine(CFG, sd, 2, F_, H, W, y.shape[1], dev, n_t=1, comm=comm)
        eng.set_context(y.to(dev)); eng.set_camera(cam.to(dev))
        eng.forward_rows(x[:, :, sl].contiguous().to(dev), t.to(dev))
        eps_shard = eng.eps_ncfhw().cpu()
        q.put(dict(rank=rank, e_single=rel_l2(eps_shard, eps_single[:, :, sl]), e_oracle=rel_l2(eps_shard, eps_oracle[:, :, sl]),
                   e_single_oracle=rel_l2(eps_single, eps_oracle), a2a=comm.n_all_to_all, ag=comm.n_all_gather))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


@pytest.mark.parametrize("F_,H,W", [(4, 8, 8), (6, 16, 8)])
def test_two_ranks_one_gpu_host_staged_collectives(F_, H, W):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F_, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
    for r in res:
        assert r["a2a"] > 0 and r["ag"] > 0
        assert r["e_oracle"] < 3e-2, r
        assert abs(r["e_oracle"] - r["e_single_oracle"]) < 5e-3, r
        assert r["e_single"] < 3e-2, r


def _rccl_world1_worker(port, q):
    """Child process: an RCCL group of ONE rank (the only RCCL configuration a 1-GPU box can run), VMV_COMM_FORCE=1 so that the
    collectives are issued although world == 1.  The communicator then holds a VmvComm handle (comm.native_comm: unique id from
    rank 0, broadcast over torch.distributed, ncclCommInitRank inside libvmv), the plan carries its collectives as VMV_OP_COMM and
    one C call replays it; the same plan captured into a hipGraph must give the same bits."""
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ["VMV_COMM_FORCE"], os.environ["VMV_GN_FUSED"] = "1", "0"
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from videomv_amd import _lib as L
        from videomv_amd.comm import FrameComm
        from videomv_amd.unet_engine import UNetEngine
        ocfg, sd, x, t, y, cam = _case(4, 16, 16)
        out = {}
        ref = UNetEngine(CFG, sd, 2, 4, 16, 16, y.shape[1], dev, n_t=1)
        ref.set_context(y.to(dev)); ref.set_camera(cam.to(dev)); ref.forward_rows(x.to(dev), t.to(dev))
        eps_ref = ref.eps_ncfhw().cpu()
        comm = FrameComm()
        out["handle"] = bool(comm.handle)
        eng = UNetEngine(CFG, sd, 2, 4, 16, 16, y.shape[1], dev, n_t=1, comm=comm)
        out["breaks"], out["comm_ops"] = len(eng.breaks), eng.n_comm_ops
        eng.set_context(y.to(dev)); eng.set_camera(cam.to(dev)); eng.forward_rows(x.to(dev), t.to(dev))
        torch.cuda.synchronize()
        eps_c = eng.eps_ncfhw().cpu()
        out["bitwise_vs_unsharded"] = bool(torch.equal(eps_c, eps_ref))
        out["err"] = rel_l2(eps_c, eps_ref)
        # the same plan as a hipGraph (RCCL collectives captured with the kernels)
        try:
            out["graph_nodes"] = eng.S.capture_graph()
            eng.eps_rows.zero_()
            eng.forward_rows(x.to(dev), t.to(dev))
            torch.cuda.synchronize()
            out["graph_bitwise"] = bool(torch.equal(eng.eps_ncfhw().cpu(), eps_c))
        except Exception as e:
            out["graph_error"] = f"{type(e).__name__}: {e}"
        q.put(out)
        del eng
        comm.close()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(error=traceback.format_exc()))
    finally:
        q.close()
        q.join_thread()      # (the queue's feeder thread has flushed the result)
        os._exit(0)          # no interpreter teardown of HIP / RCCL state in the throw-away child


def test_rccl_collectives_inside_the_plan_world1():
    """vmv_comm_* (SURVEY §8b last row): the frame-parallel plan with its collectives issued by the C replay loop through RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q))
    p.start()
    try:
        r = q.get(timeout=420)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert "error" not in r, r["error"]
    assert r["handle"] and r["breaks"] == 0 and r["comm_ops"] > 30, r
    assert r["bitwise_vs_unsharded"], r                      # identity permutations + the rank's own totals: same bits (same GroupNorm form)
    assert r.get("graph_bitwise"), r                         # RCCL collectives captured into the hipGraph with the kernels
    print("RCCL world-1 plan:", r)


def test_simulated_rank_and_graph_replay_on_the_gpu(monkeypatch):
    """comm.SimComm on the GPU: rank 0 of an 8-GPU run of a 24-view sample (3 views, HW / 8 pixels) — collectives recorded into the
    plan as device-local copies — replayed eagerly and as a hipGraph: identical bits, finite values; and the plain (unsharded) plan
    as a hipGraph equals its eager replay."""
    from videomv_amd.comm import SimComm
    from videomv_amd.unet_engine import UNetEngine
    ocfg, sd, x, t, y, cam = _case(24, 16, 16)
    dev = torch.device("cuda", 0)
    for comm, xs in ((None, x), (SimComm(8, 0), x[:, :, :3].contiguous())):
        eng = UNetEngine(CFG, sd, 2, 24, 16, 16, y.shape[1], dev, n_t=1, comm=comm)
        eng.set_context(y.to(dev)); eng.set_camera(cam.to(dev))
        eng.forward_rows(xs.to(dev), t.to(dev))
        torch.cuda.synchronize()
        e1 = eng.eps_ncfhw().clone()
        assert torch.isfinite(e1).all()
        if comm is not None:
            assert eng.F == 3 and not eng.breaks and eng.n_comm_ops > 30
        nodes = eng.S.capture_graph()
        assert nodes >= eng.S.nops                     # every launch a node (RCCL-free plans: exactly one kernel node per op)
        eng.eps_rows.zero_()
        eng.forward_rows(xs.to(dev), t.to(dev))
        torch.cuda.synchronize()
        assert torch.equal(eng.eps_ncfhw(), e1)


@pytest.mark.parametrize("n", [2, 4])
def test_bench_gpus_2_end_to_end_on_one_gpu(n):
    """`python bench.py --gpus 2` (and 4: CFG x frame as 2 groups x 2 shards, all-to-all over 4 chunks) with no launcher around it, END TO END on real kernels (round 6): bench.py starts the two
    ranks itself; here both share GPU 0 and the process group is gloo with host-staged collectives (VMV_BENCH_SHARE_GPU=1 — a smoke
    test of the N > 1 flow, never a measurement): the one JSON line says n_gpus = 2, the communicator counted 2 ranks, the replica value
    covers both ranks' steps, and the frame-parallel legs (single plan, branch-pipelined, CFG x frame) ran to finite latents."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VMV_BENCH_PG_BACKEND="gloo", VMV_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--latent", "16x16", "--frames", "4",
                        "--no-cpu-baseline", "--no-op-profile", "--frame-parallel-budget", "600"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["launcher"] == "self-spawned" and d["finite"] and d["scaling"] == "weak"
    assert abs(d["value"] - n * 2 / (d["ms_per_step"] * 2 / 1000.0)) < 1e-2 * d["value"]          # value = N x K steps / max-over-ranks time
    fp = d["frame_parallel"]
    assert fp and "error" not in fp, fp
    assert fp["views_per_gpu"] == 4 // n and fp["single_plan"]["finite"] and fp["branch_pipelined"]["finite"] and fp["cfg_x_frame"]["finite"]
    # (the leg ran in one child process per rank with a process group of its own: a fault there cannot cost the replica line)
    assert fp["child_rccl_ranks"] == n and "child processes" in fp["isolation"]
    # b prompts per plan over the group: every rank its frames of all b samples in one plan
    pp = fp["prompts_per_plan"]
    assert "error" not in pp and pp["2"]["finite"] and pp["4"]["finite"], pp
