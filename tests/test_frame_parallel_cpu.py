"""Frame-parallel sampling (BASELINE configs[2]) on 2 gloo ranks, CPU: each rank records the sharded plan
(layout-switch all-to-alls, all-gathered GroupNorm partial sums), tests/plan_interp.py executes the recorded launches,
and the result is compared with (i) the single-rank plan and (ii) the fp32 oracle.

Tolerances (rel-L2 on eps): sharded plan vs fp32 oracle <= 2e-2 — the same bound as the single-rank plan (bf16 storage
between ops), and the two must be equally close to the oracle (within 3e-3 of each other's error).  Sharded vs single-rank
is only bounded by 3e-2: the fold order of the GroupNorm partial sums differs, which flips isolated bf16 roundings early
in the net and decorrelates the two runs' rounding noise (measured 1.7e-2 when each is 1.6e-2 from the oracle).  The fused
CFG(9.0)+DDIM update amplifies eps differences, hence 6e-2 there.  A layout / index bug gives O(1) errors."""
import dataclasses
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CFG = dict(in_dim=4, dim=64, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64,
           num_res_blocks=1, attn_scales=[1.0, 0.5], camera_dim=16, use_camera_condition=True,
           use_fps_condition=False)


class _Patch:
    """monkeypatch stand-in for spawned workers (no undo needed: the process exits)."""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp
        plan_interp.install(_Patch)
        from oracle.unet_ref import UNetCfg, unet_forward
        from oracle.weights import random_state_dict, unet_param_shapes
        from videomv_amd.comm import FrameComm
        from videomv_amd.unet_engine import UNetEngine
        from videomv_amd.registry import MODEL, DIFFUSION
        import videomv_amd  # noqa: F401
        ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
        sd = random_state_dict(unet_param_shapes(ocfg), 99)
        B, F_, H, W, L = 2, 4, 8, 8, 5
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1, 4, F_, H, W, generator=g)
        t = torch.tensor([501])
        y = torch.randn(B, L, 1024, generator=g)
        cam = torch.randn(1, F_, 16, generator=g)
        dev = torch.device("cpu")
        # (i) single-rank plan
        ref_eng = UNetEngine(CFG, sd, B, F_, H, W, L, dev, n_t=1)
        ref_eng.set_context(y); ref_eng.set_camera(cam)
        ref_eng.forward_rows(x, t)
        eps_single = ref_eng.eps_ncfhw()
        # (ii) oracle, both branches
        eps_oracle = torch.cat([unet_forward(sd, ocfg, x, t, y[i:i + 1], cam) for i in range(B)], dim=0)
        # frame-parallel plan: this rank's frames
        comm = FrameComm()
        fl = F_ // world
        eng = UNetEngine(CFG, sd, B, F_, H, W, L, dev, n_t=1, comm=comm)
        eng.set_context(y); eng.set_camera(cam)
        eng.forward_rows(x[:, :, rank * fl:(rank + 1) * fl].contiguous(), t)
        eps_shard = eng.eps_ncfhw()
        sl = slice(rank * fl, (rank + 1) * fl)
        out = dict(rank=rank, n_breaks=len(eng.breaks), a2a=comm.n_all_to_all, ag=comm.n_all_gather,
                   e_single=rel_l2(eps_shard, eps_single[:, :, sl]), e_oracle=rel_l2(eps_shard, eps_oracle[:, :, sl]),
                   e_single_oracle=rel_l2(eps_single, eps_oracle))
        # whole-sample module API + one fused CFG/DDIM step on the shard, gathered
        from videomv_amd.unet_t2v import gather_frames
        m = MODEL.build(dict(type="UNetSD_T2VBase", **{k: v for k, v in CFG.items()}))
        m.load_state_dict(sd, strict=False)
        full_single = m(x, t, y=y[:1], camera_data=cam)
        m.set_frame_parallel(comm)
        full_sharded = m(x, t, y=y[:1], camera_data=cam)
        out["e_module"] = rel_l2(full_sharded, full_single)
        diff = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                                    schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                                    mean_type="eps", var_type="fixed_small"))
        kc, ku = dict(y=y[:1], camera_data=cam), dict(y=y[1:], camera_data=cam)
        xt = x[:, :, sl].clone().contiguous()
        diff.ddim_step_hip(xt, 501, m, kc, ku, 9.0, 500)
        x_next = gather_frames(comm, xt)
        # the fused pass ran branch-pipelined: two B = 1 plans, one permute copy per layout switch, shared packed weights
        pe = m._pipe["engs"]
        out["pipe"] = dict(n=len(pe), shared_w=pe[0].w is pe[1].w,
                           copies=[sum(1 for l in e.S.labels if l.startswith("shard.")) for e in pe],
                           switches=[sum(1 for i, _ in e.breaks if True) for e in pe],
                           copies_b2=sum(1 for l in eng.S.labels if l.startswith("shard.")),
                           a2a_b2=sum(1 for l in eng.S.labels if l.endswith(".unpack")))
        m.set_frame_parallel(None)
        xt1 = x.clone()
        diff.ddim_step_hip(xt1, 501, m, kc, ku, 9.0, 500)
        out["e_ddim"] = rel_l2(x_next, xt1)
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:   # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


def _tqa_fp_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ["VMV_TQA_MIN_ITEMS"] = "1"
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp
        plan_interp.install(_Patch)
        from oracle.weights import random_state_dict, unet_param_shapes
        from oracle.unet_ref import UNetCfg
        from videomv_amd import _lib as L
        from videomv_amd.comm import FrameComm
        from videomv_amd.unet_engine import UNetEngine
        cfg = dict(CFG, dim=320, dim_mult=[1], num_heads=5, attn_scales=[1.0])
        ocfg = UNetCfg(**{k: v for k, v in cfg.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
        sd = random_state_dict(unet_param_shapes(ocfg), 31)
        B, F_, H, W, Lc = 2, 4, 4, 4, 5
        g = torch.Generator().manual_seed(8)
        x = torch.randn(1, 4, F_, H, W, generator=g)
        t = torch.tensor([501])
        y = torch.randn(B, Lc, 1024, generator=g)
        cam = torch.randn(1, F_, 16, generator=g)
        dev = torch.device("cpu")
        ref = UNetEngine(cfg, sd, B, F_, H, W, Lc, dev, n_t=1)
        ref.set_context(y); ref.set_camera(cam); ref.forward_rows(x, t)
        comm = FrameComm()
        fl = F_ // world
        eng = UNetEngine(cfg, sd, B, F_, H, W, Lc, dev, n_t=1, comm=comm)
        eng.set_context(y); eng.set_camera(cam)
        eng.forward_rows(x[:, :, rank * fl:(rank + 1) * fl].contiguous(), t)
        fused = [p_ for op, p_ in eng.S.recorded if op == L.OP_GEMM and p_.epilogue == L.EPI_TATTN]
        q.put(dict(rank=rank, n_fused=len(fused), geom=sorted({(p_.F, p_.P, p_.M) for p_ in fused}),
                   n_fused_ref=sum(1 for op, p_ in ref.S.recorded if op == L.OP_GEMM and p_.epilogue == L.EPI_TATTN),
                   e_single=rel_l2(eng.eps_ncfhw(), ref.eps_ncfhw()[:, :, rank * fl:(rank + 1) * fl])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


def test_fused_qkv_temporal_attention_on_the_pixel_major_shard_two_ranks():
    """Round 6: under frame-parallel the TemporalTransformer runs on the pixel-major shard (all F frames of H W / R pixels), so the fused
    q | k | v + temporal-attention launch (VMV_EPI_TATTN) is recorded there with P = H W / R and the GLOBAL frame count; a K = 320
    one-level network on 2 gloo ranks against the single-rank plan (which records the same fused launches with P = H W)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tqa_fp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
        assert r["n_fused"] == r["n_fused_ref"] == 10 and r["geom"] == [(4, 8, 64)], r        # F = 4 global frames, 16 / 2 pixels, 2 x 4 x 8 rows
        assert r["e_single"] < 3e-2, r


def test_two_rank_frame_parallel_plan_matches_single_rank_and_oracle():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
    for r in res:
        # 2 layout switches per ResBlock (3) and per TemporalTransformer (4) of the tiny net; 4 + 1 gathered norms each
        assert r["a2a"] >= 2 * (3 + 4) and r["ag"] >= 4 * 3 + 4, r
        assert r["e_oracle"] < 2e-2, r
        assert abs(r["e_oracle"] - r["e_single_oracle"]) < 3e-3, r
        assert r["e_single"] < 3e-2, r
        assert r["e_module"] < 3e-2, r
        assert r["e_ddim"] < 6e-2, r
        pp = r["pipe"]
        assert pp["n"] == 2 and pp["shared_w"]
        assert pp["copies"][0] == pp["copies"][1] == pp["a2a_b2"] and pp["copies_b2"] == 2 * pp["a2a_b2"], pp


def _pb_fp_worker(rank, world, port, q):
    """Two prompts per plan OVER a frame-parallel group (round 6): every rank denoises its frames of BOTH samples in one plan of B = 4
    row blocks (pair-major), so a rank's GEMMs see twice the rows of its 1 / world share."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp
        plan_interp.install(_Patch)
        from oracle.unet_ref import UNetCfg, unet_forward
        from oracle.weights import random_state_dict, unet_param_shapes
        from videomv_amd.comm import FrameComm
        from videomv_amd.registry import MODEL, DIFFUSION
        from videomv_amd.unet_t2v import gather_frames
        import videomv_amd  # noqa: F401
        ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
        sd = random_state_dict(unet_param_shapes(ocfg), 99)
        F_, H, W, L = 4, 8, 8, 5
        g = torch.Generator().manual_seed(6)
        x = torch.randn(2, 4, F_, H, W, generator=g)
        yc, y0 = torch.randn(2, L, 1024, generator=g), torch.randn(1, L, 1024, generator=g)
        cam = torch.randn(1, F_, 16, generator=g)
        m = MODEL.build(dict(type="UNetSD_T2VBase", **{k: v for k, v in CFG.items()}))
        m.load_state_dict(sd, strict=False)
        dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                                   schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                                   mean_type="eps", var_type="fixed_small"))
        kc, ku = dict(y=yc, camera_data=cam), dict(y=y0, camera_data=cam)
        t2 = torch.tensor([501, 501])
        # unsharded: both prompts in one plan
        _, rows_full = m.forward_cfg_rows(x, t2, kc, ku)
        rows_full = rows_full.clone()
        x_full = x.clone()
        dif.ddim_step_hip(x_full, 501, m, kc, ku, 9.0, 500)
        # sharded
        comm = FrameComm()
        m.set_frame_parallel(comm)
        fl = F_ // world
        sl = slice(rank * fl, (rank + 1) * fl)
        eng, rows = m.forward_cfg_rows(x[:, :, sl].contiguous(), t2, kc, ku)
        T, Tl = F_ * H * W, fl * H * W
        out = dict(rank=rank, B=eng.B, Fl=eng.F, n_breaks=len(eng.breaks), errs=[], e_full=[])
        for s_ in range(2):
            for br, yy in enumerate((yc[s_:s_ + 1], y0)):
                blk = rows[(2 * s_ + br) * Tl:(2 * s_ + br + 1) * Tl, :4].reshape(fl, H * W, 4).permute(2, 0, 1).reshape(1, 4, fl, H, W)
                ref = unet_forward(sd, ocfg, x[s_:s_ + 1], torch.tensor([501]), yy, cam)[:, :, sl]
                full = rows_full[(2 * s_ + br) * T:(2 * s_ + br + 1) * T, :4].reshape(F_, H * W, 4).permute(2, 0, 1).reshape(1, 4, F_, H, W)[:, :, sl]
                out["errs"].append(rel_l2(blk, ref))
                out["e_full"].append(rel_l2(blk, full))
        xs = x[:, :, sl].clone().contiguous()
        dif.ddim_step_hip(xs, 501, m, kc, ku, 9.0, 500)
        out["e_ddim"] = rel_l2(gather_frames(comm, xs), x_full)
        # the whole loop through the sampler API: fused path, gathered at the end
        xl = dif.ddim_sample_loop(noise=x.clone(), model=m, model_kwargs=[kc, ku], guide_scale=9.0, ddim_timesteps=2, eta=0.0)
        out["loop_shape"] = tuple(xl.shape)
        m.set_frame_parallel(None)
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(dict(error=traceback.format_exc()))


def test_two_prompts_per_plan_over_a_frame_parallel_group_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pb_fp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
    for r in res:
        assert r["B"] == 4 and r["Fl"] == 2 and r["n_breaks"] > 0, r
        assert max(r["errs"]) < 2e-2 and max(r["e_full"]) < 3e-2, r              # each (sample, branch) block: oracle / unsharded plan
        assert r["e_ddim"] < 6e-2 and r["loop_shape"] == (2, 4, 4, 8, 8), r


def _entrance_worker(rank, world, port, tmp, q, prompt_batch=1):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        torch.set_num_threads(2)
        from tests import plan_interp
        plan_interp.install(_Patch)
        if prompt_batch is not None and prompt_batch != 1 or os.environ.get("VMV_TEST_NO_STEP_DRAWS") == "1":
            torch.randn_like = lambda x, **kw: torch.zeros_like(x)      # (CPU path: per-step draws at eta = 0 off the global RNG, see test_entrance_cpu)
        from videomv_amd.config import Config
        from videomv_amd.registry import INFER_ENGINE
        import videomv_amd.entrance  # noqa: F401
        argv = ["--cfg", "configs/t2v_infer.yaml", "device", "cpu", "allow_random_init", "True", "num_views", "4",
                "ddim_timesteps", "2", "test_list_path", os.path.join(tmp, "prompts.txt"), "log_dir", os.path.join(tmp, f"out{world}"),
                "UNet.num_heads", "2", "UNet.num_res_blocks", "1", "UNet.dim_mult", "[1, 2]", "test_model", "none.pth",
                "UNet.use_lgm_refine", "False", "frame_parallel", "True", "prompt_batch", str(prompt_batch or 1)]
        cu = Config(load=True, argv=argv)
        cu.cfg_dict["UNet"]["dim"] = 64
        cu.cfg_dict["UNet"]["attn_scales"] = [1.0, 0.5]
        cu.cfg_dict["resolution"] = [64, 64]
        cu.cfg_dict["auto_encoder"] = {"type": "AutoencoderKL", "embed_dim": 4, "pretrained": "none.pth",
                                       "ddconfig": {"double_z": True, "z_channels": 4, "resolution": 64, "in_channels": 3,
                                                    "out_ch": 3, "ch": 32, "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2,
                                                    "attn_resolutions": [], "dropout": 0.0}}
        cfg = INFER_ENGINE.build(dict(type=cu.TASK_TYPE), cfg_update=cu.cfg_dict)
        q.put(dict(rank=rank, outputs=list(cfg.outputs)))
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


def test_entrance_frame_parallel_two_ranks_matches_one_rank(tmp_path):
    """`frame_parallel True` through the t2v entrance: 2 gloo ranks x 2 views produce the views a single rank produces
    (same seed => same noise / text features / random-init weights).  Video rel-L2 <= 8e-2: two CFG-9 DDIM steps amplify
    the decorrelated bf16 rounding noise of the two plans (see the module docstring)."""
    (tmp_path / "prompts.txt").write_text("a wooden chair\n")
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        q, port = ctx.Queue(), _free_port()
        procs = [ctx.Process(target=_entrance_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
        for r in res:
            assert "error" not in r, r["error"]
        results[world] = {r["rank"]: r["outputs"] for r in res}
    assert len(results[1][0]) == 1 and len(results[2][0]) == 1 and results[2][1] == []      # rank 0 writes the sample
    a = torch.load(results[1][0][0])
    b = torch.load(results[2][0][0])
    assert a["video"].shape == b["video"].shape == (1, 3, 4, 64, 64)
    assert torch.isfinite(b["video"]).all()
    assert rel_l2(b["latent"], a["latent"]) < 8e-2, rel_l2(b["latent"], a["latent"])
    assert rel_l2(b["video"], a["video"]) < 8e-2, rel_l2(b["video"], a["video"])


def test_entrance_frame_parallel_with_prompt_batch_two_ranks(tmp_path):
    """`frame_parallel True` + `prompt_batch 2` through the t2v entrance: 2 gloo ranks, each denoising its 2 views of BOTH prompts in one
    pass, write the samples one rank writes one prompt at a time (same noises: drawn per prompt in list order)."""
    (tmp_path / "prompts.txt").write_text("a wooden chair\na red teapot\n")
    ctx = mp.get_context("spawn")
    results = {}
    for world, pb in ((1, None), (2, 2)):
        q, port = ctx.Queue(), _free_port()
        if pb is None:
            os.environ["VMV_TEST_NO_STEP_DRAWS"] = "1"
        procs = [ctx.Process(target=_entrance_worker, args=(r, world, port, str(tmp_path), q, pb)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=900) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
        os.environ.pop("VMV_TEST_NO_STEP_DRAWS", None)
        for r in res:
            assert "error" not in r, r["error"]
        results[world] = {r["rank"]: r["outputs"] for r in res}
    assert len(results[1][0]) == 2 and len(results[2][0]) == 2 and results[2][1] == []
    for fa, fb in zip(sorted(results[1][0]), sorted(results[2][0])):
        a, b = torch.load(fa), torch.load(fb)
        assert a["caption"] == b["caption"] and a["video"].shape == b["video"].shape
        assert torch.isfinite(b["video"]).all() and rel_l2(b["video"], a["video"]) < 8e-2, rel_l2(b["video"], a["video"])


def _cfgpar_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp
        plan_interp.install(_Patch)
        from oracle.unet_ref import UNetCfg
        from oracle.weights import random_state_dict, unet_param_shapes
        from videomv_amd.comm import CfgFrameComm
        from videomv_amd.registry import MODEL, DIFFUSION
        from videomv_amd.unet_t2v import gather_frames
        import videomv_amd  # noqa: F401
        ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
        sd = random_state_dict(unet_param_shapes(ocfg), 99)
        F_, H, W, L = 4, 8, 8, 5
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1, 4, F_, H, W, generator=g)
        y = torch.randn(2, L, 1024, generator=g)
        cam = torch.randn(1, F_, 16, generator=g)
        m = MODEL.build(dict(type="UNetSD_T2VBase", **{k: v for k, v in CFG.items()}))
        m.load_state_dict(sd, strict=False)
        diff = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                                    schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                                    mean_type="eps", var_type="fixed_small"))
        kc, ku = dict(y=y[:1], camera_data=cam), dict(y=y[1:], camera_data=cam)
        ref = x.clone()
        diff.ddim_step_hip(ref, 501, m, kc, ku, 9.0, 500)                       # single-rank batched plan
        comm = CfgFrameComm()
        m.set_frame_parallel(comm)
        fl = F_ // comm.world
        xt = x[:, :, comm.rank * fl:(comm.rank + 1) * fl].clone().contiguous()
        diff.ddim_step_hip(xt, 501, m, kc, ku, 9.0, 500)
        full = gather_frames(comm, xt)
        eng = m._pipe["engs"][0]
        # whole loop through the sampler API as well (2 steps)
        xl = diff.ddim_sample_loop(noise=x, model=m, model_kwargs=[kc, ku], guide_scale=9.0, ddim_timesteps=2, eta=0.0)
        m.set_frame_parallel(None)
        xr = diff.ddim_sample_loop(noise=x, model=m, model_kwargs=[kc, ku], guide_scale=9.0, ddim_timesteps=2, eta=0.0)
        q.put(dict(rank=rank, branch=comm.branch, fp_world=comm.world, e_step=rel_l2(full, ref), e_loop=rel_l2(xl, xr),
                   B=eng.B, breaks=len(eng.breaks), finite=bool(torch.isfinite(xl).all())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4])
def test_cfg_parallel_times_frame_parallel(world):
    """comm.CfgFrameComm on gloo: world 2 = one CFG branch per rank (no frame sharding), world 4 = 2 branch groups x 2
    frame shards.  Every rank must end a fused CFG + DDIM step / a 2-step loop with the single-rank result (same tolerance
    as the frame-parallel tests: decorrelated storage rounding, amplified by CFG 9)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfgpar_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
    for r in res:
        assert r["B"] == 1 and r["fp_world"] == world // 2 and r["branch"] == r["rank"] // (world // 2)
        assert r["finite"] and r["e_step"] < 6e-2 and r["e_loop"] < 8e-2, r
    assert {r["branch"] for r in res} == {0, 1}


# ------------------------------------------------------------------------------------------------- world 8 (BASELINE configs[2])
def _w8_reference(path):
    """Single-rank results at F = 24 (computed once, in the parent): batched eps rows and one fused CFG + DDIM step."""
    from tests import plan_interp
    plan_interp.install(_Patch)
    from oracle.unet_ref import UNetCfg
    from oracle.weights import random_state_dict, unet_param_shapes
    from videomv_amd.unet_engine import UNetEngine
    from videomv_amd.registry import MODEL, DIFFUSION
    import videomv_amd  # noqa: F401
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 99)
    B, F_, H, W, L = 2, 24, 8, 8, 5
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, F_, H, W, generator=g)
    t = torch.tensor([501])
    y = torch.randn(B, L, 1024, generator=g)
    cam = torch.randn(1, F_, 16, generator=g)
    eng = UNetEngine(CFG, sd, B, F_, H, W, L, torch.device("cpu"), n_t=1)
    eng.set_context(y); eng.set_camera(cam)
    eng.forward_rows(x, t)
    m = MODEL.build(dict(type="UNetSD_T2VBase", **{k: v for k, v in CFG.items()}))
    m.load_state_dict(sd, strict=False)
    diff = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                                schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                                mean_type="eps", var_type="fixed_small"))
    xt = x.clone()
    diff.ddim_step_hip(xt, 501, m, dict(y=y[:1], camera_data=cam), dict(y=y[1:], camera_data=cam), 9.0, 500)
    torch.save(dict(sd=sd, x=x, y=y, cam=cam, eps=eng.eps_ncfhw(), x_next=xt), path)


def _w8_worker(rank, world, port, path, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp
        plan_interp.install(_Patch)
        from videomv_amd.comm import FrameComm, CfgFrameComm
        from videomv_amd.unet_engine import UNetEngine
        from videomv_amd.registry import MODEL, DIFFUSION
        from videomv_amd.unet_t2v import gather_frames
        import videomv_amd  # noqa: F401
        ref = torch.load(path)
        sd, x, y, cam = ref["sd"], ref["x"], ref["y"], ref["cam"]
        B, F_, H, W, L = 2, 24, 8, 8, 5
        t = torch.tensor([501])
        out = dict(rank=rank)
        # (i) the single B = 2 plan, frames sharded 3 per rank, pixels HW / 8 per rank in the temporal ops
        comm = FrameComm()
        fl = F_ // world
        sl = slice(rank * fl, (rank + 1) * fl)
        eng = UNetEngine(CFG, sd, B, F_, H, W, L, torch.device("cpu"), n_t=1, comm=comm)
        eng.set_context(y); eng.set_camera(cam)
        eng.forward_rows(x[:, :, sl].contiguous(), t)
        out.update(fl=fl, e_plan=rel_l2(eng.eps_ncfhw(), ref["eps"][:, :, sl]), a2a=comm.n_all_to_all, ag=comm.n_all_gather,
                   collectives_per_plan=len(eng.breaks))
        # (ii) branch-pipelined fused CFG + DDIM step through the module API
        m = MODEL.build(dict(type="UNetSD_T2VBase", **{k: v for k, v in CFG.items()}))
        m.load_state_dict(sd, strict=False)
        diff = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                                    schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                                    mean_type="eps", var_type="fixed_small"))
        kc, ku = dict(y=y[:1], camera_data=cam), dict(y=y[1:], camera_data=cam)
        m.set_frame_parallel(comm)
        xt = x[:, :, sl].clone().contiguous()
        diff.ddim_step_hip(xt, 501, m, kc, ku, 9.0, 500)
        out["e_pipe"] = rel_l2(gather_frames(comm, xt), ref["x_next"])
        out["pipe_plans"] = len(m._pipe["engs"])
        # (iii) CFG-parallel x frame-parallel: 2 branch groups x 4 frame shards (6 frames per rank)
        m.set_frame_parallel(None)
        cc = CfgFrameComm()
        m.set_frame_parallel(cc)
        f2 = F_ // cc.world
        xt = x[:, :, cc.rank * f2:(cc.rank + 1) * f2].clone().contiguous()
        diff.ddim_step_hip(xt, 501, m, kc, ku, 9.0, 500)
        out.update(e_cfgpar=rel_l2(gather_frames(cc, xt), ref["x_next"]), branch=cc.branch, fp_world=cc.world)
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


def test_world_8_three_frames_per_rank(tmp_path):
    """BASELINE configs[2]'s partitioning — 24 views, 8 ranks, 3 frames per rank, HW / 8 pixels per rank in the temporal
    operators — on 8 gloo ranks: the single B = 2 sharded plan, the branch-pipelined fused step and CFG-parallel x
    frame-parallel (2 x 4, 6 frames per rank) against the single-rank results (tolerances: module docstring)."""
    path = str(tmp_path / "ref.pt")
    _w8_reference(path)
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_w8_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
    assert sorted(r["rank"] for r in res) == list(range(8))
    for r in res:
        assert r["fl"] == 3 and r["pipe_plans"] == 2 and r["fp_world"] == 4 and r["branch"] == r["rank"] // 4
        assert r["e_plan"] < 3e-2 and r["e_pipe"] < 6e-2 and r["e_cfgpar"] < 6e-2, r
        assert r["a2a"] >= 2 * (3 + 4) and r["ag"] >= 4 * 3 + 4, r


# ------------------------------------------------------------------------------------------------------- north-star KV all-gather
def _kvg_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp
        plan_interp.install(_Patch)
        from oracle.weights import random_state_dict, unet_param_shapes
        from oracle.unet_ref import UNetCfg
        from videomv_amd.comm import FrameComm
        from videomv_amd.unet_engine import UNetEngine
        ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
        sd = random_state_dict(unet_param_shapes(ocfg), 99)
        B, F_, H, W, L = 1, 8, 8, 8, 5
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1, 4, F_, H, W, generator=g)
        t = torch.tensor([501])
        y = torch.randn(B, L, 1024, generator=g)
        cam = torch.randn(1, F_, 16, generator=g)
        dev = torch.device("cpu")
        ref = UNetEngine(CFG, sd, B, F_, H, W, L, dev, n_t=1)
        ref.set_context(y); ref.set_camera(cam)
        ref.forward_rows(x, t)
        eps_single = ref.eps_ncfhw()
        fl = F_ // world
        sl = slice(rank * fl, (rank + 1) * fl)
        res = {}
        for mode in ("switch", "kv_gather"):
            os.environ["VMV_FP_TEMPORAL"] = mode
            comm = FrameComm()
            eng = UNetEngine(CFG, sd, B, F_, H, W, L, dev, n_t=1, comm=comm)
            eng.set_context(y); eng.set_camera(cam)
            eng.forward_rows(x[:, :, sl].contiguous(), t)
            res[mode] = dict(e=rel_l2(eng.eps_ncfhw(), eps_single[:, :, sl]), a2a=comm.n_all_to_all, ag=comm.n_all_gather,
                             packs=sum(1 for l in eng.S.labels if l.endswith(".kv.pack")))
        q.put(dict(rank=rank, **res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_kv_all_gather_temporal_mode(world):
    """BASELINE's north-star form of configs[2] (VMV_FP_TEMPORAL=kv_gather): the frames stay sharded through the
    TemporalTransformers and every temporal attention all-gathers [K | V] (Nq = F / R local frames against Nk = F keys per pixel)
    instead of the two layout switches around the block — same eps as the single-rank plan and as the default 'switch' mode; the
    collective COUNT is the same (2 per TemporalTransformer either way), the bytes are not."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kvg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
    for r in res:
        sw, kv = r["switch"], r["kv_gather"]
        assert sw["e"] < 3e-2 and kv["e"] < 3e-2, r
        assert kv["packs"] > 0 and sw["packs"] == 0
        assert kv["a2a"] == sw["a2a"] - kv["packs"] and kv["ag"] == sw["ag"] + kv["packs"], r      # 2 switches -> 2 gathers per block


def test_simulated_rank_plan_records_its_collectives(monkeypatch):
    """comm.SimComm (vmv_comm_create_sim): rank 0's plan of a 4-GPU frame-parallel run built in ONE process.  The collectives are
    plan ops (VMV_OP_COMM) — no Python break points — with the same count and per-rank byte sizes as the gloo plans' break points
    (2 layout switches per temporal block + one totals gather per all-frame GroupNorm), the plan replays through the interpreter
    (sim semantics: recv = this rank's own bytes) to finite values, and the branch-pipelined sampler path runs on it.  With every
    rank holding the SAME frames and a pixel-periodic image the simulated exchange IS the real one — but zero-padded convolutions
    break pixel periodicity, so exactness is only asserted at world 1 (where sim == real == identity)."""
    from tests import plan_interp
    plan_interp.install(monkeypatch)
    import ctypes as C
    from oracle.unet_ref import UNetCfg
    from oracle.weights import random_state_dict, unet_param_shapes
    from videomv_amd import _lib as L
    from videomv_amd.comm import SimComm
    from videomv_amd.unet_engine import UNetEngine
    from videomv_amd.registry import MODEL, DIFFUSION
    import videomv_amd  # noqa: F401
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 99)
    B, F_, H, W, Lc, R = 2, 4, 8, 8, 5, 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, F_, H, W, generator=g)
    t = torch.tensor([501])
    y = torch.randn(B, Lc, 1024, generator=g)
    cam = torch.randn(1, F_, 16, generator=g)
    dev = torch.device("cpu")
    sc = SimComm(R, 0)
    lib = L.load()
    assert lib.vmv_comm_world(sc.handle) == R and lib.vmv_comm_rank(sc.handle) == 0 and lib.vmv_comm_is_sim(sc.handle) == 1
    eng = UNetEngine(CFG, sd, B, F_, H, W, Lc, dev, n_t=1, comm=sc)
    assert eng.F == 1 and not eng.breaks                                   # nothing left for Python to issue
    comm_ops = [(lb, p) for lb, (op, p) in zip(eng.S.labels, eng.S.recorded) if op == L.OP_COMM]
    n_a2a = sum(1 for lb, p in comm_ops if p.kind == L.COMM_ALL_TO_ALL)
    n_ag = sum(1 for lb, p in comm_ops if p.kind == L.COMM_ALL_GATHER)
    blocks = [k for blk in eng.inp + [eng.mid] + eng.outb for k, _, _ in blk]
    n_res, n_tt = blocks.count("res"), blocks.count("tt")                   # (tiny net: 8 ResBlocks, 8 TemporalTransformers)
    assert n_a2a == 2 * (n_res + n_tt) and n_ag == 4 * n_res + n_tt and eng.n_comm_ops == n_a2a + n_ag
    for lb, p in comm_ops:
        assert p.comm == sc.handle and p.bytes > 0 and p.bytes % 16 == 0, lb
        if p.kind == L.COMM_ALL_GATHER:
            assert p.bytes == B * 2048 * 8                                  # B stat groups x GN_TOT int64
    # first layout switch: the [B][1 frame][64 pixels][64 ch] shard in 4 chunks
    first = next(p for lb, p in comm_ops if p.kind == L.COMM_ALL_TO_ALL)
    assert first.bytes == B * 1 * (H * W // R) * 64 * 2
    assert eng.comm_bytes_in == sum(p.bytes * (R - 1) for _, p in comm_ops)
    eng.set_context(y); eng.set_camera(cam)
    eng.forward_rows(x[:, :, :1].contiguous(), t)
    assert torch.isfinite(eng.eps_ncfhw()).all()
    # world 1: the simulated collectives are the identity, and so are the real ones -> the sharded plan equals the plain plan
    ref = UNetEngine(CFG, sd, B, F_, H, W, Lc, dev, n_t=1)
    ref.set_context(y); ref.set_camera(cam); ref.forward_rows(x, t)
    one = UNetEngine(CFG, sd, B, F_, H, W, Lc, dev, n_t=1, comm=SimComm(1, 0))
    one.set_context(y); one.set_camera(cam); one.forward_rows(x, t)
    assert one.n_comm_ops == n_a2a + n_ag and rel_l2(one.eps_ncfhw(), ref.eps_ncfhw()) < 3e-2
    # the sampler on the simulated rank (branch-pipelined: two B = 1 plans)
    m = MODEL.build(dict(type="UNetSD_T2VBase", **{k: v for k, v in CFG.items()}))
    m.load_state_dict(sd, strict=False)
    m.set_frame_parallel(sc)
    dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                               schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                               mean_type="eps", var_type="fixed_small"))
    xt = x[:, :, :1].clone().contiguous()
    dif.ddim_step_hip(xt, 501, m, dict(y=y[:1], camera_data=cam), dict(y=y[1:], camera_data=cam), 9.0, 500)
    assert torch.isfinite(xt).all()
    pe = m._pipe["engs"]
    assert len(pe) == 2 and all(not e.breaks and e.n_comm_ops == n_a2a + n_ag for e in pe)
    assert pe[0].comm.handle != pe[1].comm.handle                            # a communicator per branch stream
    # CFG-parallel x frame-parallel on a simulated rank (comm.SimCfgFrameComm: 2 branch groups of R / 2 ranks): ONE B = 1 plan over
    # 2 * F / R frames — half the launches of the pipelined pair — and the partner's eps rows are this rank's own
    from videomv_amd.comm import SimCfgFrameComm
    sc2 = SimCfgFrameComm(R)
    assert (sc2.branch, sc2.rank, sc2.world) == (0, 0, R // 2)
    m.set_frame_parallel(sc2)
    xt2 = x[:, :, :2].clone().contiguous()
    dif.ddim_step_hip(xt2, 501, m, dict(y=y[:1], camera_data=cam), dict(y=y[1:], camera_data=cam), 9.0, 500)
    assert torch.isfinite(xt2).all()
    pc = m._pipe["engs"]
    assert len(pc) == 1 and pc[0].B == 1 and pc[0].F == 2 and not pc[0].breaks and pc[0].n_comm_ops == n_a2a + n_ag
    assert pc[0].S.nops < sum(e.S.nops for e in pe)
    eps2 = m._pipe["eps"].view(2, -1)
    assert torch.equal(eps2[0], eps2[1])                                     # (sim: both branch slots hold this rank's rows)


LGM_TINY = dict(down_channels=(32, 64), down_attention=(False, True), mid_attention=True, up_channels=(64, 32),
                up_attention=(True, False), num_heads=2, input_size=64, splat_size=64, output_size=128)
VAE_TINY = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
                num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def _lgm_setup():
    """Tiny T2V UNet with the LGM branch + tiny VAE, zero-initialised layers re-randomised (4 views, latent 8 x 8 -> 64-px decodes)."""
    from videomv_amd.registry import MODEL, DIFFUSION, AUTO_ENCODER
    from videomv_amd.lgm import prepare_gs_data
    from videomv_amd.camera import entrance_camera_data
    import videomv_amd  # noqa: F401
    torch.manual_seed(0)
    cfg = dict(in_dim=4, dim=64, y_dim=1024, context_dim=1024, out_dim=4, dim_mult=[1], num_heads=2, head_dim=64, num_res_blocks=1,
               attn_scales=[1.0], use_camera_condition=True, use_lgm_refine=True, lgm_opt=LGM_TINY)
    m = MODEL.build(dict(type="UNetSD_T2VBase", **cfg)).eval()
    g = torch.Generator().manual_seed(7)
    for p_ in m.parameters():
        if p_.abs().max() == 0:
            p_.data.normal_(0, 0.02, generator=g)
    m._invalidate()
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=VAE_TINY, embed_dim=4))
    dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                               schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012, zero_terminal_snr=False),
                               mean_type="eps", var_type="fixed_small"))
    F_ = 4
    xt = torch.randn(1, 4, F_, 8, 8, generator=g)
    y, y0 = torch.randn(1, 7, 1024, generator=g), torch.randn(1, 7, 1024, generator=g)
    cam = entrance_camera_data(F_, elevation=15, camera_distance=2.0)
    gs_data = prepare_gs_data(cam, m.lgm_opt)
    kw = [dict(y=y, camera_data=cam, gs_data=gs_data), dict(y=y0, camera_data=cam, gs_data=gs_data)]
    return m, vae, dif, xt, kw


def test_lgm_branch_on_a_view_slice_equals_the_slice_of_the_whole(monkeypatch):
    """The piece of the LGM branch a frame-parallel rank runs (lgm.LgmRefiner.latent_z_pair(views=(f0, n))): x0 of the key views, the
    4-view decode and the LGM U-Net are the whole sample's; renders and the VAE re-encode cover views [f0, f0 + n) only, with the slice
    of the noise the unsharded call draws for them.  Per-view renders, the per-frame encoder and the sliced noise make the result the
    corresponding slice of the unsharded latent_z (same arithmetic: 1e-5)."""
    from tests import plan_interp
    plan_interp.install(monkeypatch)
    m, vae, dif, xt, kw = _lgm_setup()
    ref = m.lgm_refiner(torch.device("cpu"))
    g = torch.Generator().manual_seed(3)
    eps_rows = torch.randn(2 * 4 * 64, 4, generator=g)
    torch.manual_seed(5)
    full = ref.latent_z_pair(eps_rows, 4, xt, 1.2, 0.7, vae, dict(kw[0]["gs_data"]))
    for f0, n in ((0, 2), (2, 2), (1, 3)):
        torch.manual_seed(5)
        part = ref.latent_z_pair(eps_rows, 4, xt, 1.2, 0.7, vae, dict(kw[0]["gs_data"]), views=(f0, n))
        for br in range(2):
            assert part[br].shape == (1, 4, n, 8, 8)
            assert torch.allclose(part[br], full[br][:, :, f0:f0 + n], atol=1e-5, rtol=1e-5), (f0, n, br)


def test_lgm_refined_step_honours_clamp_and_eta(monkeypatch):
    """``ddim_step_lgm`` dropped the x0 clamp and the stochastic term that the reference applies on EVERY step, refined ones
    included (diffusion_ddim.py:200-205, 233-243).  The fused refined step with clamp + eta against the generic two-forward
    ``ddim_sample`` (the reference's own structure: model(..., autoencoder=...) twice -> CFG on latent_z -> clamp -> update + sigma *
    noise) from the same seeds: same posterior draws (cond, then uncond), then the same sigma-noise draw."""
    from tests import plan_interp
    plan_interp.install(monkeypatch)
    m, vae, dif, xt, kw = _lgm_setup()
    t = torch.full((1,), 581, dtype=torch.long)
    outs = {}
    for clamp, eta in ((0.35, 0.8), (None, 0.0)):
        torch.manual_seed(21)
        got = xt.clone()
        dif.ddim_step_lgm(got, 581, m, kw[0], kw[1], 9.0, 20, vae, clamp=clamp, eta=eta)
        assert torch.isfinite(got).all()
        outs[(clamp, eta)] = got
    torch.manual_seed(21)
    want, _ = dif.ddim_sample(xt.clone(), t, m, vae, kw, 0.35, None, None, 9.0, 50, 0.8)
    rel = float((outs[(0.35, 0.8)] - want).norm() / want.norm())
    assert rel < 2e-2, rel
    # and the options are not no-ops
    assert float((outs[(0.35, 0.8)] - outs[(None, 0.0)]).abs().max()) > 1e-2


def test_frame_parallel_sigma_noise_is_the_unsharded_runs_slice():
    """frame-parallel ranks share the seed, so ``randn_like(local xt)`` gave every shard the SAME noise block.  Each rank now
    draws the whole sample's noise and keeps its frames."""
    from videomv_amd.diffusion_ddim import DiffusionDDIM

    class _C:
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

    class _U:
        frame_comm = None
    xt = torch.zeros(1, 4, 2, 8, 8)
    torch.manual_seed(4)
    full = torch.randn(1, 4, 6, 8, 8)
    parts = []
    for r in range(3):
        u = _U(); u.frame_comm = _C(r, 3)
        torch.manual_seed(4)
        parts.append(DiffusionDDIM._step_noise(xt, u))
    assert torch.equal(torch.cat(parts, dim=2), full)
    assert not torch.equal(parts[0], parts[1])
    torch.manual_seed(4)
    assert torch.equal(DiffusionDDIM._step_noise(full, _U()), full)


def _lgm_fp_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp as pi
        pi.install(_Patch)
        from videomv_amd.comm import FrameComm
        from videomv_amd.unet_t2v import gather_frames
        m, vae, dif, xt, kw = _lgm_setup()
        torch.manual_seed(9)
        x_single = xt.clone()
        dif.ddim_step_lgm(x_single, 581, m, kw[0], kw[1], 9.0, 20, vae)
        comm = FrameComm()
        m.set_frame_parallel(comm)
        fl = xt.shape[2] // world
        x_loc = xt[:, :, rank * fl:(rank + 1) * fl].clone().contiguous()
        torch.manual_seed(9)
        dif.ddim_step_lgm(x_loc, 581, m, kw[0], kw[1], 9.0, 20, vae)
        x_all = gather_frames(comm, x_loc)
        d = (x_all - x_single).flatten()
        cos = float(torch.nn.functional.cosine_similarity(x_all.flatten(), x_single.flatten(), dim=0))
        q.put(dict(rank=rank, finite=bool(torch.isfinite(x_all).all()), rel=float(d.norm() / x_single.norm()), cos=cos,
                   renders=len(m.lgm_refiner(torch.device("cpu")).renderer.last_num_rendered) if hasattr(m.lgm_refiner(torch.device("cpu")).renderer, "last_num_rendered") else -1))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


def test_lgm_refined_step_frame_parallel_two_ranks():
    """BASELINE configs[2] x configs[4] (round 4: no longer NotImplementedError): one LGM-refined DDIM step with the 4 views sharded
    over 2 gloo ranks — gathered (x_t, eps) for the key views, replicated decode + LGM U-Net, each rank renders / re-encodes its own 2
    views with the unsharded run's posterior noise — against the single-rank step from the same seeds.  The two runs differ by the 16-bit
    rounding of the sharded UNet plan (GroupNorm fold order), which the random-weight decode -> LGM -> render -> encode chain amplifies:
    statistical agreement as SURVEY 8d asks for the LGM steps (rel-L2 <= 0.25, cosine >= 0.97)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lgm_fp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
        assert r["finite"] and r["rel"] < 0.25 and r["cos"] > 0.97, r


def _lgm_cfgpar_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests import plan_interp as pi
        pi.install(_Patch)
        from videomv_amd.comm import CfgFrameComm
        from videomv_amd.unet_t2v import gather_frames
        m, vae, dif, xt, kw = _lgm_setup()
        torch.manual_seed(9)
        x_single = xt.clone()
        dif.ddim_step_lgm(x_single, 581, m, kw[0], kw[1], 9.0, 20, vae)
        comm = CfgFrameComm()
        m.set_frame_parallel(comm)
        fl = xt.shape[2] // comm.world
        x_loc = xt[:, :, comm.rank * fl:(comm.rank + 1) * fl].clone().contiguous()
        torch.manual_seed(9)
        dif.ddim_step_lgm(x_loc, 581, m, kw[0], kw[1], 9.0, 20, vae)
        x_all = gather_frames(comm, x_loc)
        d = (x_all - x_single).flatten()
        cos = float(torch.nn.functional.cosine_similarity(x_all.flatten(), x_single.flatten(), dim=0))
        # the partner ranks (same frames, other branch) must hold IDENTICAL x_{t-1}: they applied the same update to the same pair of latent_z
        both = torch.empty(2 * x_all.numel())
        dist.all_gather_into_tensor(both, x_all.reshape(-1).contiguous())
        both = both.view(2, -1)
        q.put(dict(rank=rank, finite=bool(torch.isfinite(x_all).all()), rel=float(d.norm() / x_single.norm()), cos=cos,
                   pair_equal=bool(torch.equal(both[0], both[1]))))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))


def test_lgm_refined_step_cfg_parallel_two_ranks():
    """Round 5 (it raised NotImplementedError): one LGM-refined DDIM step with CFG-parallel ranks — rank 0 runs the
    conditional branch (UNet + decode + LGM U-Net + renders + re-encode), rank 1 the unconditional one; they swap their latent_z and apply
    the CFG-on-latent_z update identically.  Round 6: every rank consumes BOTH posterior draws of the unsharded run (cond,
    then uncond) and uses its branch's, so the two groups' noises are independent and equal to the single-rank run's; what is left is the
    16-bit rounding of a B = 1 plan against the B = 2 plan through the random-weight decode -> LGM -> render -> encode chain (the same
    bound as the frame-parallel test above), and bitwise agreement between the two ranks of the pair."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lgm_cfgpar_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
        assert r["finite"] and r["pair_equal"] and r["rel"] < 0.25 and r["cos"] > 0.97, r
