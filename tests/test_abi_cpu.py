"""`not gpu`: the C-ABI shared library loads here (no GPU) and exports every symbol include/vmv.h declares; the
ctypes argument blocks have the C layout; argument validation (which runs before any launch) reports VMV_E* codes."""
import ctypes as C
import os
import re

import pytest
import torch

from videomv_amd import _lib as L
from videomv_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vmv.h")).read()
    declared = set(re.findall(r"\b(vmv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    assert lib.vmv_abi_version() == L.ABI_VERSION == 11


def test_struct_layouts_match_c():
    lib = L.load()
    for which, st in ((L.OP_GEMM, L.GemmParams), (L.OP_GN_STATS, L.GroupNormParams), (L.OP_GN_APPLY, L.GroupNormParams),
                      (L.OP_LAYERNORM, L.LayerNormParams), (L.OP_ATTENTION, L.AttnParams), (L.OP_SOFTMAX, L.SoftmaxParams),
                      (L.OP_COPY, L.CopyParams), (L.OP_FF, L.FfParams), (100, L.DdimParams),
                      (101, L.GemmSeg), (102, L.SeqMap)):
        assert lib.vmv_sizeof(which) == C.sizeof(st), st.__name__


def test_argument_validation_needs_no_gpu():
    lib = L.load()
    a = torch.zeros(64, 64, dtype=L.elem())
    p = ops.gemm_params(64, 64, ops.linear_segs([(a, 64, 64)]), a, a, 64)
    p.N = 63
    assert lib.vmv_gemm(C.byref(p), None) == -1
    p = ops.gemm_params(64, 64, ops.linear_segs([(a, 64, 64)]), None, a, 64)
    assert lib.vmv_gemm(C.byref(p), None) == -3
    p = ops.gemm_params(64, 64, ops.linear_segs([(a.data_ptr() + 2, 64, 64)]), a, a, 64)
    assert lib.vmv_gemm(C.byref(p), None) == -2
    assert lib.vmv_gemm(None, None) == -3
    assert b"VMV_EALIGN" in lib.vmv_error_string(-2)
    ln = ops.ln_params(a, 64, a, 64, None, None, 64, 64)
    assert lib.vmv_layernorm(C.byref(ln), None) == -3
    ln = ops.ln_params(a, 64, a, 64, torch.zeros(64), torch.zeros(64), 64, 4096)
    assert lib.vmv_layernorm(C.byref(ln), None) == -4


def test_plan_records_and_sizes():
    lib = L.load()
    plan = lib.vmv_plan_create()
    a = torch.zeros(64, 64, dtype=L.elem())
    p = ops.gemm_params(64, 64, ops.linear_segs([(a, 64, 64)]), a, a, 64)
    assert lib.vmv_plan_add(plan, L.OP_GEMM, C.byref(p), C.sizeof(p)) == 0
    assert lib.vmv_plan_add(plan, L.OP_GEMM, C.byref(p), C.sizeof(p) - 8) == -1     # wrong block size
    assert lib.vmv_plan_add(plan, 99, C.byref(p), C.sizeof(p)) == -1
    assert lib.vmv_plan_size(plan) == 1
    lib.vmv_plan_destroy(plan)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "lib_path", lambda: "/nonexistent/libvmv_hip_f16.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


def test_gemm_tile_policy(monkeypatch):
    """vmv_gemm_pick_tile (host logic): which kernel family the default policy gives the UNet's / VAE's characteristic GEMMs at
    latent 24x40x64 — pins DESIGN 4.1's table (pointers are never dereferenced: fake non-null addresses)."""
    for k in ("VMV_GEMM_POLICY", "VMV_GEMM_XGLDS", "VMV_GEMM_XEPI", "VMV_GEMM_ASTAT", "VMV_GEMM_RS", "VMV_GEMM_TILE_GEGLU", "VMV_GEMM_TILE_LIN160", "VMV_GEMM_TILE_LIN128"):
        assert k not in os.environ, "policy overrides must be unset for this test"
    lib = L.load()
    X = 1 << 20          # any non-null, 16-byte aligned address

    def pick(M, N, segs, geom=None, **kw):
        p = ops.gemm_params(M, N, segs, X, X, N // 2 if kw.get("epilogue") == L.EPI_GEGLU else N, geom=geom, **kw)
        return lib.vmv_gemm_pick_tile(C.byref(p))
    M0, M1, M2, M3 = 122880, 30720, 7680, 1920
    g0, g1, g2 = ops.Geom(OH=40, OW=64, IH=40, IW=64), ops.Geom(OH=20, OW=32, IH=20, IW=32), ops.Geom(OH=10, OW=16, IH=10, IW=16)
    # 3x3 / temporal convolutions of the two large levels: the wide-tile kernel; the third level keeps 256 x 160 tiles
    assert pick(M0, 320, ops.conv3x3_segs([(X, 320, 320)]), g0) == L.TILE_X256x320
    assert pick(M0 // 2, 320, ops.conv3x3_segs([(X, 320, 320)]), g0) == L.TILE_X256x320          # shared CFG prefix: one branch
    assert pick(M0, 320, ops.temporal_segs(X, 320, 320), ops.Geom(F=24, P=M0 // 48)) == L.TILE_X256x320
    # (round 5: where its tiles fill whole rounds of the chip the temporal convolution goes to the frame-resident kernel: the first
    #  level at the reference's own 24 x 32 x 32 — 256 tiles of 24 frames x 8 pixels x 320 channels; 640 tiles at 24 x 40 x 64 do not)
    assert pick(2 * 24 * 1024, 320, ops.temporal_segs(X, 320, 320), ops.Geom(F=24, P=1024)) == L.TILE_TFR
    assert pick(M1, 640, ops.conv3x3_segs([(X, 640, 640)]), g1) == L.TILE_X256x320
    assert pick(M2, 1280, ops.conv3x3_segs([(X, 1280, 1280)]), g2) == L.TILE_256x160
    # transformer linears of the two large levels with K = C (qkv / q, out-projections, proj_in / proj_out, GEGLU): the
    # row-stationary kernel; the same with a folded LayerNorm (colsum + ln_eps: statistics taken in the kernel); K = 4C (FF down)
    # and the third level stay on the persistent kernel (192 x 160 / 256 x 128 tiles)
    lin = lambda k: ops.linear_segs([(X, k, k)])
    assert pick(M0, 960, lin(320)) == L.TILE_RS
    assert pick(M0, 960, lin(320), colsum=X, ln_eps=1e-5) == L.TILE_RS
    assert pick(M0 // 2, 960, lin(320), colsum=X, ln_eps=1e-5) == L.TILE_RS                      # shared CFG prefix: one branch
    assert pick(M0, 320, lin(320), residual=X, ldr=320) == L.TILE_RS
    assert pick(M0, 2560, lin(320), epilogue=L.EPI_GEGLU, colsum=X, ln_eps=1e-5) == L.TILE_RS
    assert pick(M1, 5120, lin(640), epilogue=L.EPI_GEGLU, colsum=X, ln_eps=1e-5) == L.TILE_RS
    assert pick(M1, 640, lin(640), residual=X, ldr=640) == L.TILE_RS
    assert pick(M0, 320, lin(1280), residual=X, ldr=320) == L.TILE_P256x160
    # round 4: the third level's LayerNorm-folded qkv / GEGLU on the wide tile's 256 x 256 form (fused epilogues in gemm_xglds.hip);
    # grids that would leave the second round of the chip mostly empty (q, N = 1280: 150 tiles; the middle block) stay persistent
    assert pick(M2, 10240, lin(1280), epilogue=L.EPI_GEGLU, rowstat=X, colsum=X) == L.TILE_X256x256
    assert pick(M2, 3840, lin(1280), rowstat=X, colsum=X) == L.TILE_X256x256
    assert pick(M2, 1280, lin(1280), rowstat=X, colsum=X) == L.TILE_P256x160
    assert pick(M3, 10240, lin(1280), epilogue=L.EPI_GEGLU, rowstat=X, colsum=X) == L.TILE_P256x128
    assert lib.vmv_gemm_rs_ok(C.byref(ops.gemm_params(M0, 960, lin(320), X, X, 960, colsum=X, ln_eps=1e-5))) == 1
    assert lib.vmv_gemm_rs_ok(C.byref(ops.gemm_params(M2, 3840, lin(1280), X, X, 3840, colsum=X, ln_eps=1e-5))) == 0
    # VAE decoder at 24 frames of 320 x 512: 512- / 256-channel levels on 256 x 256 wide tiles, the 128-channel level on 256 x 128
    assert pick(24 * 80 * 128, 512, ops.conv3x3_segs([(X, 512, 512)]), ops.Geom(OH=80, OW=128, IH=80, IW=128)) == L.TILE_X256x256
    assert pick(24 * 160 * 256, 256, ops.conv3x3_segs([(X, 256, 256)]), ops.Geom(OH=160, OW=256, IH=160, IW=256)) == L.TILE_X256x256
    # round 6: the VAE's 128-channel first level (N = 128, millions of rows) on the 8 x 1 wave grid's 512 x 128 tile
    assert pick(24 * 320 * 512, 128, ops.conv3x3_segs([(X, 128, 128)]), ops.Geom(OH=320, OW=512, IH=320, IW=512)) == L.TILE_X512x128
    assert pick(4 * 256 * 256, 128, ops.conv3x3_segs([(X, 128, 128)]), ops.Geom(OH=256, OW=256, IH=256, IW=256)) == L.TILE_X512x128
    assert pick(24 * 320 * 512, 128, lin(128)) != L.TILE_X512x128              # plain rows keep their kernel
    # split-K shapes of the smallest level stay on the 128-row LDS-DMA kernel; a forced tile is returned as is
    assert pick(M3, 1280, ops.conv3x3_segs([(X, 1280, 1280)]), ops.Geom(OH=5, OW=8, IH=5, IW=8), ksplit=8, workspace=X) == L.TILE_G128x160
    assert pick(M0, 320, lin(320), tile=L.TILE_128x64) == L.TILE_128x64
    # grouped weights (the VAE attention's batched Q K^T / V^T): the 128-column LDS-DMA kernels; an incompatible forced tile is refused
    assert pick(24 * 1024, 1024, lin(512), out_fp32=True, wgroup_rows=1024, wgroup_stride=1024 * 512) == L.TILE_P256x128
    assert pick(24 * 512, 1024, lin(512), wgroup_rows=512, wgroup_stride=1024 * 512) == L.TILE_G128x128
    assert pick(24 * 1024, 320, lin(512), wgroup_rows=1024, wgroup_stride=320 * 512, tile=L.TILE_P256x160) == -1


def test_comm_api_host_logic():
    """vmv_comm_* (ABI 8) without a GPU: the simulated communicator is pure host state; RCCL entry points refuse politely while the
    library is not loaded (vmv_comm_load was never called in this process); argument validation of a collective op; plan recording."""
    lib = L.load()
    assert C.sizeof(L.CommParams) == lib.vmv_sizeof(L.OP_COMM)
    assert lib.vmv_comm_loaded() == 0
    idb = (C.c_uint8 * L.COMM_ID_BYTES)()
    assert lib.vmv_comm_unique_id(idb) == -1 and not lib.vmv_comm_create(idb, 2, 0)
    assert not lib.vmv_comm_create_sim(4, 4) and not lib.vmv_comm_create_sim(0, 0)
    h = lib.vmv_comm_create_sim(8, 3)
    assert h and lib.vmv_comm_world(h) == 8 and lib.vmv_comm_rank(h) == 3 and lib.vmv_comm_is_sim(h) == 1
    X = 1 << 20
    assert lib.vmv_comm_run(C.byref(ops.comm_params(h, L.COMM_ALL_TO_ALL, None, X, 64)), None) == -3         # VMV_ENULL
    assert lib.vmv_comm_run(C.byref(ops.comm_params(h, L.COMM_ALL_TO_ALL, X, X, 0)), None) == -1            # VMV_EINVAL
    assert lib.vmv_comm_run(C.byref(ops.comm_params(None, L.COMM_ALL_TO_ALL, X, X, 64)), None) == -3
    plan = lib.vmv_plan_create()
    p = ops.comm_params(h, L.COMM_ALL_GATHER, X, X, 4096)
    assert lib.vmv_plan_add(plan, L.OP_COMM, C.byref(p), C.sizeof(p)) == 0 and lib.vmv_plan_size(plan) == 1
    assert not lib.vmv_plan_capture(None, None) and lib.vmv_graph_launch(None, None) == -3 and lib.vmv_graph_nodes(None) == -3
    lib.vmv_plan_destroy(plan)
    lib.vmv_comm_destroy(h)
    assert b"VMV_ECOMM" in lib.vmv_error_string(-5)


def test_tuned_table_is_well_formed():
    """videomv_amd/tuned_gemm.json (tools/autotune_gemm.py): every key parses as a GEMM signature, every entry names a tile id the header
    declares and a sane split-K factor, and carries the measurement that justified it (>= 3 % faster than the policy)."""
    import json
    with open(os.path.join(ROOT, "videomv_amd", "tuned_gemm.json")) as f:
        tab = json.load(f)
    assert "meta" in tab and tab["meta"]["tool"] == "tools/autotune_gemm.py"
    for elem in [k for k in ("fp16", "bf16") if k in tab]:
        assert tab[elem], elem
        for sig, e in tab[elem].items():
            assert re.fullmatch(r"\d+x\d+x\d+;[0-9:*,]+;e\da\df\dr\dv\d:\d+s\dc\dl\dg\dw\d+;\d+x\d+<\d+x\d+s\d+u\dF\d+P\d+", sig), sig
            assert 1 <= e["tile"] <= 27 and e["ksplit"] in (0, 2, 3, 4, 6, 8, 12, 16)
            assert e["us"] <= 0.97 * e["base_us"] + 1e-6, (sig, e)       # (>= 7 % per launch, or >= 3 % and confirmed by a whole-step A/B)
    # and the engine's hook finds an entry by the signature of the launch it is about to record
    sig = next(iter(tab["fp16"]))
    M, N, K = (int(v) for v in sig.split(";")[0].split("x"))
    assert M > 0 and N % 4 == 0 and K % 8 == 0


def test_autotune_cache_is_merged_only_on_request(monkeypatch, tmp_path):
    """ops.tuned_table(): the packaged table always; the per-user cache VMV_AUTOTUNE=1 writes (autotune.py) only when autotuning is on
    or VMV_TUNED_CACHE names it — a default run depends on nothing outside the package; VMV_TUNED=0 ignores both."""
    import json
    from videomv_amd import autotune
    cache = tmp_path / "cache.json"
    cache.write_text(json.dumps({"fp16": {"1x4x8;0:8*1;e0a0f0r0v0:0s0c0l0g0w0;0x0<0x0s1u0F0P0": {"tile": 4, "ksplit": 0}}, "bf16": {}}))
    key = "1x4x8;0:8*1;e0a0f0r0v0:0s0c0l0g0w0;0x0<0x0s1u0F0P0"
    for k in ("VMV_TUNED", "VMV_TUNED_FILE", "VMV_AUTOTUNE", "VMV_TUNED_CACHE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(ops, "_TUNED", None)
    base = dict(ops.tuned_table())
    assert base and key not in base
    monkeypatch.setattr(ops, "_TUNED", None)
    monkeypatch.setenv("VMV_TUNED_CACHE", str(cache))
    assert autotune.cache_path() == str(cache)
    if L.elem_name() == "fp16":
        assert key in ops.tuned_table() and len(ops.tuned_table()) == len(base) + 1
    monkeypatch.setattr(ops, "_TUNED", None)
    monkeypatch.setenv("VMV_TUNED", "0")
    assert ops.tuned_table() == {}
    # on a CPU engine the measurement is a no-op (nothing to time)
    class _E:
        device = "cpu"
    assert autotune.autotune_engine(_E()) == 0


def test_fill_rule_host_logic():
    """ops.fill_rule (the default since round 5; VMV_TILE_RULES=0 turns it off): where the built-in policy fell back to 128-row tiles it
    picks the 256-row tile and split-K factor that fill one round of the 256 CUs; few-row GEMMs go to 64 x 64 register tiles with more
    K splits; respects what a launch cannot do (no split-K under a folded LayerNorm, 128 columns for GEGLU); every choice is one the
    library accepts for that launch (vmv_gemm_validate: the launch's whole host side, no device)."""
    lib = L.load()
    X = 1 << 20
    lin = lambda k: ops.linear_segs([(X, k, k)])
    conv = lambda c: ops.conv3x3_segs([(X, c, c)])
    g = ops.Geom(OH=8, OW=8, IH=8, IW=8)

    def go(p):
        pol = lib.vmv_gemm_pick_tile(C.byref(p))
        r = ops.fill_rule(p, pol)
        if r is not None:                                   # the forced choice passes the library's host-side validation
            q = ops.gemm_params(p.M, p.N, [ops.Seg(p.seg[i].src, p.seg[i].ld, p.seg[i].k, p.seg[i].mode, p.seg[i].d0, p.seg[i].d1) for i in range(p.nseg)],
                                X, X, p.ldo, tile=r[0], ksplit=r[1], workspace=X if r[1] > 1 else None, epilogue=p.epilogue,
                                geom=ops.Geom(OH=p.OH, OW=p.OW, IH=p.IH, IW=p.IW, F=p.F, P=p.P))
            assert lib.vmv_gemm_pick_tile(C.byref(q)) == r[0], (p.M, p.N, p.ktot, r)
            assert lib.vmv_gemm_validate(C.byref(q)) == 0, (p.M, p.N, p.ktot, r)
        return pol, r
    # the third level at 24x32x32 (48 images of 8 x 8): 192 tiles of 128 x 160 -> 120 tiles of 256 x 128 split in two (measured 183 -> 96 us)
    pol, r = go(ops.gemm_params(3072, 1280, conv(1280), X, X, 1280, geom=g))
    assert pol == L.TILE_G128x160 and r == (L.TILE_256x128, 2)
    # the second level there: 240 tiles of 256 x 128 fill the chip without a split
    pol, r = go(ops.gemm_params(12288, 640, conv(640), X, X, 640, geom=ops.Geom(OH=16, OW=16, IH=16, IW=16)))
    assert pol == L.TILE_G128x160 and r == (L.TILE_256x128, 0)
    # where the policy already uses its large tiles the rule is silent
    assert go(ops.gemm_params(122880, 320, conv(320), X, X, 320, geom=ops.Geom(OH=40, OW=64, IH=40, IW=64)))[1] is None
    assert go(ops.gemm_params(122880, 960, lin(320), X, X, 960))[1] is None
    # a folded LayerNorm cannot split K; GEGLU takes 128-column tiles
    pol, r = go(ops.gemm_params(3072, 3840, lin(1280), X, X, 3840, rowstat=X, colsum=X))
    assert r is None or r[1] == 0
    pol, r = go(ops.gemm_params(3072, 10240, lin(1280), X, X, 5120, epilogue=L.EPI_GEGLU))
    assert r is None or r[0] in (L.TILE_256x128, L.TILE_P256x128)
    # few rows (a frame-parallel rank's fourth level, 120 rows): 64 x 64 register tiles; the long reduction split 12 ways (measured
    # 34.7 -> 23.9 us), the short-K linear unsplit (20.8 -> 15.0 us)
    pol, r = go(ops.gemm_params(120, 1280, conv(1280), X, X, 1280, geom=ops.Geom(OH=3, OW=5, IH=3, IW=5), ksplit=8, workspace=X))
    assert r == (L.TILE_64x64, 12)
    pol, r = go(ops.gemm_params(120, 1280, lin(1280), X, X, 1280, rowstat=X, colsum=X))
    assert r == (L.TILE_64x64, 0)


def test_gemm_validate_is_the_launch_without_the_device():
    """vmv_gemm_validate: VMV_OK iff vmv_gemm would launch — argument checks, policy, the forced tile's own launcher checks; no GPU
    needed.  A stale tuned entry (a tile that cannot serve the launch) is dropped by the tuner hook with ONE warning and the policy
    stands (it used to surface as VMV_EINVAL on the first replay)."""
    import warnings
    lib = L.load()
    X = 1 << 20
    p = ops.gemm_params(4096, 1280, ops.linear_segs([(X, 1280, 1280)]), X, X, 1280)
    assert lib.vmv_gemm_validate(C.byref(p)) == 0
    p.tile = L.TILE_RS                       # the row-stationary kernel serves K = 320 / 640 only
    assert lib.vmv_gemm_validate(C.byref(p)) == -1
    p.tile, p.N = L.TILE_AUTO, 1281
    assert lib.vmv_gemm_validate(C.byref(p)) == -1
    p.N = 1280
    p.tile, p.ksplit, p.workspace = L.TILE_X256x320, 64, X       # an impossible split of the wide tile
    assert lib.vmv_gemm_validate(C.byref(p)) != 0

    class Owner:
        device = "cpu"
    q = ops.gemm_params(4096, 1280, ops.linear_segs([(X, 1280, 1280)]), X, X, 1280)
    sig = ops.gemm_signature(q)
    old, ops._TUNED = ops._TUNED, {sig: dict(tile=L.TILE_RS, ksplit=0)}
    old_w, ops._STALE_WARNED = ops._STALE_WARNED, False
    try:
        o = Owner()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ops.make_tuner(o)(q)
            q2 = ops.gemm_params(4096, 1280, ops.linear_segs([(X, 1280, 1280)]), X, X, 1280)
            ops.make_tuner(o)(q2)
        assert q.tile == L.TILE_AUTO and q2.tile == L.TILE_AUTO and o.n_stale == 2 and getattr(o, "n_tuned", 0) == 0
        assert len([x for x in w if "stale" in str(x.message)]) == 1
    finally:
        ops._TUNED, ops._STALE_WARNED = old, old_w
