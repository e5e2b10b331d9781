"""Thin host-side launch layer over the C ABI (``include/vmv.h``).

``Stream`` either launches immediately on the current torch stream (eager) or records the launch into a
``VmvPlan`` that is replayed with one C call (see ``DESIGN.md`` §5).  Tensors are only used as device-memory
handles (``data_ptr()``); all arithmetic happens in the HIP kernels.  Nothing here falls back to PyTorch math.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

TAPS3x3 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
GN_REC = 8      # int64 per (stat group, channel group) record of VmvGroupNormParams.totals (include/vmv.h)
GN_NREP = 8     # replicas of every record (chunk c adds into replica c % 8)
GN_TOT = 32 * GN_NREP * GN_REC      # int64 per stat group


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


class Seg:
    """One K-segment of the implicit GEMM: ``k`` channels of each (gathered) row of ``src``."""
    __slots__ = ("src", "ld", "k", "mode", "d0", "d1")

    def __init__(self, src, ld, k, mode=L.SEG_LINEAR, d0=0, d1=0):
        self.src, self.ld, self.k, self.mode, self.d0, self.d1 = src, int(ld), int(k), mode, d0, d1


def conv3x3_segs(sources: Sequence[Tuple[torch.Tensor, int, int]], shift: int = 0) -> List[Seg]:
    """sources = [(rows tensor, ld, channels)] (channel-concatenated).  Order: tap-major, then source — the
    order ``pack_conv3x3`` lays the weights out in.  ``shift=1`` moves the taps to (0..2, 0..2): the VAE encoder's
    pad-(0,1,0,1)-then-valid stride-2 convolution (autoencoder.py:475-479)."""
    return [Seg(t, ld, k, L.SEG_SPATIAL, dy + shift, dx + shift) for (dy, dx) in TAPS3x3 for (t, ld, k) in sources]


def temporal_segs(src, ld, k) -> List[Seg]:
    return [Seg(src, ld, k, L.SEG_TEMPORAL, dt, 0) for dt in (-1, 0, 1)]


def linear_segs(sources) -> List[Seg]:
    return [Seg(t, ld, k, L.SEG_LINEAR) for (t, ld, k) in sources]


class Geom:
    """Row <-> pixel geometry for SPATIAL / TEMPORAL segments."""

    def __init__(self, OH=0, OW=0, IH=0, IW=0, stride=1, ups=0, F=0, P=0):
        self.OH, self.OW, self.IH, self.IW, self.stride, self.ups, self.F, self.P = OH, OW, IH, IW, stride, ups, F, P


def gemm_params(M, N, segs: Sequence[Seg], W, out, ldo, bias=None, rowvec=None, rowvec_div=1, rowvec_ld=0,
                residual=None, ldr=0, epilogue=L.EPI_NONE, act=L.ACT_NONE, out_fp32=False, geom: Optional[Geom] = None,
                ksplit=0, workspace=None, tile=L.TILE_AUTO, res_scale=0.0, rowstat=None, colsum=None, ln_eps=0.0, wgroup_rows=0, wgroup_stride=0,
                gn_table=None, gn_rows_per_stat=0, gn_silu=False, epi_scale=0.0) -> L.GemmParams:
    p = L.GemmParams()
    p.M, p.N, p.nseg = int(M), int(N), len(segs)
    if len(segs) > L.VMV_MAX_SEGS:
        raise ValueError("too many GEMM segments")
    kt = 0
    for i, s in enumerate(segs):
        g = p.seg[i]
        g.src, g.ld, g.k, g.mode, g.d0, g.d1 = _ptr(s.src), s.ld, s.k, s.mode, s.d0, s.d1
        kt += s.k
    p.ktot = kt
    p.W, p.bias, p.rowvec = _ptr(W), _ptr(bias), _ptr(rowvec)
    p.rowvec_div, p.rowvec_ld = int(rowvec_div), int(rowvec_ld)
    p.residual, p.ldr = _ptr(residual), int(ldr)
    p.epilogue, p.act, p.out_fp32 = epilogue, act, 1 if out_fp32 else 0
    p.out, p.ldo = _ptr(out), int(ldo)
    g = geom or Geom()
    p.OH, p.OW, p.IH, p.IW, p.stride, p.ups, p.F, p.P = g.OH, g.OW, g.IH, g.IW, g.stride, g.ups, g.F, g.P
    p.ksplit, p.workspace, p.tile, p.res_scale = int(ksplit), _ptr(workspace), tile, float(res_scale)
    p.rowstat, p.colsum, p.ln_eps = _ptr(rowstat), _ptr(colsum), float(ln_eps)
    p.wgroup_rows, p.wgroup_stride = int(wgroup_rows), int(wgroup_stride)
    p.gn_table, p.gn_rows_per_stat = _ptr(gn_table), int(gn_rows_per_stat)       # GroupNorm folded into the A rows (vmv.h; gemm_rs / gemm_tfr)
    p.gn_silu = 1 if gn_silu else 0
    p.epi_scale = float(epi_scale)                         # VMV_EPI_TATTN: softmax scale of the fused temporal attention (gemm_tqa.hip)
    return p


def gn_params(x, ld, C0, rows, rows_per_stat, partial, gamma, beta, eps, silu, y, ldy, x1=None, ld1=0, C1=0,
              chunk_rows=None, fold_ranks=0, totals=None, totals_clear=None, clear_count=0) -> L.GroupNormParams:
    """totals / totals_clear: int64 two-limb fixed-point stat-group accumulators, GN_TOT int64 per stat group (include/vmv.h):
    `totals` must be zero when the statistics pass starts; the apply pass zeroes `clear_count` entries of `totals_clear` (the
    next norm's accumulators)."""
    p = L.GroupNormParams()
    p.x, p.x1, p.ld, p.ld1, p.C0, p.C1 = _ptr(x), _ptr(x1), int(ld), int(ld1), int(C0), int(C1)
    p.rows, p.rows_per_stat = int(rows), int(rows_per_stat)
    p.chunk_rows = int(chunk_rows or gn_chunk_rows(rows_per_stat, C0 + C1))
    p.partial, p.gamma, p.beta = _ptr(partial), _ptr(gamma), _ptr(beta)
    p.eps, p.silu, p.y, p.ldy, p.fold_ranks = float(eps), 1 if silu else 0, _ptr(y), int(ldy), int(fold_ranks)
    p.totals, p.totals_clear, p.clear_count = _ptr(totals), _ptr(totals_clear), int(clear_count)
    return p


def gn_chunk_rows(rows_per_stat: int, C: int) -> int:
    """Rows per partial-sum block: ~64 KB of input per block, at most 256 chunks per stat group."""
    kb, cap = int(os.environ.get("VMV_GN_CHUNK_KB", "64")), int(os.environ.get("VMV_GN_MAXCHUNKS", "256"))     # (A/B experiments)
    r = max(1, kb * 1024 // (C * 2))
    r = max(r, (rows_per_stat + cap - 1) // cap)
    return min(r, rows_per_stat)


def gemm_signature(p: "L.GemmParams") -> str:
    """Shape / feature key of an implicit-GEMM launch for the tuned (tile, split-K) table (videomv_amd/tuned_gemm.json): every
    property the kernel choice can depend on and no pointer.  Runs of equal (mode, k) segments are run-length coded."""
    runs, i = [], 0
    while i < p.nseg:
        j = i
        while j < p.nseg and (p.seg[j].mode, p.seg[j].k) == (p.seg[i].mode, p.seg[i].k):
            j += 1
        runs.append(f"{p.seg[i].mode}:{p.seg[i].k}*{j - i}")
        i = j
    flags = (f"e{p.epilogue}a{p.act}f{p.out_fp32}r{int(bool(p.residual))}v{int(bool(p.rowvec))}:{p.rowvec_div if p.rowvec else 0}"
             f"s{int(bool(p.rowstat))}c{int(bool(p.colsum))}l{int(p.ln_eps > 0)}g{int(bool(p.gn_table)) + 2 * int(bool(p.gn_silu))}w{p.wgroup_rows}")
    geo = f"{p.OH}x{p.OW}<{p.IH}x{p.IW}s{p.stride}u{p.ups}F{p.F}P{p.P}"
    return f"{p.M}x{p.N}x{p.ktot};{','.join(runs)};{flags};{geo}"


_TUNED = None
_STALE_WARNED = False


def tuned_table() -> dict:
    """signature -> {"tile": id, "ksplit": n}: per-shape kernel choices measured on an MI355X by tools/autotune_gemm.py (the policy in
    csrc/gemm.hip was fitted to the 1-GPU shapes, M = 122 880 ...; the table carries what measurement found better — mostly the
    small-M shapes of a frame-parallel rank).  VMV_TUNED=0 ignores it; VMV_TUNED_FILE points at another one."""
    global _TUNED
    if _TUNED is None:
        _TUNED = {}
        if os.environ.get("VMV_TUNED", "1") != "0":
            import json
            path = os.environ.get("VMV_TUNED_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemm.json")
            if os.path.exists(path):
                with open(path) as f:
                    tab = json.load(f)
                # (measured with the fp16 library; the bf16 build runs the same kernels on the same shapes: one table serves both
                #  unless a "bf16" section says otherwise)
                _TUNED = dict(tab.get(L.elem_name()) or tab.get("fp16") or {})
            # the per-user cache VMV_AUTOTUNE=1 writes (autotune.py) goes over the packaged table; read only when autotuning is on or
            # VMV_TUNED_CACHE names a file explicitly, so that a default run depends on nothing outside the package
            if os.environ.get("VMV_AUTOTUNE", "0") == "1" or os.environ.get("VMV_TUNED_CACHE"):
                from .autotune import cache_path
                try:
                    with open(cache_path()) as f:
                        _TUNED.update(json.load(f).get(L.elem_name(), {}))
                except (OSError, ValueError):
                    pass
    return _TUNED


def fill_rule(p: "L.GemmParams", policy_tile: int):
    """The default tile RULE on top of csrc/gemm.hip's built-in policy (round 5; VMV_TILE_RULES=0 turns it off).  The built-in policy was
    fitted to the 24x40x64 plan (M = 122 880 / 30 720 / 7 680 / 1 920); this is what the MEASURED table (tools/autotune_gemm.py, round 4)
    says about every other grid, written as rules instead of 255 shape-specific entries (tools/experiments/rule_eval.py scores it against
    that table; tools/prune_tuned_table.py removes the entries it reproduces):
      * where the policy falls back to 128-row tiles (ids 7 / 8) because 256-row tiles alone do not reach 240 blocks: 256-row tiles of
        128 / 160 columns with the split-K factor (<= 8, >= 8 chunks per split) that best fills ONE round of the 256 CUs —
        score = fill x padding efficiency - 0.03 per extra split; persistent tiles for short-K linears that need no split;
      * few rows (M <= 512: the fourth level of a frame-parallel rank, the embedding GEMMs): 64 x 64 register tiles — a 256- or 128-row
        tile wastes most of its rows — with more K splits than the policy's cap of 8 on the long reductions (12 / 16), 256 x 128 tiles
        from 240 rows on when the reduction is >= 100 chunks;
      * short-K linears on a few thousand rows (200-1024 tiles of 64 x 64): 64 x 64 register tiles.
    -> (tile, ksplit) or None (keep the policy's choice).  The caller validates the choice with vmv_gemm_validate before forcing it."""
    import math
    M, N = p.M, p.N
    steps = sum((p.seg[i].k + 63) // 64 for i in range(p.nseg))
    geglu = p.epilogue == L.EPI_GEGLU
    lin = all(p.seg[i].mode == L.SEG_LINEAR for i in range(p.nseg))
    ks_ok = not (p.rowstat or p.ln_eps > 0 or p.gn_table or p.colsum)
    small_pol = policy_tile in (L.TILE_G128x128, L.TILE_G128x160, L.TILE_P256x128, L.TILE_P256x160, L.TILE_128x160, L.TILE_128x128)
    if M <= 512 and small_pol and not geglu and not p.out_fp32 and N >= 256:
        if lin and steps <= 24:
            if M > 256 and N > 1280:        # (measured: the persistent tile is within 7 % there)
                return None
            ks = 0
            if ks_ok and p.ksplit > 1:
                ks = 4 if M <= 256 else 3
            return L.TILE_64x64, ks
        if ks_ok and steps >= 56 and (M <= 256 or steps >= 100):
            ks = 6 if steps < 70 else (12 if steps < 210 or M > 256 else 16)
            if M > 128 and steps < 100:
                ks = min(ks, 8)
            tile = L.TILE_64x64 if (M <= 128 or steps < 100) else L.TILE_256x128
            return tile, ks
    if policy_tile not in (L.TILE_G128x128, L.TILE_G128x160):
        return None
    best = None
    for bn in (128, 160):
        if bn == 160 and (N % 160 or geglu):
            continue
        tm, tn = math.ceil(M / 256), math.ceil(N / bn)
        pad = (M * N) / (tm * 256 * tn * bn)
        for ks in (1, 2, 3, 4, 6, 8):
            if ks > 1 and (not ks_ok or steps // ks < 8):
                continue
            blocks = tm * tn * ks
            score = blocks / (256 * math.ceil(blocks / 256)) * pad - 0.03 * (ks - 1) + (0.01 if bn == 128 else 0.0)
            if best is None or score > best[0] + 1e-9:
                best = (score, bn, ks)
    if best and best[0] >= 0.70:
        _, bn, ks = best
        if ks == 1 and lin and steps <= 24:
            return (L.TILE_P256x128 if bn == 128 else L.TILE_P256x160), 0
        return (L.TILE_256x128 if bn == 128 else L.TILE_256x160), (ks if ks > 1 else 0)
    if lin and steps <= 24 and M <= 4096 and not geglu and 200 <= math.ceil(M / 64) * math.ceil(N / 64) <= 1024:
        return L.TILE_64x64, 0
    return None


def make_tuner(owner):
    """ops.Stream hook (Stream.tuner) of an engine: applies the measured (tile, split-K) choice of tuned_gemm.json to a GEMM about to be
    recorded — auto tiles only.  The forced choice is VALIDATED here (vmv_gemm_validate: the launch's whole host side without the
    device); an entry the library would refuse — a stale VMV_TUNED_CACHE / VMV_TUNED_FILE, a tile whose eligibility changed — is
    dropped with one warning and the built-in policy stands, instead of VMV_EINVAL on the first replay.  `owner`
    provides .device (or .dev) and keeps the split-K slab (._splitk) and the hit count (.n_tuned)."""
    def tune(p):
        if p.tile != L.TILE_AUTO or p.wgroup_rows or p.epilogue == L.EPI_TATTN:      # (the fused q|k|v + attention has ONE kernel)
            return
        ent = tuned_table().get(gemm_signature(p))
        from_rule = False
        if not ent and os.environ.get("VMV_TILE_RULES", "1") != "0":
            pol = L.load().vmv_gemm_pick_tile(C.byref(p))
            r = fill_rule(p, pol)
            if r is not None and (int(r[0]), int(r[1])) != (pol, p.ksplit if p.ksplit > 1 else 0):
                ent, from_rule = dict(tile=r[0], ksplit=r[1]), True
        if not ent:
            return
        ks = int(ent.get("ksplit", 0))
        keep = (p.tile, p.ksplit, p.workspace)
        if ks > 1:
            if getattr(owner, "_splitk", None) is None:
                owner._splitk = SplitK(owner.device if hasattr(owner, "device") else owner.dev, cap=8)
            p.workspace = owner._splitk.workspace(ks * p.M * p.N * 4).data_ptr()
        p.ksplit = ks if ks > 1 else 0
        p.tile = int(ent.get("tile", 0))
        rc = L.load().vmv_gemm_validate(C.byref(p))
        if rc != 0:
            p.tile, p.ksplit, p.workspace = keep
            if from_rule:           # (a rule's suggestion the library cannot serve for this launch: the policy stands, silently)
                return
            global _STALE_WARNED
            if not _STALE_WARNED:
                _STALE_WARNED = True
                import warnings
                warnings.warn(f"tuned GEMM entry {ent} for {gemm_signature(p)} is refused by the library (rc {rc}): stale tile table / "
                              f"cache — ignored, the built-in policy is used (further stale entries are dropped silently)")
            owner.n_stale = getattr(owner, "n_stale", 0) + 1
            return
        if from_rule:
            owner.n_ruled = getattr(owner, "n_ruled", 0) + 1
        else:
            owner.n_tuned = getattr(owner, "n_tuned", 0) + 1
    return tune


class SplitK:
    """Split-K policy + workspace for the small-M implicit GEMMs (a tile grid that cannot fill the 256 CUs with a long
    reduction): K is cut into `ks` slices (fp32 slabs in the workspace, deterministic reduce + epilogue pass, vmv.h).
    One instance per engine: the slab is shared by all of its launches (single stream: launches are ordered)."""

    def __init__(self, device, cap=8):
        self.device, self.cap, self.ws, self.keep = device, cap, None, []

    def pick(self, M, N, segs):
        steps = sum((s.k + 63) // 64 for s in segs)
        bn = 160 if N % 160 == 0 else 128
        tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
        if tiles >= 192 or steps < 16:
            return 0, None
        # the split-K shapes run on the 4-wave LDS-DMA kernel, two blocks per CU: aim at ~2 x 256 blocks
        ks = min(self.cap, max(1, (480 + tiles // 2) // tiles), steps // 8)
        if ks < 2:
            return 0, None
        need = ks * M * N * 4
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
            self.keep.append(self.ws)      # earlier recorded launches keep pointing at the old slab
        return ks, self.ws

    def workspace(self, need_bytes: int):
        """The shared fp32 slab, grown to `need_bytes` (a tuned split-K factor the heuristic did not pick)."""
        if self.ws is None or self.ws.numel() < need_bytes:
            self.ws = torch.empty(max(int(need_bytes), 1 << 20), dtype=torch.uint8, device=self.device)
            self.keep.append(self.ws)
        return self.ws


def gn_fused_cols(rows_per_stat: int, C: int) -> int:
    """Channels per block of the one-launch GroupNorm (vmv_groupnorm_fused), or 0 when a stat group does not fit on chip:
    the narrowest slab of whole groups whose rows are still >= 128 B (else >= 64 B) contiguous — most blocks, least LDS."""
    if os.environ.get("VMV_GN_FUSED", "1") == "0":
        return 0
    cpg = C // 32
    for min_bytes in (128, 64):
        G = 1
        while G <= 32:
            cols = G * cpg
            if cols % 8 == 0 and cols * 2 >= min_bytes and C % cols == 0:
                if rows_per_stat * cols * 2 <= L.GN_FUSED_BYTES:
                    return cols
                break                       # wider slabs only need more LDS
            G *= 2
    return 0


def gn_partial_floats(rows, rows_per_stat, C, chunk_rows=None) -> int:
    cr = chunk_rows or gn_chunk_rows(rows_per_stat, C)
    nchunk = (rows_per_stat + cr - 1) // cr
    return (rows // rows_per_stat) * nchunk * 64


def ln_params(x, ldx, y, ldy, gamma, beta, rows, Cc, eps=1e-5, stats_out=None) -> L.LayerNormParams:
    """stats_out: fp32 [rows][2] — write (mean, rstd) per row instead of y (LayerNorm folded into its consumer GEMM)."""
    p = L.LayerNormParams()
    p.x, p.ldx, p.y, p.ldy, p.gamma, p.beta = _ptr(x), int(ldx), _ptr(y), int(ldy), _ptr(gamma), _ptr(beta)
    p.rows, p.C, p.eps = int(rows), int(Cc), float(eps)
    p.stats_out = _ptr(stats_out)
    return p


def ff_params(M, Cc, x, ldx, w1, b1, w2, b2, out, ldo, residual=None, ldr=0, ln_eps=0.0) -> L.FfParams:
    """Fused FeedForward (vmv_ff_fused): w1 / b1 GEGLU-interleaved (+ LayerNorm folded when ln_eps > 0), w2 = packing.ff_down_permute."""
    p = L.FfParams()
    p.M, p.C, p.x, p.ldx = int(M), int(Cc), _ptr(x), int(ldx)
    p.w1, p.b1, p.w2, p.b2 = _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2)
    p.residual, p.ldr, p.ln_eps, p.out, p.ldo = _ptr(residual), int(ldr), float(ln_eps), _ptr(out), int(ldo)
    return p


def seq_map(s_outer, s_inner, s_row, inner=1) -> L.SeqMap:
    m = L.SeqMap()
    m.s_outer, m.s_inner, m.s_row, m.inner = int(s_outer), int(s_inner), int(s_row), int(inner)
    return m


def attn_params(q, k, v, o, qm, km, vm, om, n_outer, heads, Nq, Nk, scale, kv_div=1, head_dim=64, causal=False) -> L.AttnParams:
    p = L.AttnParams()
    p.q, p.k, p.v, p.o = _ptr(q), _ptr(k), _ptr(v), _ptr(o)
    p.qm, p.km, p.vm, p.om = qm, km, vm, om
    p.n_outer, p.kv_div, p.heads, p.Nq, p.Nk, p.scale = int(n_outer), int(kv_div), int(heads), int(Nq), int(Nk), float(scale)
    p.head_dim, p.causal = int(head_dim), 1 if causal else 0
    return p


def softmax_params(s, lds, p_out, ldp, rows, n, scale) -> L.SoftmaxParams:
    p = L.SoftmaxParams()
    p.s, p.lds, p.p, p.ldp, p.rows, p.n, p.scale = _ptr(s), int(lds), _ptr(p_out), int(ldp), int(rows), int(n), float(scale)
    return p


def copy_params(src, dst, n0, n1, n2, inner16, ss0, ss1, ss2=0) -> L.CopyParams:
    p = L.CopyParams()
    p.src, p.dst, p.n0, p.n1, p.n2, p.inner16 = _ptr(src), _ptr(dst), int(n0), int(n1), int(n2), int(inner16)
    p.ss0, p.ss1, p.ss2 = int(ss0), int(ss1), int(ss2)
    return p


def comm_params(comm_handle, kind, send, recv, nbytes) -> L.CommParams:
    """One collective of the frame-sharded sampler as a plan op (vmv.h VMV_OP_COMM): `comm_handle` = VmvComm* from comm.py
    (RCCL or simulated), `nbytes` = the per-rank chunk."""
    p = L.CommParams()
    p.comm, p.kind, p.send, p.recv, p.bytes = comm_handle, int(kind), _ptr(send), _ptr(recv), int(nbytes)
    return p


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Stream:
    """Eager launcher / plan recorder."""

    def __init__(self, record: bool = False):
        self.lib = L.load()
        self.record = record
        self.plan = self.lib.vmv_plan_create() if record else None
        self.keep = []          # tensors referenced by recorded argument blocks
        self.nops = 0
        self.labels = []        # label per recorded op (profiling / debugging)
        self.recorded = []      # (op code, params struct) mirror of the C plan (debugging / tests)
        self.graph = None       # VmvGraph*: the whole plan captured as a hipGraph (capture_graph())

    def __del__(self):
        try:
            if self.graph:
                self.lib.vmv_graph_destroy(self.graph)
                self.graph = None
            if self.plan:
                self.lib.vmv_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def _go(self, op, params, fn, label):
        if self.record:
            rc = self.lib.vmv_plan_add(self.plan, op, C.byref(params), C.sizeof(params))
            if rc < 0:
                L.check(rc, f"plan_add {label}")
            self.labels.append(label)
            self.recorded.append((op, params))
            self.nops += 1
        else:
            L.check(fn(C.byref(params), _stream_ptr()), label)

    def gemm(self, params, label="gemm"):
        if getattr(self, "tuner", None) is not None:
            self.tuner(params)
        self._go(L.OP_GEMM, params, self.lib.vmv_gemm, label)

    def groupnorm(self, params, label="gn"):
        """statistics + apply; one fused launch when the stat group fits on chip (gn_fused_cols), else two."""
        cols = 0 if (params.totals or params.fold_ranks > 1) else gn_fused_cols(params.rows_per_stat, params.C0 + params.C1)
        if cols:
            return self.groupnorm_fused(params, cols, label)
        self._go(L.OP_GN_STATS, params, self.lib.vmv_groupnorm_stats, label + ".stats")
        self._go(L.OP_GN_APPLY, params, self.lib.vmv_groupnorm_apply, label + ".apply")

    def groupnorm_fused(self, params, cols, label="gn"):
        """One launch: statistics + apply from an LDS-resident stage (cols from gn_fused_cols)."""
        params.chunk_rows = int(cols)
        fn = lambda pp, st: self.lib.vmv_groupnorm_fused(pp, int(cols), st)
        self._go(L.OP_GN_FUSED, params, fn, label + ".fused")

    def groupnorm_stats(self, params, label="gn"):
        self._go(L.OP_GN_STATS, params, self.lib.vmv_groupnorm_stats, label + ".stats")

    def groupnorm_apply(self, params, label="gn"):
        self._go(L.OP_GN_APPLY, params, self.lib.vmv_groupnorm_apply, label + ".apply")

    def groupnorm_table(self, params, label="gn"):
        """scale / shift table of the norm (params.y = fp32 [nstat][2][C]) for a GEMM that folds it (gemm_params(gn_table=...))."""
        self._go(L.OP_GN_TABLE, params, self.lib.vmv_groupnorm_table, label + ".table")

    def copy(self, params, label="copy"):
        self._go(L.OP_COPY, params, self.lib.vmv_permute_copy, label)

    def ff(self, params, label="ff"):
        self._go(L.OP_FF, params, self.lib.vmv_ff_fused, label)

    def comm(self, params, label="comm"):
        self._go(L.OP_COMM, params, self.lib.vmv_comm_run, label)

    def layernorm(self, params, label="ln"):
        self._go(L.OP_LAYERNORM, params, self.lib.vmv_layernorm, label)

    def attention(self, params, label="attn"):
        self._go(L.OP_ATTENTION, params, self.lib.vmv_attention, label)

    def softmax(self, params, label="softmax"):
        self._go(L.OP_SOFTMAX, params, self.lib.vmv_softmax_rows, label)

    def capture_graph(self):
        """Capture ONE replay of the whole plan into a hipGraph (vmv_plan_capture); run() then launches the graph.  The plan must
        have run eagerly once before (first-use kernel attributes).  The legacy default stream cannot be captured: when that is
        torch's current stream the graph lives on a side stream of its own, fenced against the current stream at every launch."""
        assert self.record and self.nops
        if self.graph:
            self.lib.vmv_graph_destroy(self.graph)
            self.graph = None
        cur = torch.cuda.current_stream()
        self._gstream = torch.cuda.Stream(device=cur.device) if cur.cuda_stream == 0 else None
        if self._gstream is not None:
            self._gstream.wait_stream(cur)
        g = self.lib.vmv_plan_capture(self.plan, C.c_void_p((self._gstream or cur).cuda_stream))
        if not g:
            raise L.VmvError("vmv_plan_capture failed")
        self.graph = g
        return self.lib.vmv_graph_nodes(g)

    def run_local(self):
        """Replay every recorded op EXCEPT the collectives (VMV_OP_COMM): fills the buffers with finite, realistic contents without
        needing the peers (autotune.tune_plan on a multi-rank plan: a per-rank replay with collectives would hang the ranks that
        have nothing left to tune)."""
        assert self.record
        i, n = 0, len(self.recorded)
        while i < n:
            if self.recorded[i][0] == L.OP_COMM:
                i += 1
                continue
            j = i
            while j < n and self.recorded[j][0] != L.OP_COMM:
                j += 1
            L.check(self.lib.vmv_plan_run_range(self.plan, i, j, _stream_ptr()), "plan_run_range")
            i = j

    def run(self, first=0, last=None):
        """Replay the recorded plan on the current torch stream (as ONE graph launch after capture_graph())."""
        assert self.record
        if last is None and first == 0 and self.graph:
            gs = getattr(self, "_gstream", None)
            if gs is None:
                L.check(self.lib.vmv_graph_launch(self.graph, _stream_ptr()), "graph_launch")
            else:
                cur = torch.cuda.current_stream()
                gs.wait_stream(cur)
                L.check(self.lib.vmv_graph_launch(self.graph, C.c_void_p(gs.cuda_stream)), "graph_launch")
                cur.wait_stream(gs)
        elif last is None:
            L.check(self.lib.vmv_plan_run(self.plan, _stream_ptr()), "plan_run")
        else:
            L.check(self.lib.vmv_plan_run_range(self.plan, first, last, _stream_ptr()), "plan_run_range")


# ----------------------------------------------------------------------------------- direct (non-plan) glue
def latent_to_rows(x: torch.Tensor, rows: torch.Tensor, Cpad: int, nrep: int):
    nb, Cc, F_, H, W = x.shape
    lib = L.load()
    L.check(lib.vmv_latent_to_rows(x.data_ptr(), rows.data_ptr(), nb, Cc, F_, H, W, Cpad, nrep, _stream_ptr()),
            "latent_to_rows")


def latent_to_rows_keep(x: torch.Tensor, rows: torch.Tensor, ld: int, nrep: int):
    """Write only channels [0, C) of each ld-wide row (the others keep their contents)."""
    nb, Cc, F_, H, W = x.shape
    L.check(L.load().vmv_latent_to_rows_keep(x.data_ptr(), rows.data_ptr(), nb, Cc, F_, H, W, ld, nrep, _stream_ptr()),
            "latent_to_rows_keep")


def i2v_temporal_adapter(inp, ld_in, out_ptr, ld_out, w, F_, HW, nrep, scale):
    L.check(L.load().vmv_i2v_temporal_adapter(_ptr(inp), ld_in, _ptr(out_ptr), ld_out, w.data_ptr(), F_, HW, nrep,
                                              float(scale), _stream_ptr()), "i2v_temporal_adapter")


def adaptive_avgpool_rows(inp, ld, out, ldo, n, Cc, IH, IW, OH, OW):
    L.check(L.load().vmv_adaptive_avgpool_rows(_ptr(inp), ld, _ptr(out), ldo, n, Cc, IH, IW, OH, OW, _stream_ptr()),
            "adaptive_avgpool_rows")


def rows_to_nchw(rows: torch.Tensor, ld: int, out: torch.Tensor):
    n, Cc, H, W = out.shape
    lib = L.load()
    L.check(lib.vmv_rows_to_nchw(rows.data_ptr(), 1 if rows.dtype == torch.float32 else 0, ld, out.data_ptr(), n, Cc,
                                 H * W, _stream_ptr()), "rows_to_nchw")


def cfg_ddim_step(eps_rows, ld, xt, guide_scale, c_recip, c_recipm1, c_sqrt_ac, c_sqrt_1mac, a_prev, v_pred=False,
                  x0_out=None, clamp=0.0, sigma=0.0, noise=None):
    _, Cc, F_, H, W = xt.shape
    p = L.DdimParams()
    p.eps_rows, p.ld, p.C, p.F, p.HW = eps_rows.data_ptr(), ld, Cc, F_, H * W
    p.guide_scale, p.c_recip, p.c_recipm1 = guide_scale, c_recip, c_recipm1
    p.c_sqrt_ac, p.c_sqrt_1mac, p.a_prev, p.v_pred = c_sqrt_ac, c_sqrt_1mac, a_prev, 1 if v_pred else 0
    p.xt, p.x0_out = xt.data_ptr(), _ptr(x0_out)
    p.clamp, p.sigma, p.noise = float(clamp or 0.0), float(sigma), _ptr(noise)
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.numel() == xt.numel()
    L.check(L.load().vmv_cfg_ddim_step(C.byref(p), _stream_ptr()), "cfg_ddim_step")


def posterior_sample(moments_rows, ld, noise, z, scale):
    n, zc, H, W = z.shape
    L.check(L.load().vmv_posterior_sample(moments_rows.data_ptr(), ld, noise.data_ptr(), z.data_ptr(), n, zc, H * W,
                                          float(scale), _stream_ptr()), "posterior_sample")


def lgm_x0_views(eps_rows, ld, branch, xt, idx4, c_recip, c_recipm1, inv_scale, out):
    _, Cc, F_, H, W = xt.shape
    ia = (C.c_int32 * 4)(*[int(i) for i in idx4])
    L.check(L.load().vmv_lgm_x0_views(eps_rows.data_ptr(), int(ld), int(branch), xt.data_ptr(), Cc, F_, H * W, ia,
                                      float(c_recip), float(c_recipm1), float(inv_scale), out.data_ptr(), _stream_ptr()),
            "lgm_x0_views")


def lgm_pack_input(decoded, rays, out):
    n, _, H, W = decoded.shape
    L.check(L.load().vmv_lgm_pack_input(decoded.data_ptr(), rays.data_ptr(), out.data_ptr(), n, H * W, _stream_ptr()),
            "lgm_pack_input")


def lgm_render_to_vae(images, out):
    """rendered [n,3,S_in,S_in] in [0,1] -> nearest-resampled [n,3,S,S] in [-1,1] (any size ratio, as F.interpolate)."""
    n, _, S, _ = out.shape
    L.check(L.load().vmv_lgm_render_to_vae(images.data_ptr(), out.data_ptr(), n, images.shape[-1], S, _stream_ptr()),
            "lgm_render_to_vae")


def ddim_x0_step(x0_cond, x0_uncond, xt, guide, c_recip, c_recipm1, a_prev, clamp=None, sigma=0.0, noise=None):
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.numel() == xt.numel()
    L.check(L.load().vmv_ddim_x0_step(x0_cond.data_ptr(), x0_uncond.data_ptr(), xt.data_ptr(), xt.numel(), float(guide),
                                      float(c_recip), float(c_recipm1), float(a_prev), float(clamp or 0.0), float(sigma), _ptr(noise),
                                      _stream_ptr()), "ddim_x0_step")


def gaussian_activation(raw, ld, out, n, workspace):
    L.check(L.load().vmv_gaussian_activation(raw.data_ptr(), int(ld), out.data_ptr(), int(n), workspace.data_ptr(),
                                             _stream_ptr()), "gaussian_activation")


def emb_combine_silu(temb, cam, out, rows, Cc, rows_per_t, cam_rows):
    L.check(L.load().vmv_emb_combine_silu(temb.data_ptr(), _ptr(cam), out.data_ptr(), rows, Cc, rows_per_t, cam_rows,
                                          _stream_ptr()), "emb_combine_silu")


def sinusoidal(t, out, n, dim):
    L.check(L.load().vmv_sinusoidal(t.data_ptr(), out.data_ptr(), n, dim, _stream_ptr()), "sinusoidal")
