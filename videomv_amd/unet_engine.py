"""UNet forward as a recorded plan of HIP launches (host side of the hot path).

``UNetEngine`` turns one ``UNetSD_T2VBase`` forward at a fixed shape (B branches x F frames x H x W latent,
L context tokens) into a static sequence of C-ABI launches over pooled device buffers and replays it.
Activations are channels-last 16-bit (fp16 | bf16: _lib.elem()) "rows" ``[B*F*H*W, C]`` end-to-end: the reference's
``(b f) c h w`` / ``b (hw) c`` / ``(b hw) f c`` views (unet_t2v.py:348, util.py:362, :1054-1062) are index
maps over that one buffer, so none of its ~150 ``rearrange(...).contiguous()`` copies exist here.

Block structure and parameter names follow the reference constructor (unet_t2v.py:160-265); the op sequence
per block follows ``ResBlock._forward`` (util.py:703-730), ``TemporalConvBlock_v2.forward`` (:1381-1392),
``SpatialTransformer.forward`` (:354-373), ``TemporalTransformer.forward`` (:1043-1089),
``BasicTransformerBlock.forward`` (:536-540) and ``MemoryEfficientCrossAttention.forward`` (:230-268).
"""
import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops
from . import packing as P



# --------------------------------------------------------------------------------------------- architecture
def block_plan(cfg):
    """(input_blocks, middle_block, output_blocks) as lists of (kind, prefix, meta) — the constructor's
    loop structure (unet_t2v.py:160-258), including the hard-coded decoder context_dim=1024 (:226) and the
    init temporal transformer built with ``num_heads`` heads (:171)."""
    dim = cfg["dim"]
    dim_mult = list(cfg["dim_mult"])
    nrb = cfg["num_res_blocks"]
    hd = cfg["head_dim"]
    attn_scales = list(cfg["attn_scales"])
    enc_dims = [dim * u for u in [1] + dim_mult]
    dec_dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
    shortcut, scale = [], 1.0
    inp = [[("conv_in", "input_blocks.0.0", dict(cin=cfg["in_dim"], cout=dim)),
            ("tt", "input_blocks.0.1", dict(c=dim, heads=cfg["num_heads"], dh=hd))]]
    shortcut.append(dim)
    idx, out_dim = 1, dim
    for i, (in_dim, out_dim) in enumerate(zip(enc_dims[:-1], enc_dims[1:])):
        for j in range(nrb):
            p = f"input_blocks.{idx}"
            blk = [("res", f"{p}.0", dict(cin=in_dim, cout=out_dim))]
            if scale in attn_scales:
                blk.append(("st", f"{p}.1", dict(c=out_dim, heads=out_dim // hd, dh=hd, ctx=cfg["context_dim"])))
                blk.append(("tt", f"{p}.2", dict(c=out_dim, heads=out_dim // hd, dh=hd)))
            in_dim = out_dim
            inp.append(blk)
            shortcut.append(out_dim)
            idx += 1
            if i != len(dim_mult) - 1 and j == nrb - 1:
                inp.append([("down", f"input_blocks.{idx}", dict(c=out_dim))])
                shortcut.append(out_dim)
                scale /= 2.0
                idx += 1
    mid = [("res", "middle_block.0", dict(cin=out_dim, cout=out_dim)),
           ("st", "middle_block.1", dict(c=out_dim, heads=out_dim // hd, dh=hd, ctx=cfg["context_dim"])),
           ("tt", "middle_block.2", dict(c=out_dim, heads=out_dim // hd, dh=hd)),
           ("res", "middle_block.3", dict(cin=out_dim, cout=out_dim))]
    outb, idx = [], 0
    for i, (in_dim, out_dim) in enumerate(zip(dec_dims[:-1], dec_dims[1:])):
        for j in range(nrb + 1):
            p = f"output_blocks.{idx}"
            sc = shortcut.pop()
            blk = [("res", f"{p}.0", dict(cin=in_dim + sc, cout=out_dim, c_main=in_dim, c_skip=sc))]
            k = 1
            if scale in attn_scales:
                blk.append(("st", f"{p}.{k}", dict(c=out_dim, heads=out_dim // hd, dh=hd, ctx=1024)))
                blk.append(("tt", f"{p}.{k + 1}", dict(c=out_dim, heads=out_dim // hd, dh=hd)))
                k += 2
            in_dim = out_dim
            if i != len(dim_mult) - 1 and j == nrb:
                blk.append(("up", f"{p}.{k}", dict(c=out_dim)))
                scale *= 2.0
            outb.append(blk)
            idx += 1
    return inp, mid, outb


def _attn_shapes(p, c, inner, ctx):
    kv = inner if ctx is None else ctx
    return [(f"{p}.to_q.weight", (inner, c)), (f"{p}.to_k.weight", (inner, kv)), (f"{p}.to_v.weight", (inner, kv)),
            (f"{p}.to_out.0.weight", (c, inner)), (f"{p}.to_out.0.bias", (c,))]


def _tblock_shapes(p, inner, ctx):
    s = _attn_shapes(f"{p}.attn1", inner, inner, None)
    s += [(f"{p}.ff.net.0.proj.weight", (inner * 8, inner)), (f"{p}.ff.net.0.proj.bias", (inner * 8,)),
          (f"{p}.ff.net.2.weight", (inner, inner * 4)), (f"{p}.ff.net.2.bias", (inner,))]
    s += _attn_shapes(f"{p}.attn2", inner, inner, ctx)
    for n in ("norm1", "norm2", "norm3"):
        s += [(f"{p}.{n}.weight", (inner,)), (f"{p}.{n}.bias", (inner,))]
    return s


def param_shapes(cfg) -> "Dict[str, tuple]":
    """Reference state-dict manifest (key -> shape); checked against tests/golden/manifest_unet_t2v_full.json."""
    dim = cfg["dim"]
    E = dim * 4
    s = [("time_embed.0.weight", (E, dim)), ("time_embed.0.bias", (E,)),
         ("time_embed.2.weight", (E, E)), ("time_embed.2.bias", (E,))]
    if cfg.get("use_camera_condition", False):
        s += [("camera_embedding.0.weight", (E, cfg["camera_dim"])), ("camera_embedding.0.bias", (E,)),
              ("camera_embedding.2.weight", (E, E)), ("camera_embedding.2.bias", (E,))]
    if cfg.get("use_fps_condition", False):
        s += [("fps_embedding.0.weight", (E, dim)), ("fps_embedding.0.bias", (E,)),
              ("fps_embedding.2.weight", (E, E)), ("fps_embedding.2.bias", (E,))]
    inp, mid, outb = block_plan(cfg)
    for blk in inp + [mid] + outb:
        for kind, p, m in blk:
            if kind == "conv_in":
                s += [(f"{p}.weight", (m["cout"], m["cin"], 3, 3)), (f"{p}.bias", (m["cout"],))]
            elif kind == "res":
                ci, co = m["cin"], m["cout"]
                s += [(f"{p}.in_layers.0.weight", (ci,)), (f"{p}.in_layers.0.bias", (ci,)),
                      (f"{p}.in_layers.2.weight", (co, ci, 3, 3)), (f"{p}.in_layers.2.bias", (co,)),
                      (f"{p}.emb_layers.1.weight", (co, E)), (f"{p}.emb_layers.1.bias", (co,)),
                      (f"{p}.out_layers.0.weight", (co,)), (f"{p}.out_layers.0.bias", (co,)),
                      (f"{p}.out_layers.3.weight", (co, co, 3, 3)), (f"{p}.out_layers.3.bias", (co,))]
                if ci != co:
                    s += [(f"{p}.skip_connection.weight", (co, ci, 1, 1)), (f"{p}.skip_connection.bias", (co,))]
                for name, ix in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
                    q = f"{p}.temopral_conv.{name}"
                    s += [(f"{q}.0.weight", (co,)), (f"{q}.0.bias", (co,)),
                          (f"{q}.{ix}.weight", (co, co, 3, 1, 1)), (f"{q}.{ix}.bias", (co,))]
            elif kind in ("st", "tt"):
                c = m["c"]
                inner = m["heads"] * m["dh"]
                s += [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,))]
                if kind == "st":
                    s += [(f"{p}.proj_in.weight", (inner, c)), (f"{p}.proj_in.bias", (inner,))]
                    s += _tblock_shapes(f"{p}.transformer_blocks.0", inner, m["ctx"])
                    s += [(f"{p}.proj_out.weight", (inner, c)), (f"{p}.proj_out.bias", (inner,))]
                else:
                    s += [(f"{p}.proj_in.weight", (inner, c, 1)), (f"{p}.proj_in.bias", (inner,))]
                    s += _tblock_shapes(f"{p}.transformer_blocks.0", inner, None)
                    s += [(f"{p}.proj_out.weight", (c, inner, 1)), (f"{p}.proj_out.bias", (c,))]
            elif kind == "down":
                s += [(f"{p}.op.weight", (m["c"], m["c"], 3, 3)), (f"{p}.op.bias", (m["c"],))]
            elif kind == "up":
                s += [(f"{p}.conv.weight", (m["c"], m["c"], 3, 3)), (f"{p}.conv.bias", (m["c"],))]
    s += [("out.0.weight", (dim,)), ("out.0.bias", (dim,)),
          ("out.2.weight", (cfg["out_dim"], dim, 3, 3)), ("out.2.bias", (cfg["out_dim"],))]
    return dict(s)


# --------------------------------------------------------------------------------------------- device memory
class Pool:
    """Size-bucketed reuse of device buffers.  All launches are stream-ordered on ONE stream, so a buffer can
    be handed to a later op as soon as its last consumer has been *recorded*."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[int, List[torch.Tensor]] = {}
        self.all: List[torch.Tensor] = []
        self.bytes = 0

    def get(self, nbytes: int) -> torch.Tensor:
        nbytes = (int(nbytes) + 255) // 256 * 256
        lst = self.free.get(nbytes)
        if lst:
            return lst.pop()
        t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.all.append(t)
        self.bytes += nbytes
        return t

    def put(self, t: torch.Tensor):
        self.free.setdefault(t.numel(), []).append(t)


class Act:
    """A [rows, C] activation living in a pooled buffer."""
    __slots__ = ("buf", "rows", "C", "dtype")

    def __init__(self, buf, rows, C, dtype=None):
        self.buf, self.rows, self.C, self.dtype = buf, rows, C, dtype or L.elem()

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def tensor(self) -> torch.Tensor:
        esz = 4 if self.dtype == torch.float32 else 2
        return self.buf[: self.rows * self.C * esz].view(self.dtype).view(self.rows, self.C)


# --------------------------------------------------------------------------------------------- the engine
class UNetEngine:
    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], B: int, F: int, H: int, W: int, L_ctx: int,
                 device, n_t: int = 1, taps: Optional[dict] = None, comm=None, share_prefix: bool = False,
                 packed: Optional[dict] = None, eps_out: Optional[torch.Tensor] = None):
        """weights: reference-named fp32 state dict (any device).  B = number of batched branches
        (2 = cond + uncond CFG pair sharing x_t), n_t = number of distinct timesteps rows (B // n_t branches
        share each).  F = number of frames of the whole sample.

        comm (``comm.FrameComm``) turns on frame-parallel execution: this rank owns frames
        [rank*F/R, (rank+1)*F/R); ``self.F`` is then the LOCAL frame count and ``self.Fg`` the sample's."""
        # share_prefix: the B = 2 branches are a classifier-free-guidance pair on the SAME x_t, t, camera and fps, so every op
        # before the first cross-attention (init conv + TemporalTransformer, the first ResBlock, the first SpatialTransformer up
        # to and including its self-attention) sees identical inputs in both branches: it is recorded once on
        # B = 1 rows and its three live tensors are replicated (3 copies instead of ~1.1 TFLOP per step at 40x64)
        # (round 6: B = 2 b row blocks of b prompts, PAIR-major [c_0 | u_0 | c_1 | u_1 ...]: the prefix is recorded on b row blocks, one per
        #  prompt, and replicated pairwise — unet_t2v._forward_cfg_rows_batched)
        self.share_prefix = bool(share_prefix) and B >= 2 and B % 2 == 0 and n_t == 1 and comm is None
        self.Bp = B // 2 if self.share_prefix else B        # row blocks while the shared prefix is recorded
        self.comm = comm
        self.R = comm.world if comm is not None else 1
        self.rk = comm.rank if comm is not None else 0
        self.Fg = F
        if F % self.R:
            raise ValueError(f"{F} frames do not split over {self.R} ranks")
        F = F // self.R
        self.cfg, self.B, self.F, self.H, self.W, self.L = cfg, B, F, H, W, L_ctx
        self.B_ctx = B               # number of context (text) branches; self.B drops to 1 while a shared prefix is recorded
        self.device = device
        self.breaks = []            # (op index, callable): collectives issued by Python before that recorded op (no VmvComm handle)
        self.n_comm_ops, self.comm_bytes_in = 0, 0   # collectives recorded INTO the plan (VMV_OP_COMM); bytes received from peers per replay
        self.pool = Pool(device)
        self.S = ops.Stream(record=True)
        self.Sctx = ops.Stream(record=True)      # step-invariant launches: K / V of the text context (run by context_updated())
        self._keepalive = []
        self._splitk = None
        self._gnws = torch.empty(4 << 20, dtype=torch.float32, device=device)
        # pre-folded statistics of the all-frame norms: [nstat][32][2] per rank (+ the gathered [R][nstat][32][2]), tickets
        # two buffers used alternately by consecutive all-frame norms (the apply pass of one clears the other's)
        # (records of ops.GN_REC int64 per (stat group, channel group): two-limb sums of x - pilot, sums of squares, the pilot)
        # (every all-frame norm of an engine has nstat = B stat groups — one per branch: B x GN_TOT int64 per buffer)
        self._gn_tot2 = torch.zeros(2, B * ops.GN_TOT, dtype=torch.int64, device=device)
        self._gn_zero = torch.zeros(2, B * ops.GN_TOT, dtype=torch.int64, device=device)   # source of the plan's own clearing copy
        self._gn_tot_k = 0
        self._gn_tot_all = torch.zeros(B * ops.GN_TOT * self.R, dtype=torch.int64, device=device) if comm is not None else None
        self.taps = taps            # optional dict: prefix -> Act (buffers are then never recycled)
        self.n_t = n_t
        self.dim = cfg["dim"]
        self.E = self.dim * 4
        self.inp, self.mid, self.outb = block_plan(cfg)
        # LayerNorm -> Linear pairs of the transformer blocks run as ONE GEMM on the raw rows + a statistics pass (vmv.h,
        # VmvGemmParams.rowstat): saves writing and re-reading LN(x) (VMV_FOLD_LN=0 keeps the two-kernel form)
        self.fold_ln = os.environ.get("VMV_FOLD_LN", "1") != "0"
        # VMV_LN_INLINE=1: the statistics inside the consumer GEMM's main loop instead of a pass of their own.  Measured
        # (round 2, same box): 99 launches / 1.2 ms of statistics passes go away, but the 48-64 dot2c per chunk
        # cost the short-K GEMMs 7-15 % and the step gets 0.5 ms SLOWER — off by default.
        self.ln_inline = os.environ.get("VMV_LN_INLINE", "0") == "1"
        # VMV_FF_FUSED=1 (default 0): the FeedForward of the C = 320 transformer blocks as ONE launch (csrc/gemm_ff.hip).
        # Opt-in: 383 us against 379 us for the two gemm_rs launches inside a step (51.61 / 51.52 ms, same box) — see the
        # kernel's header for why (one 1-KB LDS fragment per MFMA)
        self.ff_fused = os.environ.get("VMV_FF_FUSED", "0") == "1" and self.fold_ln
        # VMV_GN_FOLD (default 1): the transformers' GroupNorm -> proj_in with the apply pass folded into the GEMM (_gn_folded_proj_in)
        self.gn_fold = os.environ.get("VMV_GN_FOLD", "1") != "0"
        # VMV_TCONV_FOLD (default 1): the temporal conv block's GroupNorm -> SiLU -> (3,1,1) conv with the apply pass folded into the
        # frame-resident kernel's A path (_tconv_folded, csrc/gemm_tfr.hip) wherever that kernel's tiles fill the chip
        self.tconv_fold = os.environ.get("VMV_TCONV_FOLD", "1") != "0"
        # VMV_TQA (default 1): the TemporalTransformers' q | k | v projection + attention over the frames as ONE launch (csrc/gemm_tqa.hip,
        # VMV_EPI_TATTN) wherever the library serves the shape (K = 320: the first level) — q, k, v never reach memory
        self.tqa = os.environ.get("VMV_TQA", "1") != "0"
        # VMV_FP_TEMPORAL (frame-parallel plans): "switch" (default) = the TemporalTransformer runs on the pixel-major shard between two
        # all-to-all layout switches; "kv_gather" = BASELINE's north-star form — frames stay sharded, ONE all-gather of [K | V] before
        # each temporal attention (B = 1 plans, i.e. the branch-pipelined / CFG-parallel modes; 16x the bytes of the switches)
        self.use_graph, self._replays, self.graph_nodes = os.environ.get("VMV_GRAPH", "0") == "1", 0, 0
        self.fp_temporal = os.environ.get("VMV_FP_TEMPORAL", "switch")
        if self.fp_temporal not in ("switch", "kv_gather"):
            raise ValueError("VMV_FP_TEMPORAL must be 'switch' or 'kv_gather'")
        # packed weights are immutable and shape-independent: engines of one model (other B / resolution / frame count, the
        # two branch engines of the pipelined frame-parallel mode) share ONE copy (`packed` = another engine's .packed)
        if packed is not None and packed.get("fold_ln") == self.fold_ln and packed.get("device") == str(device):
            for k, v in packed.items():
                if k not in ("fold_ln", "device"):
                    setattr(self, k, v)
        else:
            self._pack(weights)
        self.packed = dict(fold_ln=self.fold_ln, device=str(device), w=self.w, has_cam=self.has_cam, has_fps=self.has_fps,
                           emb_off=self.emb_off, emb_total=self.emb_total, out_pad=self.out_pad)
        self._eps_out = eps_out
        self.n_tuned, self.n_ruled, self.n_stale = 0, 0, 0      # launches forced by the table / by ops.fill_rule / table entries the library refused
        self.S.tuner = ops.make_tuner(self)      # measured per-shape (tile, split-K) choices: videomv_amd/tuned_gemm.json
        self._static_inputs()
        self._build()
        # VMV_AUTOTUNE=1: a shape the tile table does not cover tunes itself once (autotune.py) and the plan is recorded again with
        # the winners — from a clean slate: new streams, pool and accumulators (the packed weights are kept)
        if os.environ.get("VMV_AUTOTUNE", "0") == "1" and taps is None and str(device).startswith("cuda"):
            from . import autotune
            if autotune.autotune_engine(self, tag=f"B{B} F{self.Fg} {H}x{W} world{self.R}") > 0:
                self._rerecord()

    def _rerecord(self):
        """Record the plan again (the tile table changed): fresh streams / pool / accumulators, same packed weights and geometry."""
        dev = self.device
        self.B = self.B_ctx
        self.breaks, self.n_comm_ops, self.comm_bytes_in = [], 0, 0
        self.pool = Pool(dev)
        self.S = ops.Stream(record=True)
        self.Sctx = ops.Stream(record=True)
        self._keepalive, self._splitk = [], None
        self._gn_tot2.zero_()
        self._gn_tot_k = 0
        self._replays, self.graph_nodes, self.n_tuned, self.n_ruled, self.n_stale = 0, 0, 0, 0, 0
        self.S.tuner = ops.make_tuner(self)
        self._static_inputs()
        self._build()

    # ------------------------------------------------------------------ weights
    def _pack(self, sd):
        dev = self.device
        w: Dict[str, torch.Tensor] = {}
        self.w = w

        def lin(key, bias=True):
            w[key + ".weight"] = P.pack_linear(sd[key + ".weight"], dev)
            if bias and (key + ".bias") in sd:
                w[key + ".bias"] = P.pack_bias(sd[key + ".bias"], dev)

        def norm(key):
            w[key + ".weight"] = P.f32(sd[key + ".weight"], dev)
            w[key + ".bias"] = P.f32(sd[key + ".bias"], dev)

        def folded(key, wt, b, nkey, geglu=False):
            """LayerNorm `nkey` folded into the Linear (wt, b): `key`.ln.{weight,bias,colsum} (packing.fold_layernorm)."""
            if not self.fold_ln:
                return
            wf, bf, cs = P.fold_layernorm(wt, b, sd[nkey + ".weight"], sd[nkey + ".bias"])
            if geglu:
                wf, bf, cs = P.geglu_interleave(wf), P.geglu_interleave(bf), P.geglu_interleave(cs)
            w[key + ".ln.weight"] = P.pack_linear(wf, dev)
            w[key + ".ln.bias"] = P.pack_bias(bf, dev)
            w[key + ".ln.colsum"] = P.pack_bias(cs, dev)

        def tblock(p, temporal=False):
            for a in ("attn1", "attn2"):
                q = sd[f"{p}.{a}.to_q.weight"]
                k = sd[f"{p}.{a}.to_k.weight"]
                v = sd[f"{p}.{a}.to_v.weight"]
                nk = "norm1" if a == "attn1" else "norm2"
                # (with the LayerNorm folded in, only the folded copy of a consumer's weight is packed)
                if k.shape[1] == q.shape[1]:   # self-attention: fused QKV
                    if not self.fold_ln:
                        w[f"{p}.{a}.qkv"] = P.pack_linear(torch.cat([q, k, v], dim=0), dev)
                    folded(f"{p}.{a}.qkv", torch.cat([q, k, v], dim=0), None, f"{p}.{nk}")
                    if temporal and self.tqa and q.shape[1] == 320 and q.shape[0] % 64 == 0:
                        # head-major copies for the fused projection + temporal attention (csrc/gemm_tqa.hip: K = 320 levels)
                        for sfx in ((".ln.weight", ".ln.bias", ".ln.colsum") if self.fold_ln else ("",)):
                            src = w[f"{p}.{a}.qkv{sfx}"]
                            w[f"{p}.{a}.qkv.hm{sfx}"] = P.qkv_head_major(src[: 3 * q.shape[0]]).contiguous()
                else:                          # cross-attention: Q on tokens, fused KV on the context
                    if not self.fold_ln:
                        w[f"{p}.{a}.q"] = P.pack_linear(q, dev)
                    folded(f"{p}.{a}.q", q, None, f"{p}.{nk}")
                    w[f"{p}.{a}.kv"] = P.pack_linear(torch.cat([k, v], dim=0), dev)
                lin(f"{p}.{a}.to_out.0")
            if not self.fold_ln:
                w[f"{p}.ff.net.0.proj.weight"] = P.pack_linear(P.geglu_interleave(sd[f"{p}.ff.net.0.proj.weight"]), dev)
            w[f"{p}.ff.net.0.proj.bias"] = P.pack_bias(P.geglu_interleave(sd[f"{p}.ff.net.0.proj.bias"]), dev)
            folded(f"{p}.ff.net.0.proj", sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"], f"{p}.norm3", geglu=True)
            lin(f"{p}.ff.net.2")
            if self.fold_ln and sd[f"{p}.ff.net.2.weight"].shape[0] == 320:      # fused FeedForward (csrc/gemm_ff.hip: C = 320)
                w[f"{p}.ff.net.2.weight.ffperm"] = P.pack_linear(P.ff_down_permute(sd[f"{p}.ff.net.2.weight"]), dev)
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{p}.{n}")

        lin("time_embed.0"); lin("time_embed.2")
        self.has_cam = "camera_embedding.0.weight" in sd
        if self.has_cam:
            lin("camera_embedding.0"); lin("camera_embedding.2")
        self.has_fps = self.cfg.get("use_fps_condition", False) and "fps_embedding.0.weight" in sd
        if self.has_fps:
            lin("fps_embedding.0"); lin("fps_embedding.2")
        emb_w, emb_b, self.emb_off = [], [], {}
        off = 0
        for blk in self.inp + [self.mid] + self.outb:
            for kind, p, m in blk:
                if kind == "conv_in":
                    w[p + ".weight"] = P.pack_conv3x3(sd[p + ".weight"], dev)
                    w[p + ".bias"] = P.pack_bias(sd[p + ".bias"], dev)
                elif kind == "res":
                    norm(f"{p}.in_layers.0")
                    w[f"{p}.in_layers.2.weight"] = P.pack_conv3x3(sd[f"{p}.in_layers.2.weight"], dev)
                    w[f"{p}.in_layers.2.bias"] = P.pack_bias(sd[f"{p}.in_layers.2.bias"], dev)
                    emb_w.append(sd[f"{p}.emb_layers.1.weight"]); emb_b.append(sd[f"{p}.emb_layers.1.bias"])
                    self.emb_off[p] = off
                    off += m["cout"]
                    norm(f"{p}.out_layers.0")
                    w2 = sd[f"{p}.out_layers.3.weight"]
                    b2 = sd[f"{p}.out_layers.3.bias"].float()
                    w2p = w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1)
                    if f"{p}.skip_connection.weight" in sd:   # fold the 1x1 skip into conv2's K loop
                        ws = sd[f"{p}.skip_connection.weight"].reshape(w2.shape[0], -1)
                        w2p = torch.cat([w2p, ws], dim=1)
                        b2 = b2 + sd[f"{p}.skip_connection.bias"].float()
                    w[f"{p}.conv2.weight"] = P.pack_linear(w2p, dev)
                    w[f"{p}.conv2.bias"] = P.pack_bias(b2, dev)
                    for name, ix in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
                        q = f"{p}.temopral_conv.{name}"
                        norm(f"{q}.0")
                        w[f"{q}.weight"] = P.pack_tconv(sd[f"{q}.{ix}.weight"], dev)
                        w[f"{q}.bias"] = P.pack_bias(sd[f"{q}.{ix}.bias"], dev)
                elif kind in ("st", "tt"):
                    norm(f"{p}.norm")
                    lin(f"{p}.proj_in"); lin(f"{p}.proj_out")
                    tblock(f"{p}.transformer_blocks.0", temporal=(kind == "tt"))
                elif kind == "down":
                    w[f"{p}.weight"] = P.pack_conv3x3(sd[f"{p}.op.weight"], dev)
                    w[f"{p}.bias"] = P.pack_bias(sd[f"{p}.op.bias"], dev)
                elif kind == "up":
                    w[f"{p}.weight"] = P.pack_conv3x3(sd[f"{p}.conv.weight"], dev)
                    w[f"{p}.bias"] = P.pack_bias(sd[f"{p}.conv.bias"], dev)
        self.emb_total = off
        w["emb_all.weight"] = P.pack_linear(torch.cat(emb_w, dim=0), dev)
        w["emb_all.bias"] = P.pack_bias(torch.cat(emb_b, dim=0), dev)
        norm("out.0")
        w["out.2.weight"] = P.pack_conv3x3(sd["out.2.weight"], dev)
        w["out.2.bias"] = P.pack_bias(sd["out.2.bias"], dev)
        self.out_pad = w["out.2.weight"].shape[0]
        P.check_finite_weights(w, "UNet")

    # ------------------------------------------------------------------ static I/O buffers
    def _static_inputs(self):
        dev, B, F, H, W = self.device, self.B, self.F, self.H, self.W
        self.T0 = B * F * H * W
        self.cin_pad = (self.cfg["in_dim"] + 7) // 8 * 8
        self.x_rows = torch.zeros(self.T0, self.cin_pad, dtype=L.elem(), device=dev)
        self.ctx_rows = torch.zeros(B * self.L, self.cfg["context_dim"], dtype=L.elem(), device=dev)
        self.t_dev = torch.zeros(self.n_t, dtype=torch.float32, device=dev)
        self.cam_rows = torch.zeros(B * F, (self.cfg.get("camera_dim", 16) + 7) // 8 * 8, dtype=L.elem(), device=dev)
        # eps rows: own buffer, or a caller-provided [T0, out_pad] fp32 view (the two branch engines of the pipelined
        # frame-parallel mode write the two halves of one buffer, which the fused CFG + DDIM kernel reads)
        self.eps_rows = (self._eps_out if self._eps_out is not None
                         else torch.zeros(self.T0, self.out_pad, dtype=torch.float32, device=dev))
        assert self.eps_rows.shape == (self.T0, self.out_pad) and self.eps_rows.is_contiguous()
        # embedding scratch
        self.sin_emb = torch.zeros(self.n_t, self.dim, dtype=L.elem(), device=dev)
        self.te_hidden = torch.zeros(self.n_t, self.E, dtype=L.elem(), device=dev)
        self.temb = torch.zeros(self.n_t, self.E, dtype=torch.float32, device=dev)
        self.cam_hidden = torch.zeros(B * F, self.E, dtype=L.elem(), device=dev)
        self.cam_emb = torch.zeros(B * F, self.E, dtype=torch.float32, device=dev)
        self.n_cam_rows = F
        self.emb_silu = torch.zeros(B * F, self.E, dtype=L.elem(), device=dev)
        self.emb_out = torch.zeros(B * F, self.emb_total, dtype=torch.float32, device=dev)
        self.cam_valid = False
        self.extra_emb = None        # optional fp32 [n_t, E] added to the time embedding (I2VGen: fps_embedding)

    # ------------------------------------------------------------------ helpers
    def act(self, rows, C, dtype=None) -> Act:
        esz = 4 if dtype == torch.float32 else 2
        return Act(self.pool.get(rows * C * esz), rows, C, dtype)

    def release(self, a: Act):
        if self.taps is None:
            self.pool.put(a.buf)

    def _gemm(self, label, M, N, segs, wkey, out: Act, bias=None, geom=None, stream=None, **kw):
        """N is informational: the launch always covers every (4-padded) row of the packed weight."""
        W = self.w[wkey]
        ks, ws = (0, None) if kw.get("rowstat") else self._ksplit(M, W.shape[0], segs)      # (folded-LN GEMMs: no split-K)
        p = ops.gemm_params(M, W.shape[0], segs, W, out.ptr, out.C, bias=bias, geom=geom, ksplit=ks, workspace=ws, **kw)
        (stream or self.S).gemm(p, label)

    def _ksplit(self, M, N, segs):
        """Split K when the tile grid cannot fill 256 CUs and the reduction is long (small-spatial levels)."""
        if self._splitk is None:
            self._splitk = ops.SplitK(self.device, cap=8)
        return self._splitk.pick(M, N, segs)

    def _gn(self, label, srcs, rows, rows_per_stat, wkey, eps, silu, all_frames=False) -> Act:
        """all_frames: a 5-D norm whose statistics span every frame.  Frame-parallel: the input is the
        pixel-major shard, so each rank sums its HW/R pixels of all frames, the per-chunk partial sums are
        all-gathered, and every rank folds all R shards in the same order (bitwise identical statistics)."""
        C0 = srcs[0].C
        C1 = srcs[1].C if len(srcs) > 1 else 0
        C = C0 + C1
        y = self.act(rows, C)
        nfl = ops.gn_partial_floats(rows, rows_per_stat, C)
        assert nfl <= self._gnws.numel()
        nstat = rows // rows_per_stat
        base = dict(x1=srcs[1].ptr if C1 else None, ld1=srcs[1].C if C1 else 0, C1=C1)
        args = (srcs[0].ptr, srcs[0].C, C0, rows, rows_per_stat, self._gnws, self.w[wkey + ".weight"], self.w[wkey + ".bias"],
                eps, silu, y.ptr, C)
        if not all_frames:      # (one fused launch when the stat group fits on chip — the small levels — else stats + apply)
            self.S.groupnorm(ops.gn_params(*args, **base), label)
            return y
        fused = 0 if self.comm is not None else ops.gn_fused_cols(rows_per_stat, C)
        if fused:               # all-frame norm whose stat group still fits on chip (L3 / middle block)
            self.S.groupnorm_fused(ops.gn_params(*args, **base), fused, label)
            return y
        # all-frame norm: up to 256 chunks per stat group -> a one-block fold after the stats folds them once (pre-folded totals)
        assert nstat <= self.B_ctx
        tot, nxt = self._gn_tot2[self._gn_tot_k & 1], self._gn_tot2[(self._gn_tot_k + 1) & 1]
        self._gn_tot_k += 1
        clr = dict(totals_clear=nxt, clear_count=nstat * ops.GN_TOT)      # (every all-frame norm of an engine has nstat = B)
        if self.comm is None:
            self.S.groupnorm(ops.gn_params(*args, totals=tot, **clr, **base), label)
            return y
        self.S.groupnorm_stats(ops.gn_params(*args, totals=tot, **base), label)
        loc, allr = tot[: nstat * ops.GN_TOT], self._gn_tot_all[: nstat * ops.GN_TOT * self.R]
        self._collective(L.COMM_ALL_GATHER, allr, loc, label + ".totals.gather")
        self.S.groupnorm_apply(ops.gn_params(*args, totals=self._gn_tot_all, fold_ranks=self.R, **clr, **base), label)
        return y

    def _gn_folded_proj_in(self, p, x: Act, T, rps, out: Act, all_frames) -> bool:
        """SpatialTransformer / TemporalTransformer head (util.py:354-360, 1043-1050): GroupNorm -> proj_in with the norm's apply
        pass folded into the GEMM (the row-stationary kernel scales / shifts its resident rows from a per-(stat group, channel)
        table, vmv.h gn_table): statistics + a one-block-per-group table launch instead of statistics + a read-modify-write of
        the whole tensor; the GEMM multiplies the values the apply pass would have stored.  The two large levels only (K = 320 /
        640); VMV_GN_FOLD=0 disables; not on the frame-parallel plans (their totals are gathered between the two launches)."""
        if not self.gn_fold or self.comm is not None or rps < 512 or rps % 16:
            return False
        Cc = x.C
        nstat = T // rps
        tab = self.act(nstat * 2, Cc, dtype=torch.float32)
        W = self.w[f"{p}.proj_in.weight"]
        gp = ops.gemm_params(T, W.shape[0], ops.linear_segs([(x.ptr, Cc, Cc)]), W, out.ptr, out.C, bias=self.w[f"{p}.proj_in.bias"],
                             gn_table=tab.ptr, gn_rows_per_stat=rps)
        if not self.S.lib.vmv_gemm_rs_ok(C.byref(gp)):
            self.release(tab)
            return False
        label = p + ".norm"
        args = (x.ptr, Cc, Cc, T, rps, self._gnws, self.w[f"{p}.norm.weight"], self.w[f"{p}.norm.bias"], 1e-6, False, tab.ptr, Cc)
        assert ops.gn_partial_floats(T, rps, Cc) <= self._gnws.numel()
        if all_frames:          # long stat groups: integer totals (as _gn)
            assert nstat <= self.B_ctx
            tot, nxt = self._gn_tot2[self._gn_tot_k & 1], self._gn_tot2[(self._gn_tot_k + 1) & 1]
            self._gn_tot_k += 1
            gnp = ops.gn_params(*args, totals=tot, totals_clear=nxt, clear_count=nstat * ops.GN_TOT)
        else:
            gnp = ops.gn_params(*args)
        self.S.groupnorm_stats(gnp, label)
        self.S.groupnorm_table(gnp, label)
        self.S.gemm(gp, p + ".proj_in")
        self.release(tab)
        return True

    def _tconv_folded(self, q, cur: Act, T, rps, out: Act, tg, residual=None, ldr=0) -> bool:
        """One step of TemporalConvBlock_v2 (util.py:1357-1392: GroupNorm over all frames -> SiLU -> Conv3d (3,1,1)) with the norm's
        apply pass folded into the convolution: statistics (integer totals, as _gn) + a one-block table launch + the frame-resident
        kernel (csrc/gemm_tfr.hip), which applies elem(silu(x * scale + shift)) to its A tile on the way through LDS — the values
        vmv_groupnorm_apply would have stored.  The normalised tensor is never written or re-read.  Taken where the library's policy
        runs the convolution on that kernel (vmv_gemm_tfr_ok: its tiles fill the chip — the first level at 24 x 40 x 64 and
        24 x 32 x 32); frame-parallel plans gather their totals between the statistics and the table launch as before."""
        if not self.tconv_fold:
            return False
        Cc = cur.C
        nstat = T // rps
        W = self.w[f"{q}.weight"]
        tab = self.act(nstat * 2, Cc, dtype=torch.float32)
        gp = ops.gemm_params(T, W.shape[0], ops.temporal_segs(cur.ptr, Cc, Cc), W, out.ptr, out.C, bias=self.w[f"{q}.bias"], geom=tg,
                             residual=residual, ldr=ldr, gn_table=tab.ptr, gn_rows_per_stat=rps, gn_silu=True)
        if not self.S.lib.vmv_gemm_tfr_ok(C.byref(gp)):
            self.release(tab)
            return False
        label = q + ".gn"
        assert nstat <= self.B_ctx and ops.gn_partial_floats(T, rps, Cc) <= self._gnws.numel()
        args = (cur.ptr, Cc, Cc, T, rps, self._gnws, self.w[f"{q}.0.weight"], self.w[f"{q}.0.bias"], 1e-5, False, tab.ptr, Cc)
        tot, nxt = self._gn_tot2[self._gn_tot_k & 1], self._gn_tot2[(self._gn_tot_k + 1) & 1]
        self._gn_tot_k += 1
        clr = dict(totals_clear=nxt, clear_count=nstat * ops.GN_TOT)
        if self.comm is None:
            gnp = ops.gn_params(*args, totals=tot, **clr)
            self.S.groupnorm_stats(gnp, label)
            self.S.groupnorm_table(gnp, label)
        else:
            self.S.groupnorm_stats(ops.gn_params(*args, totals=tot), label)
            loc, allr = tot[: nstat * ops.GN_TOT], self._gn_tot_all[: nstat * ops.GN_TOT * self.R]
            self._collective(L.COMM_ALL_GATHER, allr, loc, label + ".totals.gather")
            self.S.groupnorm_table(ops.gn_params(*args, totals=self._gn_tot_all, fold_ranks=self.R, **clr), label)
        self.S.gemm(gp, q)
        self.release(tab)
        return True

    # ------------------------------------------------------------------ frame-parallel layout switches
    def _break(self, fn):
        self.breaks.append((self.S.nops, fn))

    def _collective(self, kind, recv_t: torch.Tensor, send_t: torch.Tensor, label: str):
        """One collective of the frame-sharded plan.  With a VmvComm handle under the communicator (RCCL, or the simulated peers of
        bench.py --simulate-rank) it is RECORDED as a plan op and issued by the C replay loop on the replay's stream; otherwise (gloo
        on CPU, host-staged debugging, VMV_COMM_NATIVE=0) the plan is cut here and Python issues it between two segments."""
        h = getattr(self.comm, "handle", None)
        if h:
            nb = send_t.numel() * send_t.element_size()
            per_rank = nb // self.R if kind == L.COMM_ALL_TO_ALL else nb
            self.n_comm_ops += 1
            self.comm_bytes_in += per_rank * (self.R - 1)        # bytes this rank RECEIVES from its R - 1 peers (= what it sends them)
            self.S.comm(ops.comm_params(h, kind, send_t.data_ptr(), recv_t.data_ptr(), per_rank), label)
        elif kind == L.COMM_ALL_TO_ALL:
            self._break(lambda: self.comm.all_to_all(recv_t, send_t))
        else:
            self._break(lambda: self.comm.all_gather(recv_t, send_t))

    def _switch(self, x: Act, hw: int, to_pixel: bool, release_in: bool = True) -> Act:
        """frame-major shard [B][F/R][HW][C]  <->  pixel-major shard [B][F][HW/R][C]: pack -> all-to-all -> unpack.
        Chunk j of the packed buffer goes to rank j; chunk i of the received buffer came from rank i.
        With B = 1 (the branch engines of the pipelined mode) one of the two copies is the identity: frame-major -> pixel-
        major receives [rank i's frames][my pixels] = the pixel-major shard itself (no unpack), pixel-major -> frame-major
        sends [rank j's frames][my pixels] = contiguous slices of the shard (no pack)."""
        R, B, Fl = self.R, self.B, self.F
        if hw % R:
            raise ValueError(f"{hw} pixels per frame do not split over {R} ranks")
        Pl = hw // R
        T, Cc = x.rows, x.C
        assert T == B * Fl * hw and Cc % 8 == 0
        cv = Cc // 8                                 # 16-byte vectors per row
        tag = "F2P" if to_pixel else "P2F"
        skip_pack = B == 1 and not to_pixel
        skip_unpack = B == 1 and to_pixel
        if skip_pack:
            send = x
        else:
            send = self.act(T, Cc)
            if to_pixel:   # send[s][b f][p c] <- x[b f][s][p c]
                self.S.copy(ops.copy_params(x.ptr, send.ptr, R, B * Fl, 1, Pl * cv, Pl * cv, hw * cv), f"shard.{tag}.pack")
            else:          # send[r][b][f p c] <- x[b][r][f p c]
                blk = Fl * Pl * cv
                self.S.copy(ops.copy_params(x.ptr, send.ptr, R, B, 1, blk, blk, R * blk), f"shard.{tag}.pack")
            if release_in:
                self.release(x)
        recv = self.act(T, Cc)
        st, rt = send.tensor().view(R, -1), recv.tensor().view(R, -1)
        self._collective(L.COMM_ALL_TO_ALL, rt, st, f"shard.{tag}.all_to_all")
        if skip_pack and release_in:
            self.release(x)      # (recycled by LATER launches only: the collective and the launches share one stream)
        if skip_unpack:
            if not skip_pack:
                self.release(send)
            return recv
        y = self.act(T, Cc)
        if to_pixel:   # y[b][r][f p c] <- recv[r][b][f p c]
            blk = Fl * Pl * cv
            self.S.copy(ops.copy_params(recv.ptr, y.ptr, B, R, 1, blk, blk, B * blk), f"shard.{tag}.unpack")
        else:          # y[b f][s][p c] <- recv[s][b f][p c]
            self.S.copy(ops.copy_params(recv.ptr, y.ptr, B * Fl, R, 1, Pl * cv, Pl * cv, B * Fl * Pl * cv), f"shard.{tag}.unpack")
        if not skip_pack:
            self.release(send)
        self.release(recv)
        return y

    def _ln_linear(self, label, x: Act, nkey, N, wkey, out: Act, bias=None, **kw):
        """out = Linear(LayerNorm(x)).  Folded: row statistics (mean, rstd) + one GEMM on the raw rows whose epilogue
        applies them (weights pre-multiplied by gamma, beta folded into the bias: packing.fold_layernorm)."""
        if not self.fold_ln:
            ln = self._ln(label + ".ln", x, nkey)
            self._gemm(label, x.rows, N, ops.linear_segs([(ln.ptr, ln.C, ln.C)]), wkey, out, bias=bias, **kw)
            self.release(ln)
            return
        base = wkey[:-len(".weight")] if wkey.endswith(".weight") else wkey
        W = self.w[base + ".ln.weight"]
        segs = ops.linear_segs([(x.ptr, x.C, x.C)])
        # The row-stationary kernel (gemm_rs.hip: K = 320 / 640 with enough rows, vmv_gemm_rs_ok) keeps the rows it multiplies in
        # registers and normalises them there: colsum + ln_eps, no statistics launch, no rowstat traffic.  VMV_LN_INLINE=1 asks
        # for in-kernel statistics wherever any kernel offers them (the persistent kernel's in-loop sums: slower).
        p = ops.gemm_params(x.rows, W.shape[0], segs, W, out.ptr, out.C, bias=self.w[base + ".ln.bias"],
                            colsum=self.w[base + ".ln.colsum"], ln_eps=1e-5, **kw)
        if self.S.lib.vmv_gemm_rs_ok(C.byref(p)) or (self.ln_inline and self.S.lib.vmv_gemm_ln_inline_ok(C.byref(p))):
            self.S.gemm(p, label)
            return
        st = self.act(x.rows, 2, dtype=torch.float32)
        self.S.layernorm(ops.ln_params(x.ptr, x.C, None, 0, None, None, x.rows, x.C, 1e-5, stats_out=st.ptr), label + ".lnstat")
        self._gemm(label, x.rows, N, segs, base + ".ln.weight", out,
                   bias=self.w[base + ".ln.bias"], rowstat=st.ptr, colsum=self.w[base + ".ln.colsum"], **kw)
        self.release(st)

    def _ln(self, label, x: Act, wkey) -> Act:
        y = self.act(x.rows, x.C)
        self.S.layernorm(ops.ln_params(x.ptr, x.C, y.ptr, y.C, self.w[wkey + ".weight"], self.w[wkey + ".bias"],
                                       x.rows, x.C, 1e-5), label)
        return y

    def _replicate(self, x: Act, label) -> Act:
        """[Bp][T, C] (one row block per prompt) -> [Bp][2][T, C]: both CFG branches of every prompt start from the shared prefix's tensor."""
        y = self.act(2 * x.rows, x.C)
        n16 = x.rows * x.C // 8 // self.Bp
        self.S.copy(ops.copy_params(x.ptr, y.ptr, self.Bp, 2, 1, n16, n16, 0), label)
        return y

    # ------------------------------------------------------------------ blocks
    def _res_block(self, p, m, srcs: List[Act], h, w) -> Act:
        B, F = self.B, self.F
        T = B * F * h * w
        cout = m["cout"]
        geom = ops.Geom(OH=h, OW=w, IH=h, IW=w, stride=1, ups=0)
        h0 = self._gn(p + ".gn1", srcs, T, h * w, f"{p}.in_layers.0", 1e-5, True)
        h1 = self.act(T, cout)
        self._gemm(p + ".conv1", T, cout, ops.conv3x3_segs([(h0.ptr, h0.C, h0.C)]), f"{p}.in_layers.2.weight", h1,
                   bias=self.w[f"{p}.in_layers.2.bias"], geom=geom,
                   rowvec=self.emb_out.data_ptr() + 4 * self.emb_off[p], rowvec_div=h * w, rowvec_ld=self.emb_total)
        self.release(h0)
        h2 = self._gn(p + ".gn2", [h1], T, h * w, f"{p}.out_layers.0", 1e-5, True)
        self.release(h1)
        h3 = self.act(T, cout)
        segs = ops.conv3x3_segs([(h2.ptr, h2.C, h2.C)])
        if m["cin"] != cout:
            segs += ops.linear_segs([(s.ptr, s.C, s.C) for s in srcs])
            self._gemm(p + ".conv2+skip", T, cout, segs, f"{p}.conv2.weight", h3, bias=self.w[f"{p}.conv2.bias"], geom=geom)
        else:
            self._gemm(p + ".conv2", T, cout, segs, f"{p}.conv2.weight", h3, bias=self.w[f"{p}.conv2.bias"], geom=geom,
                       residual=srcs[0].ptr, ldr=srcs[0].C)
        self.release(h2)
        # temporal conv block: 4 x [GN over all frames -> SiLU -> (3,1,1) conv], + identity.  Frame-parallel: pixel-local,
        # so it runs on the pixel-major shard (all Fg frames of HW/R pixels) between two layout switches.
        if self.comm is not None:
            h3 = self._switch(h3, h * w, to_pixel=True)
        tg = ops.Geom(F=self.Fg, P=(h * w) // self.R)
        cur = h3
        for i, name in enumerate(("conv1", "conv2", "conv3", "conv4")):
            q = f"{p}.temopral_conv.{name}"
            nxt = self.act(T, cout)
            last = i == 3
            if not self._tconv_folded(q, cur, T, F * h * w, nxt, tg, residual=h3.ptr if last else None, ldr=h3.C if last else 0):
                g = self._gn(q + ".gn", [cur], T, F * h * w, f"{q}.0", 1e-5, True, all_frames=True)
                self._gemm(q, T, cout, ops.temporal_segs(g.ptr, g.C, g.C), f"{q}.weight", nxt, bias=self.w[f"{q}.bias"],
                           geom=tg, residual=h3.ptr if last else None, ldr=h3.C if last else 0)
                self.release(g)
            if cur is not h3:
                self.release(cur)
            cur = nxt
        self.release(h3)
        if self.comm is not None:
            cur = self._switch(cur, h * w, to_pixel=False)
        return cur

    def _tblock(self, p, a: Act, heads, temporal: bool, h, w, cross_ctx: bool, phase: str = "all", kv_gather: bool = False) -> Act:
        """phase "pre": stop after the first (self-)attention and return a1; "post": `a` IS a1, continue from the second
        attention (the shared-prefix cut, see __init__); "all": the whole block."""
        B, F = self.B, self.F
        T = a.rows
        inner = a.C
        hw = h * w
        scale = 64 ** -0.5

        if temporal and not kv_gather:       # frame-parallel: `a` is the pixel-major shard — all Fg frames of hw / R pixels
            hw, F = hw // self.R, self.Fg

        def maps(ld, col0=0):
            if temporal:   # problems = (b, pixel); rows strided by hw
                return ops.seq_map(F * hw * ld, ld, hw * ld, inner=hw)
            return ops.seq_map(hw * ld, 0, ld, inner=1)

        n_outer = B * hw if temporal else B * F
        Nq = F if temporal else hw

        def fused_qkv_attn(tag, x: Act, normkey):
            """q | k | v projection + the per-pixel attention over the frames in ONE launch (csrc/gemm_tqa.hip, VMV_EPI_TATTN): q, k, v
            never reach memory.  Where the library serves the shape (K = 320, 48 % F == 0) and its grid fills the chip."""
            hm = f"{p}.{tag}.qkv.hm"
            if not (temporal and not kv_gather and self.tqa and (hm + (".ln.weight" if self.fold_ln else "")) in self.w):
                return None
            ao = self.act(T, inner)
            geom = ops.Geom(F=F, P=hw)
            if self.fold_ln:
                gp = ops.gemm_params(T, 3 * inner, ops.linear_segs([(x.ptr, x.C, x.C)]), self.w[hm + ".ln.weight"], ao.ptr, ao.C,
                                     bias=self.w[hm + ".ln.bias"], colsum=self.w[hm + ".ln.colsum"], ln_eps=1e-5,
                                     epilogue=L.EPI_TATTN, epi_scale=scale, geom=geom)
                ln = None
            else:
                ln = self._ln(f"{p}.{tag}.qkv.ln", x, f"{p}.{normkey}")
                gp = ops.gemm_params(T, 3 * inner, ops.linear_segs([(ln.ptr, ln.C, ln.C)]), self.w[hm], ao.ptr, ao.C,
                                     epilogue=L.EPI_TATTN, epi_scale=scale, geom=geom)
            if not self.S.lib.vmv_gemm_tqa_ok(C.byref(gp)):
                self.release(ao)
                if ln is not None:
                    self.release(ln)      # (recorded but unused LayerNorm launch: only on the non-folded debugging configuration)
                return None
            self.S.gemm(gp, f"{p}.{tag}.qkv+attn")
            if ln is not None:
                self.release(ln)
            return ao

        def self_attn(tag, x: Act, normkey) -> Act:
            ao = fused_qkv_attn(tag, x, normkey)
            if ao is not None:
                y = self.act(T, inner)
                self._gemm(f"{p}.{tag}.out", T, inner, ops.linear_segs([(ao.ptr, ao.C, ao.C)]), f"{p}.{tag}.to_out.0.weight", y,
                           bias=self.w[f"{p}.{tag}.to_out.0.bias"], residual=x.ptr, ldr=x.C)
                self.release(ao)
                return y
            qkv = self.act(T, 3 * inner)
            self._ln_linear(f"{p}.{tag}.qkv", x, f"{p}.{normkey}", 3 * inner, f"{p}.{tag}.qkv", qkv)
            ao = self.act(T, inner)
            ld = 3 * inner
            if kv_gather:
                # north-star form: my F frames of every pixel attend to ALL Fg frames — [K | V] of the local rows is made contiguous
                # (one strided copy), all-gathered (rank r's chunk = frames [r F, (r + 1) F): frame-major since B = 1), and the
                # short-sequence kernel runs with Nq = F queries against Nk = Fg keys per (pixel, head)
                cv = inner // 8
                kvloc = self.act(T, 2 * inner)
                self.S.copy(ops.copy_params(qkv.ptr + 2 * inner, kvloc.ptr, T, 1, 1, 2 * cv, 3 * cv, 0), f"{p}.{tag}.kv.pack")
                kvall = self.act(self.R * T, 2 * inner)
                out_t, in_t = kvall.tensor().view(-1), kvloc.tensor().view(-1)
                self._collective(L.COMM_ALL_GATHER, out_t, in_t, f"{p}.{tag}.kv.gather")
                self.release(kvloc)
                qm = ops.seq_map(0, ld, hw * ld, inner=hw)
                km = ops.seq_map(0, 2 * inner, hw * 2 * inner, inner=hw)
                self.S.attention(ops.attn_params(qkv.ptr, kvall.ptr, kvall.ptr + 2 * inner, ao.ptr, qm, km, km,
                                                 ops.seq_map(0, inner, hw * inner, inner=hw), hw, heads, F, self.Fg, scale),
                                 f"{p}.{tag}.attn")
                self.release(kvall)
            else:
                self.S.attention(ops.attn_params(qkv.ptr, qkv.ptr + 2 * inner, qkv.ptr + 4 * inner, ao.ptr,
                                                 maps(ld), maps(ld), maps(ld), maps(inner), n_outer, heads, Nq, Nq, scale),
                                 f"{p}.{tag}.attn")
            self.release(qkv)
            y = self.act(T, inner)
            self._gemm(f"{p}.{tag}.out", T, inner, ops.linear_segs([(ao.ptr, ao.C, ao.C)]), f"{p}.{tag}.to_out.0.weight", y,
                       bias=self.w[f"{p}.{tag}.to_out.0.bias"], residual=x.ptr, ldr=x.C)
            self.release(ao)
            return y

        def cross_attn(tag, x: Act, normkey) -> Act:
            q = self.act(T, inner)
            self._ln_linear(f"{p}.{tag}.q", x, f"{p}.{normkey}", inner, f"{p}.{tag}.q", q)
            Lc = self.L
            # K / V of the text tokens do not depend on x_t or t: computed once per sample (context_updated()), on B*L rows
            # — the reference recomputes them on 24 copies of the tokens in every forward (unet_t2v.py:346, util.py:224-225)
            Bc = self.B_ctx
            kvt = torch.empty(Bc * Lc, 2 * inner, dtype=L.elem(), device=self.device)
            self._keepalive.append(kvt)
            kv = Act(kvt.view(torch.uint8).view(-1), Bc * Lc, 2 * inner)
            cd = self.ctx_rows.shape[1]
            self._gemm(f"{p}.{tag}.kv", Bc * Lc, 2 * inner, ops.linear_segs([(self.ctx_rows.data_ptr(), cd, cd)]),
                       f"{p}.{tag}.kv", kv, stream=self.Sctx)
            ao = self.act(T, inner)
            kvm = ops.seq_map(Lc * 2 * inner, 0, 2 * inner, inner=1)
            self.S.attention(ops.attn_params(q.ptr, kv.ptr, kv.ptr + 2 * inner, ao.ptr, maps(inner), kvm, kvm, maps(inner),
                                             n_outer, heads, Nq, Lc, scale, kv_div=F), f"{p}.{tag}.attn")
            self.release(q)
            y = self.act(T, inner)
            self._gemm(f"{p}.{tag}.out", T, inner, ops.linear_segs([(ao.ptr, ao.C, ao.C)]), f"{p}.{tag}.to_out.0.weight", y,
                       bias=self.w[f"{p}.{tag}.to_out.0.bias"], residual=x.ptr, ldr=x.C)
            self.release(ao)
            return y

        if phase == "post":
            a1 = a
        else:
            a1 = self_attn("attn1", a, "norm1")
            if phase == "pre":
                return a1
        a2 = cross_attn("attn2", a1, "norm2") if cross_ctx else self_attn("attn2", a1, "norm2")
        if phase != "post":
            self.release(a1)
        a3 = self.act(T, inner)
        if self.ff_fused and f"{p}.ff.net.2.weight.ffperm" in self.w and T >= 16384:
            # the whole FeedForward in one launch, hidden activation in registers (gemm_ff.hip; the UNet's largest level)
            fp = ops.ff_params(T, inner, a2.ptr, a2.C, self.w[f"{p}.ff.net.0.proj.ln.weight"], self.w[f"{p}.ff.net.0.proj.ln.bias"],
                               self.w[f"{p}.ff.net.2.weight.ffperm"], self.w[f"{p}.ff.net.2.bias"], a3.ptr, a3.C,
                               residual=a2.ptr, ldr=a2.C, ln_eps=1e-5)
            if self.S.lib.vmv_ff_fused_ok(C.byref(fp)):
                self.S.ff(fp, f"{p}.ff.fused")
                self.release(a2)
                return a3
        ff = self.act(T, 4 * inner)
        self._ln_linear(f"{p}.ff.geglu", a2, f"{p}.norm3", 8 * inner, f"{p}.ff.net.0.proj.weight", ff,
                        bias=self.w[f"{p}.ff.net.0.proj.bias"], epilogue=L.EPI_GEGLU)
        self._gemm(f"{p}.ff.down", T, inner, ops.linear_segs([(ff.ptr, ff.C, ff.C)]), f"{p}.ff.net.2.weight", a3,
                   bias=self.w[f"{p}.ff.net.2.bias"], residual=a2.ptr, ldr=a2.C)
        self.release(ff); self.release(a2)
        return a3

    def _transformer(self, kind, p, m, x: Act, h, w, cut: bool = False) -> Act:
        """cut (shared prefix, spatial transformer only): x and everything up to the first self-attention live on ONE
        branch's rows; x and a1 are then replicated and the rest of the block runs on both branches."""
        B, F = self.B, self.F
        T = x.rows
        C = x.C
        inner = m["heads"] * m["dh"]
        if m["dh"] != 64:
            raise NotImplementedError("the HIP attention kernel is specialised for head_dim 64")
        temporal = kind == "tt"
        sharded = temporal and self.comm is not None
        kvg = sharded and self.fp_temporal == "kv_gather" and self.B == 1 and not cut
        if sharded and not kvg:        # the whole TemporalTransformer is pixel-local: run it on the pixel-major shard
            x = self._switch(x, h * w, to_pixel=True, release_in=False)
        rps = (F * h * w) if temporal else (h * w)
        a = self.act(T, inner)
        if not self._gn_folded_proj_in(p, x, T, rps, a, temporal):
            n0 = self._gn(p + ".norm", [x], T, rps, f"{p}.norm", 1e-6, False, all_frames=temporal)
            self._gemm(p + ".proj_in", T, inner, ops.linear_segs([(n0.ptr, n0.C, n0.C)]), f"{p}.proj_in.weight", a,
                       bias=self.w[f"{p}.proj_in.bias"])
            self.release(n0)
        if cut:
            assert not temporal and self.B == self.Bp
            a1 = self._tblock(f"{p}.transformer_blocks.0", a, m["heads"], temporal, h, w, cross_ctx=True, phase="pre")
            self.release(a)
            self.B = self.B_ctx                        # ---- the branches diverge here (cross-attention on their own text)
            x_full, a1_full = self._replicate(x, p + ".share.x"), self._replicate(a1, p + ".share.a1")
            self.release(a1)
            x, T = x_full, x_full.rows                 # (the caller releases the one-branch x; x_full is released below)
            a3 = self._tblock(f"{p}.transformer_blocks.0", a1_full, m["heads"], temporal, h, w, cross_ctx=True, phase="post")
            self.release(a1_full)
        else:
            a3 = self._tblock(f"{p}.transformer_blocks.0", a, m["heads"], temporal, h, w, cross_ctx=not temporal, kv_gather=kvg)
            self.release(a)
        y = self.act(T, C)
        self._gemm(p + ".proj_out", T, C, ops.linear_segs([(a3.ptr, a3.C, a3.C)]), f"{p}.proj_out.weight", y,
                   bias=self.w[f"{p}.proj_out.bias"], residual=x.ptr, ldr=x.C)
        self.release(a3)
        if cut:
            self.release(x)
        if sharded and not kvg:
            self.release(x)
            y = self._switch(y, h * w, to_pixel=False)
        return y

    def _run_block(self, blk, srcs: List[Act], h, w):
        """srcs: the block inputs — [x] or [x, skip] (decoder concat, never materialised); they are NOT released
        here (they may be pending skip connections).  Intermediates are.  Returns (Act, h, w)."""
        x = None
        for kind, p, m in blk:
            ins = srcs if x is None else [x]
            if kind == "conv_in":
                T_in = self.B * self.F * h * w          # (one branch's rows while the shared prefix is recorded)
                y = self.act(T_in, m["cout"])
                self._gemm(p, T_in, m["cout"], ops.conv3x3_segs([(self.x_rows.data_ptr(), self.cin_pad, self.cin_pad)]),
                           p + ".weight", y, bias=self.w[p + ".bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w))
            elif kind == "res":
                y = self._res_block(p, m, ins, h, w)
            elif kind in ("st", "tt"):
                y = self._transformer(kind, p, m, ins[0], h, w, cut=(kind == "st" and self.B != self.B_ctx))
            elif kind == "down":
                xin = ins[0]
                oh, ow = (h + 1) // 2, (w + 1) // 2
                T = self.B * self.F * oh * ow
                y = self.act(T, m["c"])
                self._gemm(p, T, m["c"], ops.conv3x3_segs([(xin.ptr, xin.C, xin.C)]), p + ".weight", y,
                           bias=self.w[p + ".bias"], geom=ops.Geom(OH=oh, OW=ow, IH=h, IW=w, stride=2))
                h, w = oh, ow
            elif kind == "up":
                xin = ins[0]
                oh, ow = h * 2, w * 2
                T = self.B * self.F * oh * ow
                y = self.act(T, m["c"])
                self._gemm(p, T, m["c"], ops.conv3x3_segs([(xin.ptr, xin.C, xin.C)]), p + ".weight", y,
                           bias=self.w[p + ".bias"], geom=ops.Geom(OH=oh, OW=ow, IH=h, IW=w, stride=1, ups=1))
                h, w = oh, ow
            else:
                raise ValueError(kind)
            if x is not None:
                self.release(x)
            x = y
        return x, h, w

    # ------------------------------------------------------------------ whole forward
    def _build(self):
        B, F = self.B, self.F
        S = self.S
        # The plan cleans up after itself: its FIRST launch zeroes both stat-group accumulator buffers of the all-frame norms (each
        # norm's apply pass only clears the OTHER buffer, so after an odd number of such norms — or an aborted replay — the one
        # norm 0 adds into would still hold the last statistics).  Replays therefore need nothing from prepare_rows(): run_plan() /
        # run_segment() / a per-op replay / a hipGraph capture of the plan are all self-contained.  Only the B stat groups an engine's
        # all-frame norms ever use exist: 2 buffers x B x GN_TOT int64 (2 x B x 16 KB = 64 KB at B = 2), one copy.
        S.copy(ops.copy_params(self._gn_zero.data_ptr(), self._gn_tot2.data_ptr(), 1, 1, 1, self._gn_tot2.numel() // 2, 0, 0),
               "gn.totals.clear")
        # (0) embeddings -> one [B*F, sum Cout] table for all ResBlocks
        self.n_emb_ops_start = S.nops
        e = Act(self.emb_silu.view(torch.uint8).view(-1), B * F, self.E)
        eo = Act(self.emb_out.view(torch.uint8).view(-1), B * F, self.emb_total, torch.float32)
        self._gemm("emb_all", B * F, self.emb_total, ops.linear_segs([(e.ptr, e.C, e.C)]), "emb_all.weight", eo,
                   bias=self.w["emb_all.bias"], out_fp32=True)
        h, w = self.H, self.W
        xs = []
        x = None
        share = self.share_prefix and len(self.inp) > 1 and any(k == "st" for k, _, _ in self.inp[1])
        if share:
            self.B = self.Bp     # record the shared prefix on one branch's rows (per prompt); _transformer(cut) switches back
        for bi, blk in enumerate(self.inp):
            x, h, w = self._run_block(blk, [x] if x is not None else [], h, w)
            if share and bi == 0:            # block 0's output is also a decoder skip: both branches need their copy
                x1 = x
                x_skip = self._replicate(x1, "share.skip0")
                xs.append((x_skip, h, w))
                if self.taps is not None:
                    self.taps[blk[0][1]] = (x_skip, h, w)
                continue
            if share and bi == 1:
                self.release(x1)             # (block 1 has consumed the one-branch tensor)
            xs.append((x, h, w))
            if self.taps is not None:
                self.taps[blk[0][1]] = (x, h, w)
        assert self.B == self.B_ctx
        # encoder outputs double as skips: _run_block never releases its inputs
        x_last = x
        x, h, w = self._run_block(self.mid, [x], h, w)
        if self.taps is not None:
            self.taps["middle_block"] = (x, h, w)
        for blk in self.outb:
            skip, sh, sw = xs.pop()
            assert (sh, sw) == (h, w), (sh, sw, h, w)
            y, h, w = self._run_block(blk, [x, skip], h, w)
            if x is not skip:          # the first decoder block sees the middle-block output, not a skip
                self.release(x)
            self.release(skip)
            x = y
            if self.taps is not None:
                self.taps[blk[0][1]] = (x, h, w)
        T = self.T0
        hn = self._gn("out.gn", [x], T, h * w, "out.0", 1e-5, True)
        self.release(x)
        eps = Act(self.eps_rows.view(torch.uint8).view(-1), T, self.out_pad, torch.float32)
        self._gemm("out.conv", T, self.out_pad, ops.conv3x3_segs([(hn.ptr, hn.C, hn.C)]), "out.2.weight", eps,
                   bias=self.w["out.2.bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w), out_fp32=True)
        self.release(hn)

    # ------------------------------------------------------------------ execution
    def set_context(self, y: torch.Tensor):
        """y [B, L, ctx] -> 16-bit context rows (dtype cast only)."""
        self.ctx_rows.copy_(y.reshape(self.B * self.L, -1).to(L.elem()))
        self.context_updated()

    def context_updated(self):
        """ctx_rows changed (new prompt): recompute the step-invariant K / V of every cross-attention layer."""
        if self.Sctx.nops:
            self.Sctx.run()

    def set_camera(self, camera_data: Optional[torch.Tensor]):
        """camera_data [b, F, 16] with b = 1 (shared by all branches) or b = B: the camera-embedding MLP
        (unet_t2v.py:330-335) is constant per sample, so it runs once here, not per denoising step."""
        if not self.has_cam or camera_data is None:
            self.cam_valid = False
            return
        if self.comm is not None:       # this rank's frames
            camera_data = camera_data.reshape(-1, self.Fg, camera_data.shape[-1])[:, self.rk * self.F:(self.rk + 1) * self.F]
        cam = camera_data.reshape(-1, camera_data.shape[-1])
        n = cam.shape[0]
        if n not in (self.F, self.B * self.F):
            raise ValueError(f"camera_data has {n} rows, expected {self.F} or {self.B * self.F}")
        if self.share_prefix and n != self.F:
            raise ValueError("per-branch camera_data needs an engine built with share_prefix=False")
        self.n_cam_rows = n
        self.cam_rows.zero_()
        self.cam_rows[:n, : cam.shape[1]].copy_(cam.to(L.elem()))
        S = ops.Stream(record=False)
        cd = self.cam_rows.shape[1]
        S.gemm(ops.gemm_params(n, self.E, ops.linear_segs([(self.cam_rows, cd, cd)]),
                               self.w["camera_embedding.0.weight"], self.cam_hidden, self.E,
                               bias=self.w["camera_embedding.0.bias"], act=L.ACT_SILU), "cam.0")
        S.gemm(ops.gemm_params(n, self.E, ops.linear_segs([(self.cam_hidden, self.E, self.E)]),
                               self.w["camera_embedding.2.weight"], self.cam_emb, self.E,
                               bias=self.w["camera_embedding.2.bias"], out_fp32=True), "cam.2")
        self.cam_valid = True

    def set_fps(self, fps: Optional[torch.Tensor]):
        """fps [n_t] -> fps_embedding(sinusoidal(fps)) added to the time embedding (unet_t2v.py:155-161,323-324); constant
        per sample, so it runs once here.  None (or a model without the MLP) leaves the time embedding alone."""
        if not self.has_fps or fps is None:
            self.extra_emb = None
            return
        f = fps.to(self.device).float().reshape(-1)[: self.n_t].contiguous()
        if f.numel() != self.n_t:
            f = f[:1].expand(self.n_t).contiguous()
        S = ops.Stream(record=False)
        sin = torch.empty(self.n_t, self.dim, dtype=L.elem(), device=self.device)
        hid = torch.empty(self.n_t, self.E, dtype=L.elem(), device=self.device)
        out = torch.empty(self.n_t, self.E, dtype=torch.float32, device=self.device)
        ops.sinusoidal(f, sin, self.n_t, self.dim)
        S.gemm(ops.gemm_params(self.n_t, self.E, ops.linear_segs([(sin, self.dim, self.dim)]), self.w["fps_embedding.0.weight"],
                               hid, self.E, bias=self.w["fps_embedding.0.bias"], act=L.ACT_SILU), "fps.0")
        S.gemm(ops.gemm_params(self.n_t, self.E, ops.linear_segs([(hid, self.E, self.E)]), self.w["fps_embedding.2.weight"],
                               out, self.E, bias=self.w["fps_embedding.2.bias"], out_fp32=True), "fps.2")
        self._fps_keep = (f, sin, hid)        # stream-ordered temporaries of the launches above
        self.extra_emb = out

    def _embeddings(self):
        S = ops.Stream(record=False)
        ops.sinusoidal(self.t_dev, self.sin_emb, self.n_t, self.dim)
        S.gemm(ops.gemm_params(self.n_t, self.E, ops.linear_segs([(self.sin_emb, self.dim, self.dim)]),
                               self.w["time_embed.0.weight"], self.te_hidden, self.E,
                               bias=self.w["time_embed.0.bias"], act=L.ACT_SILU), "time_embed.0")
        S.gemm(ops.gemm_params(self.n_t, self.E, ops.linear_segs([(self.te_hidden, self.E, self.E)]),
                               self.w["time_embed.2.weight"], self.temb, self.E,
                               bias=self.w["time_embed.2.bias"], out_fp32=True,
                               rowvec=self.extra_emb, rowvec_div=1, rowvec_ld=self.E), "time_embed.2")
        rows = self.B * self.F
        # branch-major rows [b][f]: the B // n_t branches of one timestep row are contiguous
        ops.emb_combine_silu(self.temb, self.cam_emb if self.cam_valid else None, self.emb_silu, rows, self.E,
                             rows // self.n_t, self.n_cam_rows)

    def forward_rows(self, x: torch.Tensor, t: torch.Tensor):
        """x [b, C, F, H, W] fp32 on device (b divides B; replicated to the B branches; frame-parallel: this rank's
        F/R frames), t [n_t].
        Leaves eps in ``self.eps_rows`` (fp32 [B*F*H*W, out_pad])."""
        self.prepare_rows(x, t)
        self.run_plan()
        return self.eps_rows

    def prepare_rows(self, x: torch.Tensor, t: torch.Tensor, pair_major: bool = False):
        """Everything of forward_rows before the plan: latent -> rows, timestep, embeddings.  ``pair_major``: the B // nb copies of
        every sample are ADJACENT row blocks ([x_0 | x_0 | x_1 | x_1]: the batched CFG pass, unet_t2v._forward_cfg_rows_batched)
        instead of the default tiling ([x_0 | x_1 | x_0 | x_1])."""
        nb = x.shape[0]
        # only the latent's own channels are written: channels >= x.shape[1] hold zeros (T2V) or the step-invariant
        # image `concat` of the I2VGen front-end (unet_i2vgen.py:383)
        if pair_major and nb > 1:
            x, rep = x.contiguous(), self.B // nb
            per = rep * self.F * self.H * self.W
            if self.share_prefix:        # the input conv belongs to the shared prefix: it reads one row block per prompt, [x_0 | x_1 ...]
                ops.latent_to_rows_keep(x, self.x_rows[:nb * self.F * self.H * self.W], self.cin_pad, 1)
            else:
                for s in range(nb):
                    ops.latent_to_rows_keep(x[s:s + 1], self.x_rows[s * per:(s + 1) * per], self.cin_pad, rep)
        else:
            ops.latent_to_rows_keep(x.contiguous(), self.x_rows, self.cin_pad, self.B // nb)
        self.t_dev.copy_(t.to(torch.float32).reshape(-1)[: self.n_t])
        self._embeddings()

    def segments(self):
        """The recorded plan cut at its collectives: [(first, last, collective or None)] — launches [first, last) then the
        collective that precedes launch `last`."""
        segs, first = [], 0
        for idx, fn in self.breaks:
            segs.append((first, idx, fn))
            first = idx
        segs.append((first, self.S.nops, None))
        return segs

    def run_segment(self, seg):
        first, last, fn = seg
        if first == 0 and last == self.S.nops and fn is None:      # an uncut plan: the whole-plan path (one C call / one graph launch)
            return self.run_plan()
        if last > first:
            self.S.run(first, last)
        if fn is not None:
            fn()

    def run_plan(self):
        """Replay the recorded launches; frame-parallel plans are cut at their collectives."""
        if not self.breaks:
            # VMV_GRAPH=1: the whole plan (its RCCL collectives included) as ONE hipGraph launch — captured on the second replay, after
            # an eager one has set the kernels' first-use attributes.  At 1 GPU the step is GPU-bound either way; it
            # matters once a rank's share of a step is a few ms (frame-parallel at 8 GPUs).
            if self.use_graph and self.S.graph is None and self._replays >= 1:
                try:
                    self.graph_nodes = self.S.capture_graph()
                except L.VmvError:
                    self.use_graph = False
            self.S.run()
            self._replays += 1
            return
        for seg in self.segments():
            self.run_segment(seg)

    def saturation_report(self, limit: int = 20):
        """Opt-in fp16 saturation probe: with MODE.FP16_OVFL every 16-bit store clamps at +-65504 instead of producing inf,
        so a checkpoint whose activations leave fp16's range yields a plausible but WRONG video with no non-finite value anywhere.
        This replays the recorded plan launch by launch (slow: one sync per GEMM; a validation pass, not a production path — no
        collectives are replayed, so unsharded plans only) and counts, per 16-bit GEMM output, the elements sitting exactly on the
        clamp.  -> [(label, count, elements)] of the launches that saturate.  ``VMV_F16_SAT_PROBE=1`` makes the sampler run it once per
        sample after the first step and warn (diffusion_ddim.ddim_sample_loop); nothing to report on the bf16 build."""
        if L.elem_name() != "fp16" or not str(self.device).startswith("cuda") or self.breaks or self.n_comm_ops:
            return []
        from .autotune import out_view
        hits = []
        for i, ((op, p), label) in enumerate(zip(self.S.recorded, self.S.labels)):
            self.S.run(i, i + 1)
            if op != L.OP_GEMM or p.out_fp32:
                continue
            o = out_view(p)
            n = int((o.abs() >= 65504.0).sum())
            if n:
                hits.append((label, n, o.numel()))
                if len(hits) >= limit:
                    break
        return hits

    def eps_ncfhw(self) -> torch.Tensor:
        """eps rows -> [B, out_dim, F, H, W] fp32 (reference output layout)."""
        out = torch.empty(self.B * self.F, self.out_pad, self.H, self.W, dtype=torch.float32, device=self.device)
        ops.rows_to_nchw(self.eps_rows, self.out_pad, out)
        od = self.cfg["out_dim"]
        return out[:, :od].reshape(self.B, self.F, od, self.H, self.W).permute(0, 2, 1, 3, 4).contiguous()
