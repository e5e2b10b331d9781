"""Per-shape kernel choice by measurement (host logic; the kernels are csrc/gemm*.hip).

``tune_plan(engine, table, workspace, tag)`` replays every distinct implicit-GEMM launch of a recorded plan with every (tile family,
split-K factor) the library accepts, keeps a candidate only if its output agrees with the built-in policy's (rel-L2 <= 2e-3) and it is
faster by the stated margin in two independent timings, and returns the winners keyed by ``ops.gemm_signature``.  Two users:

  * ``tools/autotune_gemm.py`` — writes the table that ships with the package (``videomv_amd/tuned_gemm.json``: the BASELINE shapes);
  * ``VMV_AUTOTUNE=1`` — an engine built for a shape that table does not cover (another resolution / frame count / world size) tunes
    itself once at build time (10-40 s per plan), re-records its plan with the winners and appends them to the per-user cache
    ``~/.cache/videomv_amd/tuned_gemm.json`` (``VMV_TUNED_CACHE`` overrides the path), which ``ops.tuned_table()`` merges over the
    packaged table from then on.  Off by default: building an engine then costs nothing extra and depends on no cache file.

The built-in policy of csrc/gemm.hip was fitted by hand to M = 122 880 / 30 720 / 7 680 / 1 920 (latent 24x40x64); on the reference's
own 24x32x32 it left 16 % of the step on the table.  Measurement is the general answer; the table is data, and a stale
entry is dropped with a warning when the plan is recorded (``ops.make_tuner`` validates every forced choice with ``vmv_gemm_validate``).

Multi-rank plans: the tuning replay skips the plan's collectives (``Stream.run_local``), so a rank with nothing left to
tune never leaves its peers inside one; rank 0's winners are broadcast and every rank records the same choices (the replicated VAE /
LGM plans then round identically on every rank); signatures that were tested without a gain are remembered in the cache (``"done"``)
so a later engine build does not re-tune them; the cache file is merged under an ``flock``."""
import ctypes as C
import json
import os

import torch

from . import _lib as L
from . import ops

TILES = [L.TILE_128x128, L.TILE_128x160, L.TILE_128x64, L.TILE_64x64, L.TILE_256x128, L.TILE_256x160, L.TILE_G128x128, L.TILE_G128x160,
         L.TILE_P256x128, L.TILE_P256x160, L.TILE_Q128x128, L.TILE_Q96x160, L.TILE_X256x320, L.TILE_X256x256, L.TILE_X256x128, L.TILE_X512x128,
         L.TILE_RS, L.TILE_RS256, L.TILE_RS512]
KSPLITS = [2, 3, 4, 6, 8, 12, 16]
WS_CAP = 512 << 20


def clone(p):
    q = L.GemmParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
    return q


class _DevView:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(int(ptr), False), version=2)


def out_view(p):
    """The [M, ldo] output of a recorded GEMM as a tensor view (no copy)."""
    n_out = p.N // 2 if p.epilogue == L.EPI_GEGLU else (p.N // 3 if p.epilogue == L.EPI_TATTN else p.N)
    if p.out_fp32:
        t = torch.as_tensor(_DevView(p.out, p.M * p.ldo, "<f4"), device="cuda").view(p.M, p.ldo)
    else:
        t = torch.as_tensor(_DevView(p.out, p.M * p.ldo, "<i2"), device="cuda").view(L.elem()).view(p.M, p.ldo)
    return t[:, :n_out]


def time_us(lib, p, stream, reps=10, warm=2):
    for _ in range(warm):
        if lib.vmv_gemm(C.byref(p), stream) != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lib.vmv_gemm(C.byref(p), stream)
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


def tune_plan(eng, table, ws, tag, gains=(0.93, 0.90), verbose=True):
    lib = eng.S.lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    eng.S.run_local()                             # realistic (finite) contents in every buffer; no collectives (see the header)
    torch.cuda.synchronize()
    seen = {}
    for (op, p), label in zip(eng.S.recorded, eng.S.labels):
        if op != L.OP_GEMM or p.wgroup_rows or p.epilogue == L.EPI_TATTN:      # (the fused q|k|v + attention has one kernel: nothing to tune)
            continue
        sig = ops.gemm_signature(p)
        if sig in seen or sig in table["done"]:
            continue
        seen[sig] = label
    if verbose:
        print(f"[{tag}] {len(seen)} new GEMM signatures", flush=True)
    for sig, label in seen.items():
        p0 = next(p for (op, p) in eng.S.recorded if op == L.OP_GEMM and ops.gemm_signature(p) == sig)
        base = clone(p0)
        base_tile = lib.vmv_gemm_pick_tile(C.byref(base))
        t_base = time_us(lib, base, stream)
        if t_base is None:
            continue
        ref = out_view(base).float().clone()
        ref = torch.nan_to_num(ref, nan=0.0, posinf=0.0, neginf=0.0)
        refn = float(ref.norm()) + 1e-12
        steps = sum((base.seg[i].k + 63) // 64 for i in range(base.nseg))
        tiles128 = ((base.M + 127) // 128) * ((base.N + 127) // 128)
        ks_ok = not (base.rowstat or base.ln_eps > 0 or base.gn_table) and tiles128 < 768
        cands = []
        for tile in TILES:
            for ks in [0] + ([k for k in KSPLITS if steps >= 2 * k and k * base.M * base.N * 4 <= WS_CAP] if ks_ok else []):
                if tile in (L.TILE_RS, L.TILE_RS256, L.TILE_RS512) and ks:
                    continue
                if tile == base_tile and ks == (base.ksplit if base.ksplit > 1 else 0):
                    continue
                cands.append((tile, ks))
        best = (t_base, base_tile, base.ksplit if base.ksplit > 1 else 0)
        need = gains[1] if t_base >= 120.0 else gains[0]
        for tile, ks in cands:
            q = clone(p0)
            q.tile, q.ksplit = tile, ks
            q.workspace = ws.data_ptr() if ks > 1 else None
            if lib.vmv_gemm_pick_tile(C.byref(q)) < 0:
                continue
            out_view(q).zero_()
            if lib.vmv_gemm(C.byref(q), stream) != 0:
                continue
            torch.cuda.synchronize()
            got = torch.nan_to_num(out_view(q).float(), nan=0.0, posinf=0.0, neginf=0.0)
            err = float((got - ref).norm()) / refn
            if not (err <= 2e-3):
                continue
            t = time_us(lib, q, stream)
            if t is not None and t < best[0]:
                best = (t, tile, ks)
        entry = None
        if (best[1], best[2]) != (base_tile, base.ksplit if base.ksplit > 1 else 0):
            q = clone(p0)
            q.tile, q.ksplit, q.workspace = best[1], best[2], (ws.data_ptr() if best[2] > 1 else None)
            t2, tb2 = time_us(lib, q, stream, reps=20), time_us(lib, base, stream, reps=20)      # second, independent timing
            if t2 is not None and tb2 is not None and t2 <= need * tb2 and best[0] <= need * t_base and tb2 - t2 >= 1.5:
                entry = dict(tile=int(best[1]), ksplit=int(best[2]), us=round(t2, 1), base_us=round(tb2, 1), base_tile=int(base_tile),
                             base_ksplit=int(base.ksplit), label=label, plan=tag)
        lib.vmv_gemm(C.byref(base), stream)          # leave the policy's result in the buffer
        table["done"].add(sig)
        if entry:
            table["entries"][sig] = entry
        if entry and verbose:
            print(f"  {label:58s} {sig.split(';')[0]:22s} tile {base_tile:2d}/ks{base.ksplit} {entry['base_us']:7.1f} us -> tile {entry['tile']:2d}/ks{entry['ksplit']} "
                  f"{entry['us']:7.1f} us ({entry['us'] / entry['base_us']:.2f})", flush=True)


def cache_path() -> str:
    return os.environ.get("VMV_TUNED_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "videomv_amd", "tuned_gemm.json")


def _real_world(eng) -> int:
    c = getattr(eng, "comm", None)
    if c is None or getattr(c, "backend", "") == "sim" or getattr(c, "local_only", False):
        return 1
    return int(getattr(c, "world", 1))


def _merge_cache(entries: dict, tested: set):
    """Merge winners and tested-without-gain signatures into the per-user cache under an exclusive lock (several ranks / processes
    may finish tuning at the same time: read-modify-write of one file)."""
    path = cache_path()
    try:
        import fcntl
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            old = {}
            if os.path.exists(path):
                try:
                    with open(path) as f:
                        old = json.load(f)
                except ValueError:
                    old = {}
            old.setdefault(L.elem_name(), {}).update(entries)
            done = set(old.get("done", {}).get(L.elem_name(), [])) | set(tested)
            old.setdefault("done", {})[L.elem_name()] = sorted(done - set(old[L.elem_name()]))
            tmp = path + f".{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(old, f, indent=1, sort_keys=True)
            os.replace(tmp, path)
    except OSError:
        pass            # (a read-only home: the choices still hold for this process)


def cached_done() -> set:
    """Signatures an earlier run tested and found no better choice for (not re-tuned)."""
    try:
        with open(cache_path()) as f:
            return set(json.load(f).get("done", {}).get(L.elem_name(), []))
    except (OSError, ValueError):
        return set()


def autotune_engine(eng, tag="autotune") -> int:
    """VMV_AUTOTUNE=1: tune the GEMM signatures of `eng`'s freshly recorded plan that no table covers yet; returns the number of
    improved signatures (the caller re-records the plan when it is > 0).  Winners go into the in-memory table and the per-user cache.
    Collective when the engine is frame-parallel over real peers: every rank calls it at the same point (engine construction is
    already collective), rank 0 measures, everyone adopts rank 0's table."""
    if not torch.cuda.is_available() or str(eng.device).startswith("cpu"):
        return 0
    import torch.distributed as dist
    world = _real_world(eng)
    group = getattr(getattr(eng, "comm", None), "group", None) if world > 1 else None
    known = ops.tuned_table()
    table = dict(done=set(known.keys()) | cached_done(), entries={})
    need = 0
    for (op, p) in eng.S.recorded:
        if op == L.OP_GEMM and not p.wgroup_rows and p.tile == L.TILE_AUTO and ops.gemm_signature(p) not in table["done"]:
            need += 1
    if world > 1:           # the ranks record the same plan, but agree explicitly: one decision for the group
        flag = torch.tensor([need], device=eng.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        need = int(flag.item())
    if not need:
        return 0
    before = set(table["done"])
    measuring = world == 1 or dist.get_rank(group) == 0
    if measuring:
        ws = torch.empty(WS_CAP, dtype=torch.uint8, device=eng.device)
        with torch.cuda.device(eng.device):
            tune_plan(eng, table, ws, tag, verbose=os.environ.get("VMV_AUTOTUNE_VERBOSE", "0") == "1")
        del ws
    if world > 1:
        box = [(table["entries"], sorted(table["done"] - before)) if measuring else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        table["entries"], tested = dict(box[0][0]), set(box[0][1])
    else:
        tested = table["done"] - before
    known.update(table["entries"])
    if measuring:
        _merge_cache(table["entries"], tested)
    return len(table["entries"])
