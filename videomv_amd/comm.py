"""Frame-parallel communicator: the two collectives the frame-sharded sampler needs, on top of
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU node).

  * ``all_to_all(out, inp)`` — equal splits along dim 0; switches an activation between the frame-major shard
    ([B][F/R][HW][C]: spatial convs, per-frame GroupNorm, spatial / cross attention) and the pixel-major shard
    ([B][F][HW/R][C]: temporal convs, temporal attention).  On the fully connected xGMI mesh every rank talks to its
    7 peers at once, and the volume is (R-1)/R of the local shard — 8x less than all-gathering K and V.
  * ``all_gather(out, inp)`` — the per-chunk GroupNorm partial sums of the 5-D norms whose statistics span all frames.

With a "gloo" group the same calls work on CPU tensors (the world_size-2 CPU tests) and, for device tensors, stage
through host memory — a debugging aid that lets two processes share one GPU; it is never the fast path.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def _rccl_path() -> str:
    """The librccl this process already uses (PyTorch ships its own copy): libvmv must talk to THAT instance, not to a second
    copy from /opt/rocm.  Found in the process's memory map, else next to torch's libraries, else left to the loader."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else ""


def _all_ok(ok: bool, group, device) -> bool:
    """Did EVERY rank of the group succeed?  (one small all-reduce: the ranks must take the same branch afterwards)"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def native_comm(group, device):
    """A VmvComm* over the ranks of `group` (include/vmv.h vmv_comm_*): rank 0 draws the RCCL id, torch.distributed carries its 128
    bytes to the others, every rank joins.  Collective: all ranks of the group call it at the same point, and all of them get the
    same answer — a handle everywhere or ``None`` everywhere (a rank that failed alone used to fall back to the Python
    collectives while its peers blocked in the id broadcast, or later recorded a different plan): every local step that can fail is
    followed by an agreement all-reduce before the next collective step."""
    from . import _lib as L
    lib = L.load()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    idb = (C.c_uint8 * L.COMM_ID_BYTES)()
    err = None
    try:
        L.check(lib.vmv_comm_load(_rccl_path().encode()), "vmv_comm_load")
        if rank == 0:
            L.check(lib.vmv_comm_unique_id(idb), "vmv_comm_unique_id")
    except Exception as e:
        err = e
    if not _all_ok(err is None, group, device):
        raise L.VmvError(f"vmv_comm_load / unique_id failed on {'this' if err else 'another'} rank" + (f": {err}" if err else ""))
    t = torch.tensor(list(idb), dtype=torch.uint8, device=device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(t, src=src, group=group)
    idb = (C.c_uint8 * L.COMM_ID_BYTES)(*[int(v) for v in t.cpu()])
    with torch.cuda.device(device):
        h = lib.vmv_comm_create(idb, world, rank)
    if not _all_ok(bool(h), group, device):
        if h:
            lib.vmv_comm_destroy(h)
        raise L.VmvError("vmv_comm_create failed (ncclCommInitRank) on " + ("another rank" if h else "this rank"))
    return h


class FrameComm:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("frame-parallel sampling needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # world 1 needs no exchange; VMV_COMM_FORCE=1 still issues the collectives (smoke-tests RCCL on a 1-GPU box)
        self.local_only = self.world == 1 and os.environ.get("VMV_COMM_FORCE") != "1"
        self.n_all_to_all = 0
        self.n_all_gather = 0
        self._twin = None
        # handle (VmvComm*): with RCCL underneath the collectives are recorded INTO the plans (VMV_OP_COMM) and issued by the C
        # replay loop on the replay's stream — one host call per forward instead of one Python collective per plan segment
        # (DESIGN.md §8).  Created on first use (collective: every rank of the group records its first plan at the same point);
        # VMV_COMM_NATIVE=0 keeps the torch.distributed calls (the gloo path always does).
        self._handle, self._handle_tried = None, False

    @property
    def handle(self):
        if not self._handle_tried:
            self._handle_tried = True
            if self.backend == "nccl" and not self.local_only and os.environ.get("VMV_COMM_NATIVE", "1") != "0" and torch.cuda.is_available():
                try:
                    self._handle = native_comm(self.group, torch.device("cuda", torch.cuda.current_device()))
                except Exception as e:      # (the torch.distributed path needs nothing from libvmv: keep going on it)
                    import warnings
                    warnings.warn(f"vmv_comm_* unavailable ({type(e).__name__}: {e}); collectives stay in Python")
        return self._handle

    def close(self):
        """Destroy the RCCL communicator under this FrameComm (collective: every rank calls it; call it BEFORE
        dist.destroy_process_group()).  Not done from __del__: a communicator torn down during interpreter shutdown, after the HIP
        runtime / torch's own process group are gone, can block the exiting process."""
        if self._handle:
            from . import _lib as L
            L.load().vmv_comm_destroy(self._handle)
        self._handle, self._handle_tried = None, True
        if self._twin is not None and self._twin._handle:
            self._twin.close()

    def twin(self) -> "FrameComm":
        """A second communicator over the same ranks (its own process group, so that its collectives can be in flight on
        another stream at the same time): the pipelined frame-parallel mode runs the two CFG branches on two streams, one
        communicator each.  Collective: every rank must call it at the same point."""
        if self._twin is None:
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(dist.get_world_size()))
            self._twin = FrameComm(dist.new_group(ranks=ranks, backend=self.backend))
            self._twin._twin = self
        return self._twin

    def _staged(self, t):
        return self.backend == "gloo" and t.is_cuda

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        self.n_all_to_all += 1
        if self.local_only:
            out.copy_(inp)
        elif self._staged(inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, group=self.group)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        self.n_all_gather += 1
        out, inp = out.view(-1), inp.view(-1)          # [R * n] <- n per rank (gloo insists on flat tensors)
        if self.local_only:
            out.copy_(inp)
        elif self._staged(inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)


class SimComm:
    """Rank `rank` of a `world`-GPU frame-parallel run with the other W - 1 ranks ABSENT (include/vmv.h vmv_comm_create_sim): every
    collective is a device-local copy of the bytes the real one would deliver, recorded into the plan like the RCCL ones.  Lets ONE
    GPU build, replay, time and profile the rank-local plan of BASELINE configs[2] (3 of 24 views, HW / 8 pixels per temporal block):
    launch sequence, tile choices, local HBM traffic and host time are the real rank's; the OUTPUT is not a sample (peers' data is
    this rank's own) and the wire time is absent — bench.py --simulate-rank reports both facts next to the numbers."""

    def __init__(self, world: int, rank: int = 0):
        from . import _lib as L
        self.world, self.rank = int(world), int(rank)
        self.backend, self.local_only, self.group = "sim", False, None
        self.n_all_to_all = self.n_all_gather = 0
        self._twin = None
        self.handle = L.load().vmv_comm_create_sim(self.world, self.rank)
        if not self.handle:
            raise ValueError(f"bad simulated communicator {rank}/{world}")

    def __del__(self):
        try:
            if self.handle:
                from . import _lib as L
                L.load().vmv_comm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def twin(self) -> "SimComm":
        if self._twin is None:
            self._twin = SimComm(self.world, self.rank)
            self._twin._twin = self
        return self._twin

    # (eager forms, for callers outside a plan: gather_frames at the end of a sample)
    def all_to_all(self, out, inp):
        self.n_all_to_all += 1
        out.copy_(inp)

    def all_gather(self, out, inp):
        self.n_all_gather += 1
        out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))


class SimCfgFrameComm:
    """Rank 0 of a W-GPU CFG-parallel x frame-parallel run (2 branch groups of W / 2 ranks) with every peer ABSENT — the ``SimComm``
    idea for ``CfgFrameComm``: the branch group's layout switches / totals gathers are device-local copies recorded into the B = 1
    plan, and the once-per-step eps exchange with the partner rank delivers this rank's own rows in both slots.  Same launch sequence,
    tiles, local traffic and host time as the real rank; not a sample (bench.py --simulate-rank reports it as mode "cfg x frame")."""

    def __init__(self, world: int):
        if world % 2:
            raise ValueError("CFG-parallel needs an even number of ranks")
        self.branch, self.rank, self.world = 0, 0, world // 2
        self.fp = SimComm(self.world, 0)
        self.local_only, self.backend, self.group = False, "sim", None

    def all_gather(self, out, inp):
        self.fp.all_gather(out, inp)

    def all_to_all(self, out, inp):
        self.fp.all_to_all(out, inp)

    def exchange_branches(self, out, mine):
        out.view(2, -1).copy_(mine.view(1, -1).expand(2, -1))

    def close(self):
        pass


class CfgFrameComm:
    """CFG-parallel x frame-parallel (SURVEY §8e "other cheap axis"): W ranks = 2 branch groups of W/2.  Group b runs ONLY the
    b-th classifier-free-guidance branch (0 = cond, 1 = uncond), its frames sharded W/2 ways with the same layout-switch
    schedule as ``FrameComm`` — half the launches per rank, half as many peers per collective, no second copy of anything —
    and the partner ranks (r, r + W/2), which hold the same frames of the two branches, exchange their eps rows once per
    step (one small all-gather in a 2-rank group) before the fused CFG + DDIM update, which both then compute identically.
    Frame-sharding view for callers (sampler, ``gather_frames``): ``world`` = W/2, ``rank`` = index inside the branch group.
    Collective: every rank constructs it at the same point (it creates W/2 + 2 process groups)."""

    def __init__(self):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("CFG-parallel sampling needs an initialised torch.distributed process group")
        W, r = dist.get_world_size(), dist.get_rank()
        if W % 2:
            raise ValueError(f"CFG-parallel needs an even number of ranks, got {W}")
        half = W // 2
        backend = dist.get_backend()
        self.branch, self.rank, self.world = r // half, r % half, half
        fp_groups = [dist.new_group(ranks=[b * half + i for i in range(half)], backend=backend) for b in range(2)]
        pair_groups = [dist.new_group(ranks=[i, i + half], backend=backend) for i in range(half)]
        self.fp = FrameComm(fp_groups[self.branch])          # frame-parallel communicator of my branch group
        self.pair = FrameComm(pair_groups[self.rank])        # me and the rank holding the other branch of my frames
        assert self.fp.rank == self.rank and self.fp.world == half and self.pair.rank == self.branch
        self.local_only = False

    # frame-sharding collectives (gather_frames, whole-sample forward) go to the branch group
    def all_gather(self, out, inp):
        self.fp.all_gather(out, inp)

    def all_to_all(self, out, inp):
        self.fp.all_to_all(out, inp)

    def close(self):
        """Destroy the RCCL communicators of both sub-groups (collective; before dist.destroy_process_group())."""
        self.fp.close()
        self.pair.close()

    def exchange_branches(self, out, mine):
        """out [2, n] <- (cond rows, uncond rows) of this rank's frames: slot b comes from the rank of branch b."""
        self.pair.all_gather(out, mine)
