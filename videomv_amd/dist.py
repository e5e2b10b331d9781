"""Multi-GPU glue (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

Round-1 parallelism = the reference's: **replicas** — every rank samples with ``seed + rank`` and there is no
data-path collective (inference_text2video_entrance.py:79,152-156).  The reference makes every rank run the *whole*
prompt list; ``shard_prompts`` additionally offers the obvious split (rank r takes prompts r, r+W, …) so that N GPUs
finish a prompt list N times sooner.  ``max_over_ranks`` is the timing reduction bench.py reports."""
from typing import List, Sequence

import torch
import torch.distributed as dist


def rank_seed(base_seed: int, rank: int) -> int:
    return int(base_seed) + int(rank)          # :79  seed + rank


def shard_prompts(prompts: Sequence, rank: int, world: int, replicate: bool = False) -> List:
    """replicate=True reproduces the reference (each rank gets the full list); otherwise round-robin."""
    if replicate or world <= 1:
        return list(prompts)
    return [p for i, p in enumerate(prompts) if i % world == rank]


def max_over_ranks(seconds: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_counts(n_local: int, device="cpu") -> List[int]:
    """All ranks learn how many samples every rank produced (used to assemble `value` = total units / max time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(n_local)]
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(o[0]) for o in out]
