"""Multi-GPU partitioning of a flattened signature batch and the verdict-bitmap exchange.

Signatures are independent, so the batch is cut into contiguous, 64-aligned, equal-sized shards (one
per rank = one process per GPU); each rank writes whole u64 verdict words and a single all-gather
(RCCL over xGMI when the tensors are on GPUs, gloo in the CPU tests) leaves the merged bitmap on every
rank.  The reference has no analogue: its fan-in is a Go channel
(core/committer/txvalidator/v20/validator.go:215-239)."""
from __future__ import annotations

from typing import Tuple


def shard_words(n: int, world: int) -> int:
    """u64 verdict words per rank: ceil(ceil(n/64) / world)."""
    words = (n + 63) // 64
    return (words + world - 1) // world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the tuples rank `rank` verifies; lo is a multiple of 64; hi-lo may be 0 for tail ranks."""
    per = shard_words(n, world) * 64
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def allgather_verdicts(local_words, n: int, world: int, group=None):
    """local_words: torch int64 tensor [shard_words] (bit-identical view of the u64 words, zero padded).
    Returns the merged int64 tensor [ceil(n/64)] on every rank."""
    import torch
    import torch.distributed as dist

    sw = shard_words(n, world)
    assert local_words.numel() == sw
    if world == 1:
        return local_words[: (n + 63) // 64]
    out = torch.empty(sw * world, dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(out, local_words.contiguous(), group=group)
    return out[: (n + 63) // 64]
