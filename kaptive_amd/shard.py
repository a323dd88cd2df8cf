"""Partitioning of assemblies across ranks (one process per GPU, no data-path collective).

The reference types genomes in a serial loop and ignores ``--threads`` (src/kaptive/serotyping/cli.py:167-174,
206-208); assemblies are independent, so the multi-GPU path is a static partition plus a gather of result rows.
"""

from __future__ import annotations

from typing import Sequence, TypeVar

T = TypeVar("T")


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition: rank r gets items [lo, hi); sizes differ by at most one, earlier ranks larger."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(items: Sequence[T], rank: int, world: int) -> Sequence[T]:
    lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def gather_rows(local_rows: list[bytes], group=None) -> list[bytes] | None:
    """Rank 0 receives every rank's report rows in rank order (others get None). Uses torch.distributed object
    gather: rows are host bytes, nothing touches the device."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = [None] * world if rank == 0 else None
    dist.gather_object(local_rows, out, dst=0, group=group)
    if rank != 0:
        return None
    return [row for part in out for row in part]
