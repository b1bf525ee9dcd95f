"""Multi-GPU sharding of the hot path: tiles (= SingleAlign calls) are independent, so
ranks take disjoint subsets and no collective touches the data path.

``shard_tiles`` is greedy longest-processing-time on DP cells so every rank gets the same
amount of work; in end-to-end mode the unit would be a read (all tiles of a read stay on
one rank because interval i+1 is trimmed against the result of interval i, reference
src/AlignmentBuffer.cpp:3361-3406) -- pass ``group`` for that."""
from __future__ import annotations

import heapq
from typing import List, Optional, Sequence


def shard_tiles(cells: Sequence[int], world: int, group: Optional[Sequence[int]] = None) -> List[List[int]]:
    """Assign item indices to `world` ranks, balancing the sum of `cells`.
    Items sharing a `group` id stay together.  Deterministic."""
    if world <= 0:
        raise ValueError("world must be positive")
    units = {}
    for i, c in enumerate(cells):
        g = group[i] if group is not None else i
        u = units.setdefault(g, [0, []])
        u[0] += int(c)
        u[1].append(i)
    order = sorted(units.items(), key=lambda kv: (-kv[1][0], kv[0]))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    out: List[List[int]] = [[] for _ in range(world)]
    for _, (c, idx) in order:
        load, r = heapq.heappop(heap)
        out[r].extend(idx)
        heapq.heappush(heap, (load + c, r))
    for r in range(world):
        out[r].sort()
    return out


def gather_results(local: dict, world: int, rank: int, dist=None) -> dict:
    """Merge per-rank {tile_index: result} dicts on every rank (control plane only:
    Python objects over the process group, never the DP data)."""
    if dist is None or world == 1:
        return dict(local)
    parts = [None] * world
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
