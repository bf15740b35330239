"""One process per GPU: shard independent (instruction, pose) items across ranks, all-gather the per-item rows.

The contact path shards embarrassingly: every force_single_elastic_intersection!
(src/contact_algorithms_non_friction.jl:70-84) reads immutable meshes plus its own pose / twist / state and yields an
independent wrench; the coupling (f_generalized += J' w, :283-284) happens afterwards on the host.  So meshes and trees
are replicated on every GPU, items are partitioned, and the only exchange is ONE all-gather per evaluation of the
per-item result rows [wrench 6 | sdot 6 | counts 4] (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU in the
tests).  Rows are a few hundred KB at most (C5: 2 016 x 128 B), i.e. latency-bound: one fused collective, no buckets.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

ROW = 16      # wrench 6 + sdot 6 + counts 4 (stored as float64; counts < 2^53 are exact)


def shard_block(n_items: int, world: int) -> List[np.ndarray]:
    """Contiguous, near-equal blocks (C4: 256 scenes -> 32 per GPU)."""
    bounds = [(n_items * r) // world for r in range(world + 1)]
    return [np.arange(bounds[r], bounds[r + 1], dtype=np.int64) for r in range(world)]


def shard_by_cost(cost: Sequence[float], world: int) -> List[np.ndarray]:
    """Cost-weighted partition (C5: cost ~ n_leaf_1 * n_leaf_2 or the previous step's candidate count): longest
    processing time first onto the least loaded rank; deterministic (ties by index)."""
    cost = np.asarray(cost, dtype=np.float64)
    order = np.lexsort((np.arange(cost.size), -cost))
    load = np.zeros(world)
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        parts[r].append(int(i))
        load[r] += cost[i]
    return [np.sort(np.asarray(p, dtype=np.int64)) for p in parts]


def pack_rows(wrench, sdot, counts) -> torch.Tensor:
    w = torch.as_tensor(wrench, dtype=torch.float64)
    rows = torch.zeros((w.shape[0], ROW), dtype=torch.float64, device=w.device)
    rows[:, 0:6] = w
    rows[:, 6:12] = torch.as_tensor(sdot, dtype=torch.float64, device=w.device)
    rows[:, 12:16] = torch.as_tensor(counts, device=w.device).to(torch.float64)
    return rows


def unpack_rows(rows: torch.Tensor):
    r = rows.cpu().numpy()
    return r[:, 0:6].copy(), r[:, 6:12].copy(), np.rint(r[:, 12:16]).astype(np.int32)


def all_gather_rows(local_rows: torch.Tensor, parts: List[np.ndarray], n_items: int,
                    group: Optional[dist.ProcessGroup] = None, force_collective: bool = False) -> torch.Tensor:
    """All-gather the ranks' result rows into item order.  parts[r] = item indices owned by rank r (every rank
    knows the whole assignment, it is a pure function of the inputs).  force_collective: issue the collective even in a
    process group of one rank (rehearsal of the RCCL path on a one-GPU box)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        out = torch.zeros((n_items, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        out[torch.as_tensor(parts[0], device=local_rows.device)] = local_rows
        return out
    n_max = max(int(p.size) for p in parts)
    pad = torch.zeros((n_max, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    gathered = torch.zeros((world * n_max, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    out = torch.zeros((n_items, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    for r, p in enumerate(parts):
        if p.size:
            out[torch.as_tensor(p, device=local_rows.device)] = gathered[r * n_max: r * n_max + p.size]
    return out


class RowExchange:
    """The per-evaluation exchange with everything that does not change from one evaluation to the next done once: the
    padded send block, the receive block, and ONE index vector that brings the gathered rows into item order (a single
    gather instead of one scatter per rank, no host-to-device copy of indices per call).  On device tensors the call is
    pack (wrench | sdot | counts -> rows of 16 doubles) + collective + gather: about five launches.

    parts[r] = item indices owned by rank r (every rank knows the whole assignment)."""

    def __init__(self, parts: List[np.ndarray], n_items: int, device, group: Optional[dist.ProcessGroup] = None,
                 force_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(parts) == self.world, "one index set per rank"
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.n_items = int(n_items)
        self.n_mine = int(parts[self.rank].size)
        self.n_max = max(max(int(p.size) for p in parts), 1)
        self.device = torch.device(device)
        src = np.zeros(self.n_items, dtype=np.int64)
        covered = np.zeros(self.n_items, dtype=bool)
        for r, p in enumerate(parts):
            src[p] = r * self.n_max + np.arange(p.size)
            covered[p] = True
        if not covered.all():
            raise ValueError("the partition does not cover every item")
        self.src = torch.as_tensor(src, device=self.device)
        self.send = torch.zeros((self.n_max, ROW), dtype=torch.float64, device=self.device)
        self.recv = torch.zeros((self.world * self.n_max, ROW), dtype=torch.float64, device=self.device)

    def __call__(self, wrench: torch.Tensor, sdot: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
        """wrench, sdot (n_mine, 6) float64, counts (n_mine, 4) int32 of the rank's own items, on the plan's device ->
        (n_items, 16) rows [wrench | sdot | counts] in item order."""
        n = self.n_mine
        if n:
            self.send[:n, 0:6] = wrench[:n]
            self.send[:n, 6:12] = sdot[:n]
            self.send[:n, 12:16] = counts[:n]          # int32 -> float64 in the copy (counts < 2^53 are exact)
        if self.collective:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            return self.recv.index_select(0, self.src)
        return self.send.index_select(0, self.src)


def evaluate_sharded(evaluator: Callable[[np.ndarray], tuple], n_items: int, parts: Optional[List[np.ndarray]] = None,
                     device: Optional[torch.device] = None):
    """Evaluate all items across the ranks of the default process group.

    evaluator(item_indices) -> (wrench (k,6), sdot (k,6), counts (k,4)) for the rank's own items; on a GPU rank it is
    MechanismScenario.force_all_elastic_intersections over the selected poses, in the CPU tests it is the oracle.
    Returns the full (wrench, sdot, counts) in item order on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if parts is None:
        parts = shard_block(n_items, world)
    mine = parts[rank]
    if mine.size:
        w, s, c = evaluator(mine)
    else:
        w, s, c = np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((0, 4), dtype=np.int32)
    rows = pack_rows(w, s, c)
    if device is not None:
        rows = rows.to(device)
    return unpack_rows(all_gather_rows(rows, parts, n_items))
