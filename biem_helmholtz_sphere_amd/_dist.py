"""Batch sharding of independent (k, eta, incidence) systems over the GPUs of one node.

The path has no exchange inside a solve (reference _biem.py:797 solves per leading batch index), so ranks
only marshal inputs and outputs: geometry is broadcast from rank 0, every rank solves a contiguous block of the batch,
densities are all-gathered.  One process per GPU; backend "nccl" is RCCL over xGMI on MI355X, "gloo" works on CPU (tests).
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "shard_sizes", "broadcast_geometry", "gather_batch", "biem_sharded"]


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of a batch of n systems owned by `rank`; the remainder goes to the first ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(n: int, world: int) -> Sequence[int]:
    return [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]


def broadcast_geometry(centers: Optional[torch.Tensor], radii: Optional[torch.Tensor], device: torch.device, src: int = 0, group=None):
    """Rank `src` holds centers[B, d] and radii[B]; every rank returns copies (a few hundred bytes over xGMI)."""
    rank = dist.get_rank(group)
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        meta[0], meta[1] = centers.shape[0], centers.shape[1]
    dist.broadcast(meta, src, group=group)
    B, d = int(meta[0]), int(meta[1])
    c = centers.to(device=device, dtype=torch.float64).contiguous() if rank == src else torch.empty((B, d), dtype=torch.float64, device=device)
    r = radii.to(device=device, dtype=torch.float64).contiguous() if rank == src else torch.empty((B,), dtype=torch.float64, device=device)
    dist.broadcast(c, src, group=group)
    dist.broadcast(r, src, group=group)
    return c, r


def gather_batch(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank blocks local[n_r, ...] (n_r = shard size of the rank) into full[n_total, ...] on every rank."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    mx = max(sizes) if sizes else 0
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if pad.is_complex():
        buf = torch.view_as_real(pad).contiguous()
    else:
        buf = pad.contiguous()
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    parts = []
    for r, o in enumerate(outs):
        o = torch.view_as_complex(o) if local.is_complex() else o
        parts.append(o[: sizes[r]])
    return torch.cat(parts, 0)


def biem_sharded(
    c: Any,
    *,
    centers: Optional[torch.Tensor],
    radii: Optional[torch.Tensor],
    k: torch.Tensor,
    n_end: int,
    eta: Optional[torch.Tensor] = None,
    incident: Optional[Callable[[torch.Tensor, slice], Tuple[Optional[Callable], Optional[Callable]]]] = None,
    device: Optional[torch.device] = None,
    group=None,
    solver: Optional[Callable[..., Any]] = None,
    **biem_kwargs: Any,
):
    """Solve a batch k[nb] (same on every rank) of systems that share one geometry, sharded over the ranks of `group`.

    centers/radii need only be valid on rank 0 (they are broadcast).  `incident(k_local, index_slice)` returns the
    `(uin, uin_grad)` pair for this rank's block (e.g. ``plane_wave(k=k_local, direction=...)``).  Returns
    ``(local_result, density_full)``: the rank's own BIEMResultCalculator and the all-gathered density [nb, B, H].
    `solver` defaults to :func:`biem_helmholtz_sphere_amd.biem` (injectable for CPU tests of the marshalling).
    """
    if solver is None:
        from ._biem import biem as solver  # the HIP path; raises loudly without a GPU
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None:
        device = k.device
    cen, rad = broadcast_geometry(centers, radii, device, 0, group)
    nb = int(k.shape[0])
    lo, hi = shard_bounds(nb, rank, world)
    k_loc = k[lo:hi].to(device)
    eta_loc = None if eta is None else eta[lo:hi].to(device)
    uin = ugr = None
    if incident is not None:
        uin, ugr = incident(k_loc, slice(lo, hi))
    res = solver(c, centers=cen[None], radii=rad[None], k=k_loc, eta=eta_loc, n_end=n_end, uin=uin, uin_grad=ugr, **biem_kwargs)
    dens = res.density
    full = None
    if dens is not None:
        dens = dens if isinstance(dens, torch.Tensor) else torch.as_tensor(dens)
        full = gather_batch(dens.to(device), nb, group)
    return res, full
