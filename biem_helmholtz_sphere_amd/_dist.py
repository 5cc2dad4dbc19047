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
    """Rank `src` holds centers[..., B, d] and radii[..., B] (shared geometry: no leading axes; per-system geometry: the batch axes
    first); every rank returns copies (a few hundred bytes to a few hundred KB over xGMI)."""
    rank = dist.get_rank(group)
    meta = torch.zeros(18, dtype=torch.int64, device=device)          # ndim + up to 8 extents, for centers and for radii
    if rank == src:
        if centers.ndim > 8 or radii.ndim > 8:
            raise ValueError("broadcast_geometry: at most 8 axes")
        meta[0] = centers.ndim
        meta[1:1 + centers.ndim] = torch.tensor(list(centers.shape), dtype=torch.int64)
        meta[9] = radii.ndim
        meta[10:10 + radii.ndim] = torch.tensor(list(radii.shape), dtype=torch.int64)
    dist.broadcast(meta, src, group=group)
    m = meta.tolist()
    cshape, rshape = tuple(m[1:1 + m[0]]), tuple(m[10:10 + m[9]])
    c = centers.to(device=device, dtype=torch.float64).contiguous() if rank == src else torch.empty(cshape, dtype=torch.float64, device=device)
    r = radii.to(device=device, dtype=torch.float64).contiguous() if rank == src else torch.empty(rshape, dtype=torch.float64, device=device)
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


def _flat_block(t: torch.Tensor, batch: Tuple[int, ...], tail: int, lo: int, hi: int) -> torch.Tensor:
    """Rows lo..hi-1 of `t` (shape (*b, *tail axes), b broadcastable to `batch`) over the flattened batch; an operand whose batch
    axes all have extent 1 is shared by every system and comes back with ONE leading row."""
    lead = tuple(t.shape[: t.ndim - tail])
    tl = tuple(t.shape[t.ndim - tail:])
    if all(e == 1 for e in lead):
        return t.reshape((1,) + tl)
    lead = (1,) * (len(batch) - len(lead)) + lead
    nb = 1
    for e in batch:
        nb *= e
    return t.reshape(lead + tl).expand(tuple(batch) + tl).reshape((nb,) + tl)[lo:hi].contiguous()


def biem_sharded(
    c: Any,
    *,
    centers: Optional[torch.Tensor],
    radii: Optional[torch.Tensor],
    k: torch.Tensor,
    n_end: int,
    eta: Optional[torch.Tensor] = None,
    alpha: Any = 1.0,
    beta: Any = 0.0,
    incident: Optional[Callable[[torch.Tensor, slice], Tuple[Optional[Callable], Optional[Callable]]]] = None,
    device: Optional[torch.device] = None,
    group=None,
    solver: Optional[Callable[..., Any]] = None,
    **biem_kwargs: Any,
):
    """Solve a batch of independent systems sharded over the ranks of `group` (reference _biem.py:797 solves per leading batch
    index; nothing couples batch elements).

    The batch shape is the broadcast of k (...), eta (...), centers (..., B, d), radii (..., B) and alpha / beta (scalars or
    (..., B)) exactly as in ``biem()`` - e.g. k (32, 1) with eta (1, 16) is BASELINE config 5's (32, 16) batch of 512 systems.  The
    flattened batch (C order) is cut into contiguous blocks, one per rank (sizes differ by at most one; a rank may own none).
    k, eta, alpha, beta must be valid on every rank; centers / radii need only be valid on rank 0 (they are broadcast: shared geometry
    (B, d) / (B), or per-system geometry with the batch axes first).  `incident(k_local, index_slice)` returns the ``(uin, uin_grad)``
    pair for this rank's block: k_local is the 1-D tensor of the block's wavenumbers, index_slice its range in the FLATTENED batch
    (flat index i is ``numpy.unravel_index(i, batch_shape)``).  Returns ``(local_result, density_full)``: the rank's own
    BIEMResultCalculator (1-D batch of the block) and the all-gathered density of shape (*batch_shape, B, H) on every rank.
    `solver` defaults to :func:`biem_helmholtz_sphere_amd.biem` (injectable for CPU tests of the marshalling).
    """
    if solver is None:
        from ._biem import biem as solver  # the HIP path; raises loudly without a GPU
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None:
        device = k.device
    cen, rad = broadcast_geometry(centers, radii, device, 0, group)
    B, d = int(cen.shape[-2]), int(cen.shape[-1])
    k = torch.as_tensor(k)
    eta_t = None if eta is None else torch.as_tensor(eta)
    al = alpha if isinstance(alpha, torch.Tensor) else None
    be = beta if isinstance(beta, torch.Tensor) else None
    if (al is not None and al.ndim == 0) or (be is not None and be.ndim == 0):
        raise ValueError("alpha / beta: Python scalars or tensors of shape (..., B)")
    shapes = [tuple(k.shape), tuple(cen.shape[:-2]), tuple(rad.shape[:-1])]
    if eta_t is not None:
        shapes.append(tuple(eta_t.shape))
    for t in (al, be):
        if t is not None:
            shapes.append(tuple(t.shape[:-1]))
    batch = tuple(torch.broadcast_shapes(*shapes))
    nb = 1
    for e in batch:
        nb *= e
    lo, hi = shard_bounds(nb, rank, world)
    k_loc = _flat_block(k, batch, 0, 0, nb)
    k_loc = (k_loc.expand(nb) if k_loc.shape[0] == 1 and nb != 1 else k_loc)[lo:hi].to(device).contiguous()
    eta_loc = None
    if eta_t is not None:
        e_ = _flat_block(eta_t, batch, 0, 0, nb)
        eta_loc = (e_.expand(nb) if e_.shape[0] == 1 and nb != 1 else e_)[lo:hi].to(device).contiguous()
    cen_loc = _flat_block(cen, batch, 2, lo, hi)                 # (1 | n_r, B, d)
    rad_loc = _flat_block(rad, batch, 1, lo, hi)                 # (1 | n_r, B)
    kw = dict(biem_kwargs)
    kw["alpha"] = alpha if al is None else _flat_block(al.to(device), batch, 1, lo, hi)
    kw["beta"] = beta if be is None else _flat_block(be.to(device), batch, 1, lo, hi)
    uin = ugr = None
    if incident is not None:
        uin, ugr = incident(k_loc, slice(lo, hi))
    res = solver(c, centers=cen_loc, radii=rad_loc, k=k_loc, eta=eta_loc, n_end=n_end, uin=uin, uin_grad=ugr, **kw)
    dens = res.density
    full = None
    if dens is not None:
        dens = dens if isinstance(dens, torch.Tensor) else torch.as_tensor(dens)
        full = gather_batch(dens.to(device), nb, group)
        full = full.reshape(batch + tuple(full.shape[1:]))
    return res, full
