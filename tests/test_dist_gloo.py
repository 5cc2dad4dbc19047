"""world_size-2 gloo tests of the batch-sharding marshalling (broadcast geometry, contiguous shards of the flattened batch,
all-gather): 1-D batches, uneven shards with eta, a 2-D (k, eta) batch like BASELINE config 5's, per-system geometry and alpha / beta."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from biem_helmholtz_sphere_amd import _dist


class _FakeResult:
    def __init__(self, density):
        self.density = density


def _fake_solver(c, *, centers, radii, k, eta, n_end, uin, uin_grad, alpha=1.0, beta=0.0, **kw):
    """Deterministic stand-in for biem() on a 1-D batch: density[s, b, h] = k_s (b + 1) + 10 eta_s + shift
    + 1j (h + radii_sb + centers_sb0 + 1000 alpha_sb + 10000 beta_sb), shift = uin(0)."""
    assert k.ndim == 1 and centers.ndim == 3 and radii.ndim == 2
    n = k.shape[0]
    B = radii.shape[-1]
    H = n_end * n_end
    b = torch.arange(B, dtype=torch.float64)[None, :, None]
    h = torch.arange(H, dtype=torch.float64)[None, None, :]
    e = torch.ones(n, dtype=torch.float64) if eta is None else eta
    assert e.shape == (n,)
    per_ball = lambda t: torch.as_tensor(t, dtype=torch.float64).expand(n, B)[:, :, None] if isinstance(t, torch.Tensor) else torch.full((n, B, 1), float(t), dtype=torch.float64)
    shift = 0.0 if uin is None else float(uin(torch.zeros(1)))
    re = k[:, None, None] * (b + 1) + 10.0 * e[:, None, None] + shift + 0 * h
    im = h + radii.expand(n, B)[:, :, None] + centers.expand(n, B, centers.shape[-1])[:, :, :1] + 1000.0 * per_ball(alpha) + 10000.0 * per_ball(beta) + 0 * re
    return _FakeResult(torch.complex(re, im))


_CEN = [[0.0, 2.0, 0.0], [1.0, -2.0, 0.0], [5.0, 0.0, 0.0]]
_RAD = [1.0, 0.5, 0.25]


def _case(name, nb):
    """Inputs of a case as every rank sees them (geometry is only USED from rank 0)."""
    cen = torch.tensor(_CEN, dtype=torch.float64)
    rad = torch.tensor(_RAD, dtype=torch.float64)
    if name == "1d":
        return dict(k=torch.linspace(0.5, 8.0, nb, dtype=torch.float64), eta=None, centers=cen, radii=rad, alpha=1.0, beta=0.0)
    if name == "1d_eta":
        return dict(k=torch.linspace(0.5, 8.0, nb, dtype=torch.float64), eta=torch.linspace(0.25, 4.0, nb, dtype=torch.float64), centers=cen, radii=rad,
                    alpha=1.0, beta=0.0)
    if name == "2d":      # config 5's shape in small: k (3, 1) x eta (1, 5) -> (3, 5) = 15 systems, 8 + 7 over two ranks
        return dict(k=torch.linspace(0.5, 4.0, 3, dtype=torch.float64)[:, None], eta=torch.linspace(0.25, 4.0, 5, dtype=torch.float64)[None, :],
                    centers=cen, radii=rad, alpha=1.0, beta=0.0)
    if name == "per_system":   # geometry and Robin coefficients per system, (2, 3) batch
        g = torch.arange(6, dtype=torch.float64).reshape(2, 3, 1, 1)
        return dict(k=torch.linspace(1.0, 2.0, 6, dtype=torch.float64).reshape(2, 3), eta=torch.tensor([[1.0], [2.0]], dtype=torch.float64),
                    centers=cen[None, None] + g, radii=rad[None, None] * (1.0 + 0.1 * g[..., 0]),
                    alpha=(1.0 + torch.arange(6, dtype=torch.float64).reshape(2, 3, 1)).expand(2, 3, 3).contiguous(),
                    beta=torch.full((1, 1, 3), 0.5, dtype=torch.float64))
    raise KeyError(name)


def _expected(name, nb):
    """The whole batch solved in one piece by the same stand-in (flattened by hand), reshaped to the batch shape."""
    kw = _case(name, nb)
    shapes = [tuple(kw["k"].shape), tuple(kw["centers"].shape[:-2]), tuple(kw["radii"].shape[:-1])]
    if kw["eta"] is not None:
        shapes.append(tuple(kw["eta"].shape))
    for t in (kw["alpha"], kw["beta"]):
        if isinstance(t, torch.Tensor):
            shapes.append(tuple(t.shape[:-1]))
    batch = tuple(torch.broadcast_shapes(*shapes))
    n = int(np.prod(batch)) if batch else 1
    fl = lambda t, tail: t.expand(batch + tuple(t.shape[t.ndim - tail:])).reshape((n,) + tuple(t.shape[t.ndim - tail:]))
    out = _fake_solver(None, centers=fl(kw["centers"], 2), radii=fl(kw["radii"], 1), k=fl(kw["k"], 0), eta=None if kw["eta"] is None else fl(kw["eta"], 0),
                       n_end=2, uin=lambda x: torch.tensor(100.0), uin_grad=None,
                       alpha=fl(kw["alpha"], 1) if isinstance(kw["alpha"], torch.Tensor) else kw["alpha"],
                       beta=fl(kw["beta"], 1) if isinstance(kw["beta"], torch.Tensor) else kw["beta"]).density
    return out.reshape(batch + tuple(out.shape[1:])).numpy(), n


def _worker(rank, world, port, name, nb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = _case(name, nb)
        if rank != 0:
            kw["centers"] = kw["radii"] = None          # only rank 0's geometry may be used
        seen = {}

        def incident(k_loc, sl):
            seen["slice"] = (sl.start, sl.stop, tuple(k_loc.shape))
            return (lambda x: torch.tensor(100.0)), None

        res, full = _dist.biem_sharded(object(), n_end=2, incident=incident, device=torch.device("cpu"), solver=_fake_solver, **kw)
        n_total = int(np.prod(full.shape[:-2]))
        lo, hi = _dist.shard_bounds(n_total, rank, world)
        assert seen["slice"] == (lo, hi, (hi - lo,))
        assert res.density.shape == (hi - lo, 3, 4)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,nb", [("1d", 5), ("1d", 8), ("1d_eta", 7), ("1d", 1), ("2d", 0), ("per_system", 0)])
def test_sharded_solve_gathers_in_batch_order(name, nb):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect, n = _expected(name, nb)
    if name == "2d":
        assert expect.shape[:2] == (3, 5)
    for r in range(world):
        assert outs[r].shape == expect.shape
        assert np.array_equal(outs[r], expect)
