"""world_size-2 gloo test of the batch-sharding marshalling (broadcast geometry, contiguous shards, all-gather)."""
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


def _fake_solver(c, *, centers, radii, k, eta, n_end, uin, uin_grad, **kw):
    """Deterministic stand-in for biem(): density[s, b, h] = k_s * (b + 1) + 1j * (h + radii_b + centers_b0)."""
    B = radii.shape[-1]
    H = n_end * n_end
    b = torch.arange(B, dtype=torch.float64)[None, :, None]
    h = torch.arange(H, dtype=torch.float64)[None, None, :]
    re = k[:, None, None] * (b + 1) + 0 * h
    im = h + radii.reshape(1, B, 1) + centers.reshape(1, B, -1)[:, :, :1] + 0 * re
    shift = 0.0 if uin is None else float(uin(torch.zeros(1)))
    return _FakeResult(torch.complex(re + shift, im))


def _worker(rank, world, port, nb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = torch.linspace(0.5, 8.0, nb, dtype=torch.float64)
        centers = torch.tensor([[0.0, 2.0, 0.0], [1.0, -2.0, 0.0], [5.0, 0.0, 0.0]], dtype=torch.float64) if rank == 0 else None
        radii = torch.tensor([1.0, 0.5, 0.25], dtype=torch.float64) if rank == 0 else None
        seen = {}

        def incident(k_loc, sl):
            seen["slice"] = (sl.start, sl.stop, tuple(k_loc.shape))
            return (lambda x: torch.tensor(100.0)), None

        res, full = _dist.biem_sharded(object(), centers=centers, radii=radii, k=k, n_end=2, incident=incident,
                                       device=torch.device("cpu"), solver=_fake_solver)
        lo, hi = _dist.shard_bounds(nb, rank, world)
        assert seen["slice"] == (lo, hi, (hi - lo,))
        assert res.density.shape == (hi - lo, 3, 4)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nb", [5, 8])
def test_sharded_solve_gathers_in_batch_order(nb):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    k = torch.linspace(0.5, 8.0, nb, dtype=torch.float64)
    cen = torch.tensor([[0.0, 2.0, 0.0], [1.0, -2.0, 0.0], [5.0, 0.0, 0.0]], dtype=torch.float64)
    rad = torch.tensor([1.0, 0.5, 0.25], dtype=torch.float64)
    expect = _fake_solver(None, centers=cen[None], radii=rad[None], k=k, eta=None, n_end=2, uin=lambda x: torch.tensor(100.0), uin_grad=None).density.numpy()
    for r in range(world):
        assert outs[r].shape == (nb, 3, 4)
        assert np.array_equal(outs[r], expect)
