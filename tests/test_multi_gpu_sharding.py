"""The N > 1 path: 16x16 sample tiles dealt round-robin to ranks, per-rank films summed with one reduce.
Runs with world_size 2 and 8 on CPU (gloo). The per-rank compute here is the CPU oracle (tile_rank / tile_world have
the same meaning in PtRenderParams for both back ends); on the GPU box the same plumbing runs over RCCL in bench.py."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from _pkg import import_pkg
    from oracle.oracle_binding import Oracle
    pkg = import_pkg()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd, rp = pkg.scenes.ganesha_scale(n=12, xres=80, yres=48, spp=4).world_end()
    rp.tile_rank, rp.tile_world = rank, world
    s = Oracle(pkg._abi, pkg.runtime.TABLES_PATH).scene(sd)
    film = torch.from_numpy(s.render(rp, nthreads=1))
    own = (film[..., 3] > 0).numpy().copy()
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    masks = [torch.zeros_like(torch.from_numpy(own.astype(np.uint8))) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.from_numpy(own.astype(np.uint8)), masks, dst=0)
    if rank == 0:
        np.savez(out_path, film=film.numpy(), masks=np.stack([m.numpy() for m in masks]))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_tile_sharding_gloo(tmp_path, pkg, oracle, world):
    """world 2, and 8 -- the world size of the driver's SCALE run -- as gloo processes on the CPU: every pixel owned by exactly one rank, ownership = tile index % world, the
    reduced film bit-identical to one render (box filter: tiles map to disjoint pixels)."""
    out = str(tmp_path / "film.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = np.load(out)
    sd, rp = pkg.scenes.ganesha_scale(n=12, xres=80, yres=48, spp=4).world_end()
    full = oracle.scene(sd).render(rp, nthreads=1)
    masks = r["masks"].astype(np.int64)
    assert (masks.sum(axis=0) == 1).all()               # disjoint and complete
    # tile t belongs to rank t % world (tiles are 16x16, 5 per row here: 15 tiles)
    owner = np.zeros((48, 80), np.int64)
    for ty in range(3):
        for tx in range(5):
            owner[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = (ty * 5 + tx) % world
    for rk in range(world):
        assert np.array_equal(masks[rk], (owner == rk).astype(np.int64)), rk
    assert np.array_equal(r["film"], full)              # disjoint pixels: the sum is bit-identical to one render
