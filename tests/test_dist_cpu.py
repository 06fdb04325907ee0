"""World-size-2 `gloo` tests (CPU) of the multi-GPU path: interleaved tile sharding + frame gather, and the
data-parallel gradient exchange.  The collectives and the sharding maths are the real ones from
instantvnr_amd/dist.py; only the GPU kernels are replaced by deterministic stand-ins."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instantvnr_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_share_standin(width, height, block, world, rank):
    """what a rank's renderer leaves in its framebuffer: its own pixels = f(global pixel index), others untouched"""
    n = width * height
    frame = torch.full((n, 4), -1.0)
    idx = torch.arange(n)
    mine = ((idx // block) % world) == rank
    vals = torch.stack([idx.float(), (idx % width).float(), (idx // width).float(), torch.ones(n)], 1)
    frame[mine] = vals[mine]
    return frame, vals


def _worker(rank, world, port, width, height, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = width * height
        block = 8 * width
        # ---- render path: pack own interleaved share, all_gather, assemble ---------------------------------
        frame, want = _render_share_standin(width, height, block, world, rank)
        share = vdist.pack_share(frame, block, world, rank, n)
        _, per_part, n_local = vdist.interleave_layout(n, block, world)
        assert share.shape == (n_local, 4)
        gathered = torch.empty((world, n_local, 4))
        dist.all_gather_into_tensor(gathered.view(-1), share.view(-1))
        full = vdist.assemble_shares(gathered, block, world, n)
        ok_render = bool(torch.equal(full, want))
        # the three-operation path of ShardedRenderer (blocks divide evenly): a strided view instead of pack, one strided copy
        # instead of assemble.  It must exist exactly when the division is even and give the same frame.
        view = vdist.share_view(frame, block, world, rank)
        if (height % (8 * world)) == 0:
            share2 = torch.empty((n_local, 4))
            share2.view(view.shape).copy_(view)
            ok_render = ok_render and bool(torch.equal(share2, share))
            gathered2 = torch.empty((world, n_local, 4))
            dist.all_gather_into_tensor(gathered2.view(-1), share2.view(-1))
            full2 = vdist.assemble_shares_into(torch.empty((n, 4)), gathered2, block, world)
            ok_render = ok_render and bool(torch.equal(full2, want))
        else:
            ok_render = ok_render and view is None
        # local index -> global pixel mapping used by the kernels agrees with the packing
        for i in (0, 1, block - 1, block, n_local - 1):
            g = vdist.local_to_global(i, block, world, rank)
            if g < n:
                assert float(share[i, 0]) == float(g)
        # ---- training path: sum all-reduce + 1/world == gradient of the concatenated batch -----------------
        rng = np.random.default_rng(100 + rank)
        per_sample = torch.from_numpy(rng.normal(size=(64, 1000)).astype(np.float32))  # per-sample gradients
        local_grad = per_sample.sum(0) / 64.0            # loss normalised by the local batch (tcnn L1/L2 loss)
        g = local_grad.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g *= 1.0 / world                                  # grad_scale passed to vnrAmdNeuralVolumeTrainEnd
        allp = [torch.empty_like(per_sample) for _ in range(world)]
        dist.all_gather(allp, per_sample)
        want_g = torch.cat(allp, 0).sum(0) / (64.0 * world)
        ok_train = bool(torch.allclose(g, want_g, atol=1e-6))
        # identical update on every rank (bitwise identical all-reduce result)
        chk = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(chk, g)
        ok_same = all(torch.equal(chk[0], c) for c in chk)
        q.put((rank, ok_render, ok_train, ok_same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("width,height", [(64, 48), (40, 36)])  # first: blocks divide evenly; second: height not a multiple of 8 x world
def test_world2_gloo_tiles_and_gradients(width, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, width, height, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok_render, ok_train, ok_same in res:
        assert ok_render and ok_train and ok_same, (rank, ok_render, ok_train, ok_same)


def test_interleave_layout_covers_every_pixel_once():
    for (w, h, world) in [(1024, 1024, 8), (40, 36, 2), (64, 8, 4), (24, 100, 3)]:
        n, block = w * h, 8 * w
        seen = np.zeros(n, np.int32)
        _, per_part, n_local = vdist.interleave_layout(n, block, world)
        for part in range(world):
            i = np.arange(n_local)
            g = (i // block * world + part) * block + i % block
            g = g[g < n]
            seen[g] += 1
        assert np.all(seen == 1)
