"""CPU tests of the multi-GPU path: more than one rank for real, without a GPU.

The collectives live behind the C-ABI (csrc/dist.cpp).  Here 2 and 3 processes meet over the library's own rendezvous (abstract
unix socket named after MASTER_PORT) and run the host-staged "shm" transport on host buffers: every collective the sharded paths
use (in-place all-gather of frame shares, fp16 / fp32 all-reduce of gradients, min / max of macrocell ranges, broadcast of the
replica state, reduce-scatter), the host control plane the bench uses, and the sharding maths.  A world-size-2 `gloo` group
(torch.distributed) computes the same collectives independently as a cross-check of their semantics.  The same code paths run on
the GPU with 2 and 4 ranks on one device in tests/test_gpu_dist.py."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from instantvnr_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(target, world, *args, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    [p.start() for p in procs]
    try:
        res = [q.get(timeout=timeout) for _ in procs]
    finally:
        [p.join(timeout=60) for p in procs]
        [p.kill() for p in procs if p.is_alive()]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return sorted(res, key=lambda r: r[0])


def _env(rank, world, port):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "VNR_AMD_DIST_TRANSPORT": "shm", "VNR_AMD_SHM_SLOT_MB": "1",
                       "VNR_AMD_DIST_TIMEOUT": "60"})


def _ptr(a):
    import ctypes as C
    return a.ctypes.data_as(C.c_void_p)


def _rank_data(rank, n, seed=0):
    return np.random.default_rng(1000 * seed + rank).normal(size=n).astype(np.float32)


# ------------------------------------------------------------------------------------------------ the transport's collectives
def _collectives_worker(rank, world, port, q):
    _env(rank, world, port)
    from instantvnr_amd._lib import check, lib
    L = lib()
    ctx = vdist.init_from_env()
    assert (ctx.rank, ctx.world, ctx.transport) == (rank, world, "shm")
    out = {}
    n = 700_001   # > one 1 MiB slot of fp32: the chunk loop runs, and the last chunk is ragged
    # all-reduce fp32 sum / max / min, in place
    for name, op in (("sum", vdist.SUM), ("max", vdist.MAX), ("min", vdist.MIN)):
        a = _rank_data(rank, n)
        check(L.vnrAmdDistAllReduce(_ptr(a), a.size, vdist.F32, op))
        out["f32_" + name] = a
    # fp16 sum (the gradient payload)
    h = (_rank_data(rank, n, 1) * 1e-2).astype(np.float16)
    check(L.vnrAmdDistAllReduce(_ptr(h), h.size, vdist.F16, vdist.SUM))
    out["f16_sum"] = h
    # fp16 mean: the data-parallel gradient exchange (DistOp::Avg = ncclAvg)
    h2 = (_rank_data(rank, n, 1) * 1e-2).astype(np.float16)
    check(L.vnrAmdDistAllReduce(_ptr(h2), h2.size, vdist.F16, vdist.AVG))
    out["f16_avg"] = h2
    # u8 max
    u = np.random.default_rng(50 + rank).integers(0, 255, 4096, dtype=np.uint8)
    check(L.vnrAmdDistAllReduce(_ptr(u), u.size, vdist.U8, vdist.MAX))
    out["u8_max"] = u
    # all-gather, in place: the rank's share already sits in its slot (what the renderer does)
    share = 300_017
    g = np.full(world * share, -1.0, np.float32)
    g[rank * share:(rank + 1) * share] = _rank_data(rank, share, 2)
    import ctypes as C
    mine = C.c_void_p(g.ctypes.data + rank * share * 4)
    check(L.vnrAmdDistAllGather(mine, _ptr(g), share * 4))
    out["gather"] = g
    # reduce-scatter: rank r owns slice r of the sum
    per = 123_457
    rs = _rank_data(rank, world * per, 3)
    check(L.vnrAmdDistReduceScatter(_ptr(rs), per, vdist.F32))
    out["rs_mine"] = rs[rank * per:(rank + 1) * per].copy()
    # broadcast from the last rank
    b = _rank_data(rank, 5000, 4)
    check(L.vnrAmdDistBroadcast(_ptr(b), b.nbytes, world - 1))
    out["bcast"] = b
    # host control plane
    out["host_max"] = vdist.all_reduce_host([float(rank), 10.0 - rank], vdist.MAX)
    out["host_sum"] = vdist.all_reduce_host([1.0, float(rank)], vdist.SUM)
    vdist.barrier()
    vdist.finalize()
    q.put((rank, out))


@pytest.mark.parametrize("world", [2, 3, 8])   # 8: the world size of the driver's scaling run (one process per GPU of a node)
def test_shm_transport_collectives(world):
    res = _spawn(_collectives_worker, world)
    n = 700_001
    data = [_rank_data(r, n) for r in range(world)]
    want = {"f32_sum": data[0].copy(), "f32_max": np.max(data, 0), "f32_min": np.min(data, 0)}
    for r in range(1, world):
        want["f32_sum"] = want["f32_sum"] + data[r]   # rank order, fp32: what the transport promises
    h = [(_rank_data(r, n, 1) * 1e-2).astype(np.float16) for r in range(world)]
    acc = h[0].astype(np.float32)
    for r in range(1, world):
        acc = acc + h[r].astype(np.float32)
    want["f16_sum"] = acc.astype(np.float16)           # summed in fp32, rounded once
    want["f16_avg"] = (acc * np.float32(1.0 / world)).astype(np.float16)   # ... divided in fp32 before the one rounding
    want["u8_max"] = np.max([np.random.default_rng(50 + r).integers(0, 255, 4096, dtype=np.uint8) for r in range(world)], 0)
    share, per = 300_017, 123_457
    want["gather"] = np.concatenate([_rank_data(r, share, 2) for r in range(world)])
    rs = [_rank_data(r, world * per, 3) for r in range(world)]
    rs_sum = rs[0].copy()
    for r in range(1, world):
        rs_sum = rs_sum + rs[r]
    want["bcast"] = _rank_data(world - 1, 5000, 4)
    for rank, out in res:
        for k in ("f32_sum", "f32_max", "f32_min", "f16_sum", "f16_avg", "u8_max", "gather", "bcast"):
            assert np.array_equal(out[k], want[k]), (rank, k)
        assert np.array_equal(out["rs_mine"], rs_sum[rank * per:(rank + 1) * per]), rank
        assert out["host_max"] == [float(world - 1), 10.0]
        assert out["host_sum"] == [float(world), float(sum(range(world)))]


# ------------------------------------------------------------------------------------------------ the same collectives through gloo
def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = torch.from_numpy(_rank_data(rank, 10_000))
        s = a.clone(); td.all_reduce(s, op=td.ReduceOp.SUM)
        m = a.clone(); td.all_reduce(m, op=td.ReduceOp.MAX)
        parts = [torch.empty_like(a) for _ in range(world)]
        td.all_gather(parts, a)
        q.put((rank, {"sum": s.numpy().copy(), "max": m.numpy().copy(), "gather": torch.cat(parts).numpy().copy()}))
    finally:
        td.destroy_process_group()


def _shm_small_worker(rank, world, port, q):
    _env(rank, world, port)
    import ctypes as C
    from instantvnr_amd._lib import check, lib
    L = lib()
    vdist.init_from_env()
    a = _rank_data(rank, 10_000)
    s = a.copy(); check(L.vnrAmdDistAllReduce(_ptr(s), s.size, vdist.F32, vdist.SUM))
    m = a.copy(); check(L.vnrAmdDistAllReduce(_ptr(m), m.size, vdist.F32, vdist.MAX))
    g = np.empty(world * a.size, np.float32)
    check(L.vnrAmdDistAllGather(_ptr(a), _ptr(g), a.nbytes))
    vdist.finalize()
    q.put((rank, {"sum": s, "max": m, "gather": g}))


def test_world2_collectives_agree_with_gloo():
    """an independent implementation (torch.distributed, gloo backend, world size 2) of the collectives the sharded paths use:
    same inputs, same results (a two-term fp32 sum has one rounding, so even the sums are bitwise equal)"""
    pytest.importorskip("torch")
    ours = _spawn(_shm_small_worker, 2)
    theirs = _spawn(_gloo_worker, 2)
    for (r0, a), (r1, b) in zip(ours, theirs):
        assert r0 == r1
        for k in ("sum", "max", "gather"):
            assert np.array_equal(a[k], b[k]), (r0, k)


# ------------------------------------------------------------------------------------------------ tiles
def _share_standin(width, height, world, rank):
    """what a rank's renderer leaves in ITS slot of the gathered buffer: its tile rows, packed, value = f(global pixel)"""
    n, block = width * height, 8 * width
    idx = np.arange(n)
    vals = np.stack([idx, idx % width, idx // width, np.ones(n)], 1).astype(np.float32)
    return vdist.pack_share(vals, block, world, rank, n), vals


def _tiles_worker(rank, world, port, q, width, height):
    _env(rank, world, port)
    import ctypes as C
    from instantvnr_amd._lib import check, lib
    L = lib()
    vdist.init_from_env()
    n, block = width * height, 8 * width
    _, _, n_local = vdist.interleave_layout(n, block, world)
    share, want = _share_standin(width, height, world, rank)
    gathered = np.full((world, n_local, 4), -7.0, np.float32)
    gathered[rank] = share
    mine = C.c_void_p(gathered.ctypes.data + rank * n_local * 16)
    check(L.vnrAmdDistAllGather(mine, _ptr(gathered), n_local * 16))      # in place, as Renderer::issue_gather does
    full = vdist.assemble_shares(gathered, block, world, n)
    # local index -> global pixel mapping of the kernels (map_pixel / write_pixel) agrees with the packing
    for i in (0, 1, block - 1, block, n_local - 1):
        g = vdist.local_to_global(i, block, world, rank)
        if g < n:
            assert float(share[i, 0]) == float(g)
    vdist.finalize()
    q.put((rank, bool(np.array_equal(full, want))))


@pytest.mark.parametrize("world,width,height", [(2, 64, 48), (2, 40, 36), (3, 24, 100), (8, 1024, 1024), (8, 200, 72)])  # even division, ragged height, ragged blocks, the bench frame on a node, fewer bands than twice the ranks
def test_frame_shares_gather_and_assemble(world, width, height):
    for rank, ok in _spawn(_tiles_worker, world, width, height):
        assert ok, rank


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


# ------------------------------------------------------------------------------------------------ gradient exchange
def _grad_worker(rank, world, port, q):
    """the data-parallel step's arithmetic on a synthetic gradient: per-rank fp32 gradient (loss-scaled, normalised by the local
    batch), exchanged once as fp32 and once as the fp16 payload train_data_parallel sends, then the oracle's Adam step"""
    _env(rank, world, port)
    from instantvnr_amd._lib import check, lib
    from oracle import train_oracle as T
    L = lib()
    vdist.init_from_env()
    n, n_matrix, batch = 40_000, 4096, 64
    rng = np.random.default_rng(7)                       # identical replicas
    master = rng.normal(size=n).astype(np.float32) * 0.1
    m0 = rng.normal(size=n).astype(np.float64) * 1e-4
    v0 = np.abs(rng.normal(size=n)).astype(np.float64) * 1e-7 + 1e-9   # a schedule in progress (a first step is lr * sign(g) for any g)
    steps = np.full(n, 3.0)
    per_sample = np.random.default_rng(100 + rank).normal(size=(batch, n)).astype(np.float32)
    per_sample[:, n_matrix:] *= (np.random.default_rng(200 + rank).random(n - n_matrix) < 0.3)[None, :]   # a batch touches a part of the grid entries
    local = (T.LOSS_SCALE * per_sample.sum(0) / batch).astype(np.float32) * np.float32(1e-2)
    g32 = local.copy()
    check(L.vnrAmdDistAllReduce(_ptr(g32), n, vdist.F32, vdist.SUM))
    g16 = local.astype(np.float16)
    check(L.vnrAmdDistAllReduce(_ptr(g16), n, vdist.F16, vdist.SUM))
    new32 = T.adam_step(master.astype(np.float64), g32, m0, v0, steps, n_matrix, grad_scale=1.0 / world)[0]
    new16 = T.adam_step(master.astype(np.float64), g16.astype(np.float32), m0, v0, steps, n_matrix, grad_scale=1.0 / world)[0]
    # gradient of the concatenated batch, computed from everyone's samples
    allp = np.empty((world,) + per_sample.shape, np.float32)
    check(L.vnrAmdDistAllGather(_ptr(per_sample), _ptr(allp), per_sample.nbytes))
    concat = T.LOSS_SCALE * allp.reshape(-1, n).astype(np.float64).sum(0) / (batch * world) * 1e-2
    vdist.finalize()
    q.put((rank, {"g32": g32, "new32": new32, "new16": new16, "concat": concat, "master": master}))


def test_world2_fp16_gradient_exchange_equals_fp32_exchange():
    """sum over ranks x 1 / world is the gradient of the concatenated batch; the update made from the fp16 payload equals the
    update made from an fp32 exchange within 2^-10 of a parameter's magnitude (the resolution of the fp16 parameter itself),
    and every rank computes the same bits"""
    res = _spawn(_grad_worker, 2)
    (r0, a), (r1, b) = res
    for k in ("g32", "new32", "new16"):
        assert np.array_equal(a[k], b[k]), k                         # identical update on every rank
    assert np.allclose(a["g32"] / 2.0, a["concat"], rtol=1e-5, atol=1e-7)
    moved = np.abs(a["new32"] - a["master"])
    assert moved.max() > 1e-4                                          # the step did something
    err = np.abs(a["new16"] - a["new32"])
    assert err.max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(a["master"])).max()
    assert err.max() <= 2.0 ** -10, err.max()
    # entries nobody touched stay untouched through either exchange
    untouched = (a["g32"] == 0)
    untouched[:4096] = False
    assert untouched.any() and np.array_equal(a["new16"][untouched], a["master"][untouched].astype(np.float64))


# ------------------------------------------------------------------------------------------------ failure behaviour
def _lonely_worker(rank, world, port, q):
    _env(rank, world, port)
    os.environ["VNR_AMD_DIST_TIMEOUT"] = "2"
    from instantvnr_amd._lib import VnrAmdError
    try:
        vdist.init_from_env()
        q.put((rank, "initialised"))
    except VnrAmdError as e:
        q.put((rank, str(e)))


def test_a_missing_rank_is_an_error_not_a_hang():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_lonely_worker, args=(1, 2, _free_port(), q))   # rank 1 of 2, rank 0 never starts
    p.start()
    rank, msg = q.get(timeout=60)
    p.join(timeout=30)
    assert "cannot reach rank 0" in msg, msg


def test_collectives_without_init_fail_loudly():
    from instantvnr_amd._lib import lib
    a = np.zeros(4, np.float32)
    assert lib().vnrAmdDistAllReduce(_ptr(a), 4, vdist.F32, vdist.SUM) != 0
    assert b"not initialised" in lib().vnrAmdGetLastError()
