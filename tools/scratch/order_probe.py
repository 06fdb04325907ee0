"""GPU box (a -DVNR_DIAG build: VNR_AMD_DEBUG_MAX_ITERS stops the frame at the iteration whose queue is read): does the ORDER OF 64-SAMPLE CHUNKS in the real sample queue matter?  Takes the queue of march iteration ITER of the
bench frame and times the fused kernel on it (a) as it is, (b) with its 64-sample chunks sorted by the Morton code of their
centroid, (c) with 1024-sample runs (about one 64-ray group's claim) sorted the same way, (d) fully Morton sorted, (e) shuffled.
usage: python tools/order_probe.py [ITER ...]"""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
os.environ["VNR_AMD_RENDER_HALVES"] = "1"
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
size = 1024
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(size / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
cam = syn.oblique_camera(dims, distance_scale=1.1)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)


def morton(q):
    def part(x):
        x = (x | (x << 16)) & 0x030000FF0000FF
        x = (x | (x << 8)) & 0x0300F00F00F00F
        x = (x | (x << 4)) & 0x030C30C30C30C3
        x = (x | (x << 2)) & 0x09249249249249
        return x
    return part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)


def run(name, c):
    c = np.ascontiguousarray(c, np.float32)
    m = c.shape[0]
    d_c = api.DeviceArray.from_numpy(c); d_o = api.DeviceArray((m,), np.float32)
    for _ in range(3): check(L.vnrAmdNeuralVolumeInference(nv.h, m, d_c.ptr, d_o.ptr, None))
    check(L.vnrAmdSynchronize()); t0 = time.perf_counter()
    for _ in range(10): check(L.vnrAmdNeuralVolumeInference(nv.h, m, d_c.ptr, d_o.ptr, None))
    check(L.vnrAmdSynchronize()); dt = (time.perf_counter() - t0) / 10
    print(f"  {name:52s} n={m:9d} {dt*1e3:7.3f} ms {m/dt/1e6:8.1f} Msamples/s", flush=True)


def chunk_sorted(coords, chunk):
    m = (coords.shape[0] // chunk) * chunk
    c = coords[:m].reshape(-1, chunk, 3)
    cen = np.floor(c.mean(axis=1) * 1024).astype(np.int64).clip(0, 1023)
    order = np.argsort(morton(cen), kind="stable")
    return np.concatenate([c[order].reshape(-1, 3), coords[m:]])


def regroup_by_pixel(coords, tile_order):
    """what an ORDER-PRESERVING ray compaction would hand the kernel: alive rays in tile order (8x8 tiles, row-major or Morton
    over the tile grid), cut into groups of 64 rays, each group's samples counting-sorted by 8-voxel depth bins"""
    W = H = 1024
    frm = np.array(cam["from"], np.float64); at = np.array(cam["at"], np.float64); up = np.array(cam["up"], np.float64)
    cdir = (at - frm) / np.linalg.norm(at - frm)
    t = 2.0 * np.tan(np.radians(cam["fovy"]) * 0.5)
    hor = np.cross(cdir, up); hor = t * hor / np.linalg.norm(hor)
    ver = np.cross(hor, cdir)
    world = coords.astype(np.float64) * size - size / 2.0
    d = world - frm
    depth = np.linalg.norm(d, axis=1)
    p = d / (d @ cdir)[:, None]
    ix = np.floor(((p @ hor) / (hor @ hor) + 0.5) * W).astype(np.int64).clip(0, W - 1)
    iy = np.floor(((p @ ver) / (ver @ ver) + 0.5) * H).astype(np.int64).clip(0, H - 1)
    tx, ty, lane = ix >> 3, iy >> 3, ((iy & 7) << 3) | (ix & 7)
    if tile_order == "morton":
        def part(x):
            x = (x | (x << 8)) & 0x00FF00FF
            x = (x | (x << 4)) & 0x0F0F0F0F
            x = (x | (x << 2)) & 0x33333333
            x = (x | (x << 1)) & 0x55555555
            return x
        tkey = part(tx) | (part(ty) << 1)
    else:
        tkey = ty * (W >> 3) + tx
    raykey = tkey * 64 + lane
    uniq, inv = np.unique(raykey, return_inverse=True)     # rank of each sample's ray in tile order
    group = inv >> 6
    front = np.full(group.max() + 1, np.inf); np.minimum.at(front, group, depth)
    dbin = np.minimum(((depth - front[group]) / 8.0).astype(np.int64), 63)
    order = np.lexsort((depth, inv & 63, dbin, group))
    print(f"    ({len(uniq)} alive rays, {group.max() + 1} groups)")
    return coords[order]


def lines_per_sample(coords, chunk=64, n_max=1 << 21):
    """distinct 128-B lines per sample and level inside a 64-sample chunk: the table as it is (x-fastest rows / hash) against a
    de-hashed dense image in 4x4x2-entry bricks (F = 2: 32 entries = one line)"""
    c = coords[: (min(len(coords), n_max) // chunk) * chunk].astype(np.float32)
    scale0 = np.float32(pls)
    out = []
    for l in range(16):
        scale = np.float32(np.exp2(np.float32(l) * np.log2(scale0)) * np.float32(16.0) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        g = np.floor(c * scale + np.float32(0.5)).astype(np.int64)
        hashed = res ** 3 > (1 << 22)
        cur, brk = [], []
        for k in range(8):
            x, y, z = g[:, 0] + (k & 1), g[:, 1] + ((k >> 1) & 1), g[:, 2] + (k >> 2)
            if hashed:
                idx = (x ^ (y * 2654435761) ^ (z * 805459861)) & ((1 << 22) - 1)
            else:
                idx = x + y * res + z * res * res
            cur.append(idx >> 5)
            bx, by, bz = (res + 4) // 4, (res + 4) // 4, (res + 2) // 2
            brk.append((x >> 2) + bx * ((y >> 2) + by * (z >> 1)))
        def uniq(a):
            a = np.sort(np.stack(a, 1).reshape(-1, chunk * 8), axis=1)
            return (1 + (np.diff(a, axis=1) != 0).sum(1)).mean() / chunk
        out.append((l, res, hashed, uniq(cur), uniq(brk)))
    return out


for it in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4]:
    os.environ["VNR_AMD_DEBUG_MAX_ITERS"] = str(it)
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (1024, 1024))
    api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    dc, dn = C.c_void_p(), C.c_void_p()
    ms = (C.c_float * 32)()
    check(L.vnrAmdRendererDebugQueues(ren.h, C.byref(dc), C.byref(dn), ms, 32))
    cnt = np.zeros(16, np.uint32)
    check(L.vnrAmdMemcpyD2H(cnt.ctypes.data_as(C.c_void_p), dn, 64))
    n = int(cnt[2 + ((it - 1) & 1)])
    rec = np.empty((n, 4), np.float32)
    check(L.vnrAmdMemcpyD2H(rec.ctypes.data_as(C.c_void_p), dc, n * 16))
    coords = np.ascontiguousarray(rec[:, :3])
    print(f"iteration {it}: {n} samples (N_ITERS {os.environ.get('VNR_RM_N_ITERS', '24')})", flush=True)
    if "--lines" in sys.argv:
        tot_c = tot_b = 0.0
        for l, res, hashed, lc, lb in lines_per_sample(coords):
            print(f"    level {l:2d} res {res:5d} {'hash ' if hashed else 'dense'} lines/sample in a 64-chunk: as is {lc:.3f}   4x4x2 bricks {lb:.3f}")
            tot_c += lc; tot_b += lb
        print(f"    all levels: as is {tot_c:.2f} lines = {tot_c * 128:.0f} B per sample; bricks {tot_b:.2f} lines = {tot_b * 128:.0f} B per sample", flush=True)
        continue
    run("queue order", coords)
    # round 5: the CEILING of any order of the samples: every sample at its Morton position (1-voxel cells of the finest
    # level), over the whole queue and inside runs of one 64-ray group's batch (64 x 24 samples), which is all a per-group sort could reach
    q = np.floor(coords.astype(np.float64) * 1024).astype(np.int64).clip(0, 1023)
    run("ALL samples in Morton order (ceiling)", coords[np.argsort(morton(q), kind="stable")])
    m = (coords.shape[0] // 1536) * 1536
    key = morton(q[:m]).reshape(-1, 1536)
    order = (np.argsort(key, axis=1, kind="stable") + (np.arange(key.shape[0]) * 1536)[:, None]).reshape(-1)
    run("samples in Morton order inside runs of 1536 (one group's batch)", np.concatenate([coords[:m][order], coords[m:]]))
    run("regrouped: rays in row-major tile order", regroup_by_pixel(coords, "row"))
    run("regrouped: rays in Morton tile order", regroup_by_pixel(coords, "morton"))
    run("64-sample chunks sorted by Morton(centroid)", chunk_sorted(coords, 64))
    run("1024-sample runs sorted by Morton(centroid)", chunk_sorted(coords, 1024))
    q = np.floor(coords * 1024).astype(np.int64).clip(0, 1023)
    run("samples sorted by Morton code (upper bound)", coords[np.argsort(morton(q), kind="stable")])
    run("shuffled (lower bound)", coords[np.random.default_rng(0).permutation(n)])
    del ren


