#!/usr/bin/env python3
"""Authors the golden vectors of SURVEY.md §8(c) (i)-(vi) and writes them next to this file.

    python tests/golden/make_golden.py          # regenerate every fixture

What the fixtures are, and what they are not.  The reference has no tests or golden vectors for this path
 and cannot be built or run in this image (CUDA + OptiX + tiny-cuda-nn), so nothing below comes
from the reference itself: PARITY VERSUS THE REFERENCE STAYS UNPINNED.  Two kinds of vectors are committed:

 * hand cases (`grid_hand_cases.npz`, `grid_hand_cases_r04.npz`, `bson_params_like.*`): computed HERE by an independent pure-Python
   restatement (dyadic inputs, exact arithmetic) or by an independent third-party encoder (pymongo's `bson`);
   the oracle and the HIP path are both checked against them;
 * frozen oracle outputs (`network_*.npz`, `network_r04_*.npz`, `c1_*.npz`, `dda_cases.npz`): produced by `oracle/` at the commit
   that introduced them.  They anchor the HIP path on the GPU box (where /root/reference does not exist) and
   catch silent drift of the oracle itself.

Fixtures hold data only (inputs, seeds, expected outputs).  Large seeded inputs (parameter blobs, the C1 volume)
are regenerated from their seed by `instantvnr_amd.synthetic` and pinned here by SHA-256.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from instantvnr_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------- (i) hand-computable hash-grid cases
# Independent restatement of tcnn_impl_decoder.cu:7-175 for DYADIC inputs (every product and sum below is exact in
# binary floating point, so plain Python floats are exact rationals here); fp16 rounding via numpy's IEEE conversion.
PRIMES = (1, 2654435761, 805459861)


def hand_grid_index(hashmap_size, res, g):
    stride, index = 1, 0
    for d in range(3):
        if stride <= hashmap_size:
            index += g[d] * stride
            stride *= res
    if hashmap_size < stride:
        index = 0
        for d in range(3):
            index ^= (g[d] * PRIMES[d]) & 0xFFFFFFFF
    return (index & 0xFFFFFFFF) % hashmap_size


def hand_encode(n_levels, F, log2_T, base, table, coords):
    """per_level_scale = 2, Linear interpolation.  table: fp16 [entries*F]; returns fp16 [n, pad16(L*F)]"""
    width = ((n_levels * F + 15) // 16) * 16
    out = np.zeros((len(coords), width), dtype=np.float16)
    offset = 0
    for lvl in range(n_levels):
        scale = float(2 ** lvl * base - 1)
        res = int(np.ceil(scale)) + 1
        size = min(((res ** 3 + 7) // 8) * 8, 1 << log2_T)
        for i, x in enumerate(coords):
            pos = [float(x[d]) * scale + 0.5 for d in range(3)]
            g = [int(np.floor(p)) for p in pos]
            w = [p - np.floor(p) for p in pos]
            acc = [np.float16(0)] * F
            for corner in range(8):
                weight, gl = 1.0, [0, 0, 0]
                for d in range(3):
                    if corner & (1 << d):
                        weight *= w[d]; gl[d] = g[d] + 1
                    else:
                        weight *= 1.0 - w[d]; gl[d] = g[d]
                e = hand_grid_index(size, res, gl)
                for f in range(F):
                    prod = np.float16(np.float32(weight) * np.float32(table[(offset + e) * F + f]))  # (T)(w * data)
                    acc[f] = np.float16(np.float32(acc[f]) + np.float32(prod))  # fp16 accumulate
            for f in range(F):
                out[i, lvl * F + f] = acc[f]
        offset += size
    return out, offset


def make_grid_hand_cases():
    # L=2, F=2, T=2^4, base 2: level 0 = 2^3 dense entries, level 1 = res 4 -> 64 > 16 -> hashed into 16 entries
    L, F, log2_T, base = 2, 2, 4, 2
    n_entries = 8 + 16
    table = np.zeros(n_entries * F, dtype=np.float16)
    e = np.arange(n_entries)
    table[0::2] = e                       # feature 0 = entry index (index ramp)
    table[1::2] = -0.25 * e               # feature 1
    k = np.array([[4, 4, 4], [2, 4, 4], [0, 0, 0], [8, 8, 8], [1, 3, 5], [7, 2, 6], [3, 3, 3], [5, 0, 8], [6, 7, 1]])
    coords = (k / 8.0).astype(np.float32)  # dyadic
    feats, total = hand_encode(L, F, log2_T, base, table, coords)
    assert total == n_entries
    np.savez(os.path.join(HERE, "grid_hand_cases.npz"), n_levels=L, n_features=F, log2_hashmap_size=log2_T,
             base_resolution=base, per_level_scale=2.0, table_f16_bits=table.view(np.uint16), coords=coords,
             features_f16_bits=feats.view(np.uint16))


# --------------------------------------------------------------------------- (ii) seeded networks
NETWORKS = {
    # name: (L, F, log2T, base, per_level_scale, n_hidden_layers, param seed)   -- SURVEY §8(c)(ii) shapes
    "L8_F8_H2": (8, 8, 15, 16, 2.0, 2, 101),
    "L16_F2_H3": (16, 2, 16, 16, 1.3195079565048218, 3, 102),
    "L16_F4_H3": (16, 4, 14, 8, 1.5, 3, 103),
}


def make_networks():
    for name, (L, F, T, base, pls, H, seed) in NETWORKS.items():
        cfg = oracle.grid_config(L, F, T, base, per_level_scale=pls)
        in_w = oracle.padded_width(cfg)
        n_mlp = oracle.mlp_n_params(in_w, 64, H - 1)
        n_params = oracle.n_params(cfg, 64, H)
        params = syn.random_params(n_params, n_mlp, seed=seed)
        coords = np.random.default_rng(seed + 1000).random((4096, 3), dtype=np.float32)
        bits = params.view(np.uint16)
        out32 = oracle.network_inference(cfg, 64, H, bits, coords, acc_mode=0)
        out16 = oracle.network_inference(cfg, 64, H, bits, coords, acc_mode=1)
        feats = oracle.grid_encode(cfg, bits[n_mlp:], coords[:512])
        np.savez_compressed(os.path.join(HERE, f"network_{name}.npz"), n_levels=L, n_features=F, log2_hashmap_size=T,
                            base_resolution=base, per_level_scale=pls, n_hidden_layers=H, n_neurons=64, param_seed=seed,
                            n_params=n_params, n_mlp_params=n_mlp, params_sha256=sha(params), coords=coords,
                            out_acc_f32=out32, out_acc_f16=out16, features_f16_bits_first512=feats)


# --------------------------------------------------------------------------- (iii) BSON
def make_bson():
    import bson
    blob = (np.arange(777, dtype=np.uint16) * 31 % 65521).astype(np.uint16).tobytes()
    mc = np.linspace(-1, 2, 16, dtype=np.float32).tobytes()
    doc = {  # params.json schema, network.cu:827-857 (keys sorted the way nlohmann's std::map stores them)
        "macrocell": {"data": bson.Binary(mc, 0), "dims": {"x": 2, "y": 2, "z": 2},
                      "spacings": {"x": 0.5, "y": 0.5, "z": 0.5}},
        "model": {"encoding": {"base_resolution": 16, "log2_hashmap_size": 19, "n_features_per_level": 8, "n_levels": 8,
                               "otype": "HashGrid"},
                  "loss": {"otype": "L1"},
                  "network": {"activation": "ReLU", "n_hidden_layers": 4, "n_neurons": 64, "otype": "FullyFusedMLP",
                              "output_activation": "None"}},
        "parameters": {"n_params": 777, "params_binary": bson.Binary(blob, 0), "params_type": "__half"},
        "volume": {"big": 5000000000, "dims": {"x": 32, "y": 32, "z": 32}, "flags": [True, False, None],
                   "name": "héllo", "neg": -7, "scale": 1.5},
    }
    enc = bson.encode(doc)
    open(os.path.join(HERE, "bson_params_like.bson"), "wb").write(enc)
    meta = {"params_binary_sha256": hashlib.sha256(blob).hexdigest(), "macrocell_data_sha256": hashlib.sha256(mc).hexdigest(),
            "n_bytes": len(enc),
            "plain": {"model": doc["model"], "volume": doc["volume"], "n_params": 777, "params_type": "__half",
                      "macrocell_dims": doc["macrocell"]["dims"]}}
    json.dump(meta, open(os.path.join(HERE, "bson_params_like.json"), "w"), indent=1, sort_keys=True)


# --------------------------------------------------------------------------- (iv) + (v) C1 scene
def c1_scene():
    """64^3 analytic field, 256^2, camera (0,0,-2.5*64) -> origin, up +y, fovy 60, rate 1"""
    vol = syn.analytic_volume(64)
    colors, alphas = syn.tfn_ramp_with_bumps()
    cam = syn.default_camera((64, 64, 64))
    return vol, colors, alphas, cam


def make_c1():
    vol, colors, alphas, cam = c1_scene()
    tfn = oracle.TfnHolder(colors, alphas)
    vr = oracle.macrocell_compute_implicit(vol)
    mo = oracle.macrocell_max_opacity(tfn, vr)
    sc = oracle.SceneHolder(256, 256, (64, 64, 64), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"])
    mono, _ = oracle.render_monolithic(sc, vol, n_threads=8)
    stream, _, st = oracle.render_streaming(sc, lambda c: oracle.sample_volume(vol, c, nodal=True))
    np.savez_compressed(os.path.join(HERE, "c1_scene.npz"), volume_sha256=sha(vol), tfn_colors=colors, tfn_alphas=alphas,
                        cam_from=np.array(cam["from"], np.float32), cam_at=np.array(cam["at"], np.float32),
                        cam_up=np.array(cam["up"], np.float32), fovy=cam["fovy"], macrocell_value_range=vr,
                        macrocell_max_opacity=mo, image_mode4_monolithic=mono, image_mode5_streaming=stream,
                        streaming_n_samples=st["n_samples"], streaming_n_rays_hit=st["n_rays_hit"],
                        streaming_n_iterations=st["n_iterations"])


# --------------------------------------------------------------------------- (vi) DDA cell sequences
DDA_RAYS = [
    # name, org, dir, t_min, t_max   (grid 4^3; dda.h:26-137)
    ("axis_aligned_x", (-1.0, 1.5, 2.5), (1.0, 0.0, 0.0), 1.0, 5.0),
    ("negative_x_zero_yz", (5.0, 0.5, 3.5), (-1.0, 0.0, 0.0), 1.0, 5.0),
    ("diagonal", (0.1, 0.2, 0.3), (1.0, 0.7, 0.4), 0.0, 3.5),
    ("exact_diagonal", (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 0.0, 4.0),
    ("grazing", (0.0, 3.9999, 0.5), (1.0, 1e-6, 0.0), 0.0, 4.0),
    ("zero_x", (2.5, -1.0, -1.0), (0.0, 1.0, 1.0), 1.0, 5.0),
    ("steep_negative", (3.9, 3.8, 3.7), (-0.2, -1.0, -0.55), 0.0, 3.7),
]


def make_dda():
    out = {}
    for name, org, d, t0, t1 in DDA_RAYS:
        cells, ts = oracle.dda_trace(org, d, t0, t1, (4, 4, 4))
        out[name + "_cells"] = cells
        out[name + "_ts"] = ts
        out[name + "_ray"] = np.array(list(org) + list(d) + [t0, t1], dtype=np.float32)
    # hand-checkable expectations (independent of the oracle): unit steps along an axis
    assert out["axis_aligned_x_cells"].tolist() == [[0, 1, 2], [1, 1, 2], [2, 1, 2], [3, 1, 2]]
    assert out["exact_diagonal_cells"].tolist() == [[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]]
    np.savez(os.path.join(HERE, "dda_cases.npz"), **out)


# --------------------------------------------------------------------------- (viii) the other rendering modes on the C1 scene
MODE_SHADING = {8: 1, 11: 2, 10: 2, 9: 4, 12: 3}   # rendering mode -> oracle shading_mode (ray marching)


def make_c1_modes():
    """frozen oracle outputs, 96 x 96: gradient shading (8, 9), single-shade heuristic (10, 11, 12), path tracing (13, 14, 15 at
    density scale 4, first frame)"""
    vol, colors, alphas, cam = c1_scene()
    tfn = oracle.TfnHolder(colors, alphas)
    mo = oracle.macrocell_max_opacity(tfn, oracle.macrocell_compute_implicit(vol))
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    out = {}
    for mode, sm in MODE_SHADING.items():
        sc = oracle.SceneHolder(96, 96, (64, 64, 64), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=sm)
        if mode == 10:
            out["mode10"] = oracle.render_monolithic(sc, vol, n_threads=8)[0]
        else:
            out[f"mode{mode}"] = oracle.render_streaming(sc, f, n_iters=512 if mode in (9, 12) else 16)[0]
    for mode, sm in ((14, 0), (15, 5)):
        sc = oracle.SceneHolder(96, 96, (64, 64, 64), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=sm, density_scale=4.0)
        out[f"mode{mode}"] = oracle.render_pathtracing(sc, f)[0]
    sc = oracle.SceneHolder(96, 96, (64, 64, 64), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], density_scale=4.0)
    out["mode13"] = oracle.render_pathtracing_monolithic(sc, vol)[0]
    assert np.array_equal(out["mode13"], out["mode15"])    # one estimator
    np.savez_compressed(os.path.join(HERE, "c1_modes.npz"), **{k: v.astype(np.float32) for k, v in out.items()})


# --------------------------------------------------------------------------- (ix) one batch of the out-of-core sampler
def ooc_volume():
    return np.random.default_rng(2024).integers(0, 65535, (6, 70, 300), dtype=np.uint16)   # [z, y, x]: 55-row slabs, 2 x 6 of them


def make_ooc():
    vol = ooc_volume()
    rng = np.random.default_rng(7)
    blocks = np.stack([rng.integers(0, 2, 20), rng.integers(0, 6, 20)], axis=1).astype(np.int32)
    n, offset = 1500, 12345
    r = oracle.pcg32_floats(5 * n, offset, 1337, 0xda3e39cb94b95bdb)
    c, v, bad = oracle.OocSlabSet(vol, blocks).sample((1000.0, 60000.0), r[:3 * n].reshape(n, 3), r[3 * n:4 * n], r[4 * n:],
                                                       lower=(0.1, 0.0, 0.2), upper=(0.9, 1.0, 0.7))
    assert bad == 0
    np.savez_compressed(os.path.join(HERE, "ooc_batch.npz"), volume_sha256=sha(vol), blocks=blocks, n=n, rng_offset=offset, coords=c, values=v,
                        random_head=r[:8])


# --------------------------------------------------------------------------- (x) round 4: grid types, Nearest, widths, activations
def hand_grid_index_typed(grid_type, size, res, g):
    """EXTERNAL tcnn grid_index for any grid type ("Hash", "Dense", "Tiled"): only a Hash grid replaces the partial stride walk by the hash"""
    stride, index = 1, 0
    for d in range(3):
        if stride <= size:
            index += g[d] * stride
            stride *= res
    if grid_type == "Hash" and size < stride:
        index = 0
        for d in range(3):
            index ^= (g[d] * PRIMES[d]) & 0xFFFFFFFF
    return (index & 0xFFFFFFFF) % size


def hand_level_size(grid_type, res, base, log2_T):
    n = ((res ** 3 + 7) // 8) * 8
    if grid_type == "Tiled":
        n = min(n, base ** 3)
    elif grid_type == "Hash":
        n = min(n, 1 << log2_T)
    return n


def hand_encode_typed(grid_type, nearest, n_levels, F, log2_T, base, table, coords):
    """the independent restatement above for any grid type and for Nearest (the lower corner's entry as it is, tcnn_impl_decoder.cu:73-94)"""
    width = ((n_levels * F + 15) // 16) * 16
    out = np.zeros((len(coords), width), dtype=np.float16)
    offset = 0
    for lvl in range(n_levels):
        scale = float(2 ** lvl * base - 1)
        res = int(np.ceil(scale)) + 1
        size = hand_level_size(grid_type, res, base, log2_T)
        for i, x in enumerate(coords):
            pos = [float(x[d]) * scale + 0.5 for d in range(3)]
            g = [int(np.floor(p)) for p in pos]
            w = [p - np.floor(p) for p in pos]
            if nearest:
                e = hand_grid_index_typed(grid_type, size, res, g)
                for f in range(F):
                    out[i, lvl * F + f] = table[(offset + e) * F + f]
                continue
            acc = [np.float16(0)] * F
            for corner in range(8):
                weight, gl = 1.0, [0, 0, 0]
                for d in range(3):
                    if corner & (1 << d):
                        weight *= w[d]; gl[d] = g[d] + 1
                    else:
                        weight *= 1.0 - w[d]; gl[d] = g[d]
                e = hand_grid_index_typed(grid_type, size, res, gl)
                for f in range(F):
                    prod = np.float16(np.float32(weight) * np.float32(table[(offset + e) * F + f]))
                    acc[f] = np.float16(np.float32(acc[f]) + np.float32(prod))
            for f in range(F):
                out[i, lvl * F + f] = acc[f]
        offset += size
    return out, offset


def make_grid_hand_cases_r04():
    """hand cases (independent pure-Python restatement, dyadic inputs) for what round 4 added to the encoding: Dense and Tiled grids
    (L = 3, base 2: resolutions 2, 4, 8; Tiled caps every level at 2^3 = 8 entries, so levels 1 and 2 wrap over 2 / 1 index dimensions;
    Dense keeps 8 + 64 + 512 entries) and Nearest on a Hash grid.  Index-ramp tables: a value says which entry was read."""
    L, F, log2_T, base = 3, 2, 5, 2
    k = np.array([[4, 4, 4], [2, 4, 4], [0, 0, 0], [8, 8, 8], [1, 3, 5], [7, 2, 6], [3, 3, 3], [5, 0, 8], [6, 7, 1]])
    coords = (k / 8.0).astype(np.float32)
    out = {"n_levels": L, "n_features": F, "log2_hashmap_size": log2_T, "base_resolution": base, "coords": coords}
    for kind, gtype, nearest in (("dense", "Dense", False), ("tiled", "Tiled", False), ("hash_nearest", "Hash", True), ("tiled_nearest", "Tiled", True)):
        n_entries = sum(hand_level_size(gtype, 2 ** l * base, base, log2_T) for l in range(L))
        table = np.zeros(n_entries * F, dtype=np.float16)
        e = np.arange(n_entries)
        table[0::2] = (e % 512).astype(np.float16)          # feature 0 = entry index (exact in fp16 below 2048)
        table[1::2] = (-0.125 * (e % 512)).astype(np.float16)
        feats, total = hand_encode_typed(gtype, nearest, L, F, log2_T, base, table, coords)
        assert total == n_entries
        out[f"{kind}_table_f16_bits"] = table.view(np.uint16)
        out[f"{kind}_features_f16_bits"] = feats.view(np.uint16)
    np.savez(os.path.join(HERE, "grid_hand_cases_r04.npz"), **out)


NETWORKS_R04 = {
    # name: (L, F, log2T, base, pls, n_hidden_layers, n_neurons, interpolation, activation, output_activation, grid type, param seed, mlp_scale)
    "W128_H2": (8, 2, 14, 8, 1.5, 2, 128, "Linear", "ReLU", "None", "Hash", 201, 1.0),
    "W16_sigmoid_dense": (5, 4, 12, 4, 2.0, 3, 16, "Smoothstep", "Sigmoid", "None", "Dense", 202, 6.0),   # (larger weights: a sigmoid net with small ones is nearly constant)
    "W32_tiled_nearest_squareplus_expout": (8, 2, 12, 4, 2.0, 2, 32, "Nearest", "Squareplus", "Exponential", "Tiled", 203, 2.0),
}
INTERP = {"Linear": 0, "Smoothstep": 1, "Nearest": 2}


def make_networks_r04():
    """frozen oracle outputs for the kinds of model round 4 put on the MFMA kernels"""
    for name, (L, F, T, base, pls, H, W, interp, act, out_act, gtype, seed, mlp_scale) in NETWORKS_R04.items():
        cfg = oracle.grid_config(L, F, T, base, per_level_scale=pls, interpolation=INTERP[interp], grid_type=gtype)
        in_w = oracle.padded_width(cfg)
        n_mlp = oracle.mlp_n_params(in_w, W, H - 1)
        n_params = oracle.n_params(cfg, W, H)
        params = syn.random_params(n_params, n_mlp, seed=seed, mlp_scale=mlp_scale)
        coords = np.random.default_rng(seed + 1000).random((4096, 3), dtype=np.float32)
        bits = params.view(np.uint16)
        code = oracle.act_code(act, out_act)
        out32 = oracle.network_inference(cfg, W, H, bits, coords, activation=code, acc_mode=0)
        out16 = oracle.network_inference(cfg, W, H, bits, coords, activation=code, acc_mode=1)
        feats = oracle.grid_encode(cfg, bits[n_mlp:], coords[:512])
        np.savez_compressed(os.path.join(HERE, f"network_r04_{name}.npz"), n_levels=L, n_features=F, log2_hashmap_size=T,
                            base_resolution=base, per_level_scale=pls, n_hidden_layers=H, n_neurons=W, interpolation=interp, activation=act,
                            output_activation=out_act, grid_type=gtype, param_seed=seed, mlp_scale=mlp_scale,
                            n_params=n_params, n_mlp_params=n_mlp, params_sha256=sha(params), coords=coords,
                            out_acc_f32=out32, out_acc_f16=out16, features_f16_bits_first512=feats)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "r04":   # only what round 4 added (the older fixtures stay byte-identical)
        make_grid_hand_cases_r04()
        make_networks_r04()
        sys.exit(0)
    make_grid_hand_cases_r04()
    make_networks_r04()
    make_c1_modes()
    make_ooc()
    make_grid_hand_cases()
    make_networks()
    make_bson()
    make_c1()
    make_dda()
    for f in sorted(os.listdir(HERE)):
        print(f"{os.path.getsize(os.path.join(HERE, f)):>9}  {f}")
