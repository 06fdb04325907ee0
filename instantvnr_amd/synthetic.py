"""Seeded synthetic inputs for the BASELINE configs (pure numpy, no GPU, no oracle).

The reference's datasets (vorts1, ...) live in OVR's data folder and are not
available (SURVEY.md §8c "Dataset caveat"), so every config runs on a seeded
synthetic field of the same shape.  1024^3-class volumes are generated on the
GPU by the library itself (vnrAmdVolumeCreatePerlin); this module covers the
small CPU-side cases and the transfer functions / model configs.
"""
import json

import numpy as np


def analytic_volume(n=64):
    """C1: f = 0.5 + 0.5 sin(6 pi x) cos(4 pi y) sin(2 pi z), blended with a Gaussian blob.
    Returns [z, y, x] float32 in [0, 1], x fastest (REGULAR_GRID_RAW_BINARY order)."""
    c = (np.arange(n, dtype=np.float64) + 0.5) / n
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    f = 0.5 + 0.5 * np.sin(6 * np.pi * x) * np.cos(4 * np.pi * y) * np.sin(2 * np.pi * z)
    blob = np.exp(-((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) / (2 * 0.18 ** 2))
    v = 0.55 * f * blob + 0.45 * blob
    v = (v - v.min()) / (v.max() - v.min())
    return v.astype(np.float32)


def vortex_volume(n=128, seed=1234, n_tubes=12):
    """C2/C3 stand-in for vorts1: sum of seeded Gaussian vortex tubes, normalised to [0,1]."""
    rng = np.random.default_rng(seed)
    c = (np.arange(n, dtype=np.float32) + 0.5) / n
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    p = np.stack([x, y, z], axis=-1)
    v = np.zeros((n, n, n), dtype=np.float32)
    for _ in range(n_tubes):
        a = rng.uniform(0.1, 0.9, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        curl = rng.uniform(2.0, 6.0)
        amp = rng.uniform(0.5, 1.0)
        sig = rng.uniform(0.03, 0.07)
        rel = p - a
        t = rel @ d
        # helical centre line around the axis
        perp1 = np.cross(d, np.array([0.3, 0.5, 0.81], np.float32))
        perp1 /= np.linalg.norm(perp1)
        perp2 = np.cross(d, perp1)
        off = 0.05 * (np.cos(curl * 2 * np.pi * t)[..., None] * perp1 + np.sin(curl * 2 * np.pi * t)[..., None] * perp2)
        r = rel - t[..., None] * d - off
        v += (amp * np.exp(-(r * r).sum(-1) / (2 * sig * sig))).astype(np.float32)
    v = (v - v.min()) / (v.max() - v.min())
    return v.astype(np.float32)


def _colormap(n):
    """cool-to-warm style map, deterministic"""
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)
    r = np.clip(1.5 * t + 0.1, 0, 1)
    g = np.clip(1.0 - np.abs(2.0 * t - 1.0) * 0.9, 0, 1)
    b = np.clip(1.4 - 1.5 * t, 0, 1)
    return np.stack([r, g, b], axis=1).astype(np.float32)


def tfn_ramp_with_bumps(n=256, seed=7, zero_below=0.15, opacity_scale=1.0):
    """256-entry ramp-with-bumps transfer function.  Values below
    `zero_below` are fully transparent so empty-space skipping has work to do; `opacity_scale` thins the
    medium (large volumes need a small per-voxel-step opacity for rays to penetrate).
    Returns (colors [n,3], alphas [n])."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)
    a = np.clip((t - zero_below) / (1.0 - zero_below), 0, 1) * 0.35
    for _ in range(3):
        mu = rng.uniform(0.3, 0.9)
        s = rng.uniform(0.02, 0.05)
        a = a + 0.55 * np.exp(-((t - mu) ** 2) / (2 * s * s))
    a = np.clip(a, 0.0, 1.0)
    a[t < zero_below] = 0.0
    a = np.clip(a * opacity_scale, 0.0, 1.0)
    return _colormap(n), a.astype(np.float32)


def model_config(n_levels=8, n_features=8, log2_hashmap_size=19, base_resolution=16, n_neurons=64,
                 n_hidden_layers=2, per_level_scale=None):
    """Model JSON in the reference's format (example-model.json:2-32)."""
    enc = {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features,
           "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution}
    if per_level_scale is not None:
        enc["per_level_scale"] = float(per_level_scale)
    return {
        "optimizer": {"otype": "ExponentialDecay", "decay_start": 2000, "decay_interval": 1000, "decay_base": 0.99,
                      "nested": {"otype": "Adam", "learning_rate": 5e-3, "beta1": 0.9, "beta2": 0.999,
                                 "epsilon": 1e-15, "l2_reg": 1e-6}},
        "loss": {"otype": "L1"},
        "encoding": enc,
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                    "n_neurons": n_neurons, "n_hidden_layers": n_hidden_layers},
    }


def model_config_text(**kw):
    return json.dumps(model_config(**kw))


def random_params(n_params, mlp_params, seed=0, mlp_scale=1.0):
    """Seeded fp16 parameter blob: MLP weights ~ U(-0.35, 0.35) x mlp_scale, grid ~ U(-1, 1).
    (Larger than tcnn's init on purpose: parity tests want non-trivial activations.)"""
    rng = np.random.default_rng(seed)
    p = np.empty(n_params, dtype=np.float16)
    p[:mlp_params] = (rng.uniform(-0.35, 0.35, mlp_params) * mlp_scale).astype(np.float16)
    p[mlp_params:] = rng.uniform(-1.0, 1.0, n_params - mlp_params).astype(np.float16)
    return p


def default_camera(dims, distance_scale=2.5):
    """camera looking down +z at the volume centre (batch_renderer.cpp:79-81 style defaults)"""
    d = float(max(dims))
    return {"from": (0.0, 0.0, -distance_scale * d), "at": (0.0, 0.0, 0.0), "up": (0.0, 1.0, 0.0), "fovy": 60.0}


def oblique_camera(dims, distance_scale=1.6):
    d = float(max(dims))
    return {"from": (0.9 * distance_scale * d, 0.55 * distance_scale * d, -1.1 * distance_scale * d),
            "at": (0.0, 0.0, 0.0), "up": (0.0, 1.0, 0.0), "fovy": 45.0}
