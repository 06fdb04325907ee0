"""CPU checks of the numpy training restatement: analytic gradient vs finite differences of a smooth
surrogate, and the vectorised corner/weight code vs the C oracle's encode."""
import numpy as np

from oracle import train_oracle as T


def test_corner_weights_reproduce_c_encode(oracle):
    cfg = oracle.grid_config(5, 2, 10, 4, 1.7)
    lay = oracle.grid_layout(cfg)
    rng = np.random.default_rng(0)
    table = rng.uniform(-1, 1, lay["total_entries"] * 2).astype(np.float16)
    coords = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    enc = oracle.grid_encode(cfg, table.view(np.uint16), coords).view(np.float16).astype(np.float32)
    cw = T.corner_indices_and_weights(cfg, lay, coords)
    for l, (idx, w) in enumerate(cw):
        assert np.allclose(w.sum(1), 1.0, atol=1e-6)
        base = int(lay["offsets"][l]) * 2
        for f in range(2):
            vals = table[base + idx * 2 + f].astype(np.float32)
            assert np.allclose((vals * w).sum(1), enc[:, l * 2 + f], atol=4e-3)   # fp16 accumulate in the C path


def test_l1_gradients_match_finite_differences_on_last_layer(oracle):
    cfg = oracle.grid_config(3, 2, 8, 4)
    W, H = 64, 2
    rng = np.random.default_rng(1)
    n = oracle.n_params(cfg, W, H)
    n_mlp = oracle.mlp_n_params(16, W, H - 1)
    params = np.concatenate([rng.uniform(-0.4, 0.4, n_mlp), rng.uniform(-1, 1, n - n_mlp)]).astype(np.float16)
    coords = rng.uniform(0, 1, (64, 3)).astype(np.float32)
    targets = rng.uniform(0, 1, 64).astype(np.float32)
    out = T.training_gradients(cfg, W, H, params.view(np.uint16), coords, targets, loss="L2")
    # L2 loss is smooth in the last-layer weights: dL/dw_k = sum_b 2 (y-t)/B a_k
    off = W * 16 + (H - 1) * W * W
    _, acts = oracle.mlp_forward(params[:n_mlp].view(np.uint16), 16, W, H - 1,
                                 oracle.grid_encode(cfg, params[n_mlp:].view(np.uint16), coords), want_activations=True)
    a = acts.view(np.float16).astype(np.float64)[H - 1]
    want = (2 * (out["y"] - targets)[:, None] / 64 * a).sum(0) * T.LOSS_SCALE
    got = out["grads"][off:off + W]
    assert np.allclose(got, want, rtol=2e-2, atol=2e-3 * np.abs(want).max())


def test_adam_first_step_is_lr_sized():
    master = np.array([0.5, -0.25, 0.1, 0.2], np.float64)
    grads = np.array([128.0, -64.0, 0.0, 12.8], np.float64)  # loss-scaled
    new, m, v, s = T.adam_step(master, grads, np.zeros(4), np.zeros(4), np.zeros(4), n_matrix=2, l2_reg=0.0)
    assert np.allclose(new[:2], master[:2] - 5e-3 * np.sign(grads[:2]), atol=1e-9)  # bias-corrected first step = lr * sign
    assert new[2] == master[2] and s[2] == 0                                        # untouched grid entry skipped
    assert np.isclose(new[3], master[3] - 5e-3, atol=1e-9)
