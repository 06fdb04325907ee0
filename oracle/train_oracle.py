"""numpy restatement of one training step (forward via the C oracle, backward + Adam in numpy).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the training arithmetic lives in the un-vendored tiny-cuda-nn
submodule (Trainer::training_step, call site /root/reference/core/networks/tcnn_network.h:231); this file
restates the published upstream algorithm (L1/L2 loss with loss scale 128, fp16 activation gradients, ReLU
masking, hash-grid scatter-add, Adam with per-parameter step count) — every assumption is listed in
SURVEY.md Appendix A.  Used only for gradient checks on small cases.
Model shapes beyond the reference's example (n_neurons 16 / 32 / 128, Nearest, max_level, quantize_threshold): the MLP part is generic in the
width; Nearest sends the gradient to the one entry the forward pass read; a masked level gets none; quantize_threshold (a forward-pass zeroing of
small table values that exists only in the reference's fork, tcnn_impl_decoder.cu:120) is ASSUMED to pass the gradient through unchanged
(straight-through), the only reading under which a zeroed entry can ever move again.
"""
import numpy as np

from . import oracle as o

LOSS_SCALE = 128.0


def f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def split_mlp(params, in_w, W, nh):
    p = np.asarray(params).view(np.float16) if np.asarray(params).dtype == np.uint16 else np.asarray(params, np.float16)
    off = 0
    w1 = p[off:off + W * in_w].reshape(W, in_w); off += W * in_w
    wh = [p[off + i * W * W:off + (i + 1) * W * W].reshape(W, W) for i in range(nh)]; off += nh * W * W
    wl = p[off:off + 16 * W].reshape(16, W); off += 16 * W
    return w1, wh, wl, off


def corner_indices_and_weights(cfg, lay, coords):
    """vectorised restatement of tcnn pos_fract + grid_index; returns per level (idx [n,8], w [n,8])"""
    coords = np.asarray(coords, np.float32)
    out = []
    for l in range(cfg.n_levels):
        size = int(lay["offsets"][l + 1] - lay["offsets"][l])
        res = int(lay["resolution"][l])
        scale = np.float32(lay["scale"][l])
        pos = (coords.astype(np.float64) * np.float64(scale) + 0.5).astype(np.float32)  # single rounding = fma
        g = np.floor(pos)
        w = (pos - g).astype(np.float32)
        if cfg.interpolation == 1:
            w = (w * w * (np.float32(3.0) - np.float32(2.0) * w)).astype(np.float32)
        g = g.astype(np.int64)
        stride, hashed, d = 1, False, 0
        while d < 3 and stride <= size:
            stride *= res
            d += 1
        hashed = size < stride and getattr(cfg, "grid_type", 0) == 0   # EXTERNAL tcnn grid_index: only a Hash grid hashes
        n_dims = d                                                        # a Dense / Tiled level indexes over the dimensions the walk covered
        idxs = np.zeros((coords.shape[0], 8), np.int64)
        ws = np.zeros((coords.shape[0], 8), np.float32)
        if float(l) >= cfg.max_level + 1e-3:
            # a masked level encodes to zero and receives no gradient (EXTERNAL tcnn kernel_grid_backward tests max_level like the forward pass,
            # /root/reference/core/networks/tcnn_impl_decoder.cu:17): all weights zero
            out.append((idxs, ws))
            continue
        for c in range(8):
            pl = [(g[:, k] + ((c >> k) & 1)) & 0xFFFFFFFF for k in range(3)]
            wk = [np.where((c >> k) & 1, w[:, k], np.float32(1) - w[:, k]).astype(np.float32) for k in range(3)]
            ws[:, c] = ((wk[0] * wk[1]).astype(np.float32) * wk[2]).astype(np.float32)
            if hashed:
                idx = pl[0] ^ ((pl[1] * 2654435761) & 0xFFFFFFFF) ^ ((pl[2] * 805459861) & 0xFFFFFFFF)
            else:
                idx = pl[0]
                if n_dims >= 2:
                    idx = idx + pl[1] * res
                if n_dims >= 3:
                    idx = idx + pl[2] * res * res
                idx = idx & 0xFFFFFFFF
            idxs[:, c] = idx % size
        if cfg.interpolation == 2:
            # Nearest (tcnn_impl_decoder.cu:73-94 reads the lower corner's entry as it is): EXTERNAL tcnn kernel_grid_backward gives that one
            # entry the whole gradient
            ws[:, 0] = 1.0
            ws[:, 1:] = 0.0
        out.append((idxs, ws))
    return out


def act_backward(d, y, activation):
    """EXTERNAL tcnn warp_activation_backward: the gradient through the activation from its OUTPUT y (fp16 values as float32), every factor
    rounded to fp16 as tcnn's half arithmetic rounds it.  d, y: float32 arrays of fp16 values; returns float32 of fp16 values."""
    a = o.ACTIVATIONS.get(activation, activation)
    if a == 0:
        return d
    if a == 1:
        return np.where(y > 0, d, np.float32(0))
    if a == 2:
        return f16(d * y).astype(np.float32)
    if a == 3:
        inner = f16(np.float32(1) - y).astype(np.float32)
        return f16(d * f16(y * inner).astype(np.float32)).astype(np.float32)
    if a == 4:
        t = y * np.float32(10)
        return f16(d * f16(t * t / (t * t + np.float32(1))).astype(np.float32)).astype(np.float32)
    if a == 5:
        return f16(d * f16(np.float32(1) - np.exp(-y * np.float32(10))).astype(np.float32)).astype(np.float32)
    raise ValueError(activation)


def training_gradients(cfg, width, n_hidden_layers, params_bits, coords, targets, loss="L1", activation=1, output_activation=0):
    """returns dict(loss, grads [n_params] float64 (loss-scaled), y)"""
    lay = o.grid_layout(cfg)
    in_w = o.padded_width(cfg)
    nh = n_hidden_layers - 1
    F = cfg.n_features
    params = np.asarray(params_bits).view(np.float16)
    w1, wh, wl, n_mlp = split_mlp(params, in_w, width, nh)
    B = coords.shape[0]
    feat = o.grid_encode(cfg, params[n_mlp:].view(np.uint16), coords)
    y, acts = o.mlp_forward(params[:n_mlp].view(np.uint16), in_w, width, nh, feat, activation=o.act_code(activation, output_activation), want_activations=True)
    feat = feat.view(np.float16).astype(np.float32)
    acts = acts.view(np.float16).astype(np.float32)  # [nh+1, B, W]
    diff = y - np.asarray(targets, np.float32)
    if loss == "L1":
        loss_val = float(np.abs(diff).sum() / B)
        g = np.copysign(np.float32(1), diff)
    else:
        loss_val = float((diff * diff).sum() / B)
        g = 2 * diff
    dy = f16(np.float32(LOSS_SCALE) * g / np.float32(B)).astype(np.float32)
    # EXTERNAL tcnn FullyFusedMLP::backward: with an output activation the loss gradient first goes through it, from the output values
    dy = act_backward(dy, y.astype(np.float32), output_activation)
    grads = np.zeros(params.size, np.float64)
    # last layer (row 0 only; padded outputs have zero gradient)
    off_last = width * in_w + nh * width * width
    grads[off_last:off_last + width] = (dy[:, None].astype(np.float64) * acts[nh]).sum(0)
    d = f16(wl[0].astype(np.float32)[None, :] * dy[:, None]).astype(np.float32)
    d = act_backward(d, acts[nh], activation)
    for l in range(nh - 1, -1, -1):
        off = width * in_w + l * width * width
        grads[off:off + width * width] = (d.astype(np.float64).T @ acts[l].astype(np.float64)).ravel()
        d = f16(d.astype(np.float64) @ wh[l].astype(np.float64)).astype(np.float32)
        d = act_backward(d, acts[l], activation)
    grads[0:width * in_w] = (d.astype(np.float64).T @ feat.astype(np.float64)).ravel()
    dfeat = f16(d.astype(np.float64) @ w1.astype(np.float64)).astype(np.float32)  # [B, in_w]
    cw = corner_indices_and_weights(cfg, lay, coords)
    for l, (idxs, ws) in enumerate(cw):
        base = n_mlp + int(lay["offsets"][l]) * F
        for f in range(F):
            gf = dfeat[:, l * F + f].astype(np.float64)
            contrib = (ws.astype(np.float64) * gf[:, None]).ravel()
            np.add.at(grads, base + idxs.ravel() * F + f, contrib)
    return {"loss": loss_val, "grads": grads, "y": y, "n_mlp": n_mlp, "dfeat": dfeat}


def adam_step(master, grads_scaled, m, v, steps, n_matrix, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-15, l2_reg=1e-6,
              grad_scale=1.0):
    """EXTERNAL tcnn adam_step (per-parameter step count, zero-gradient grid entries skipped, l2 on matrices only)"""
    g = grads_scaled.astype(np.float64) * grad_scale / LOSS_SCALE
    idx = np.arange(master.size)
    active = (idx < n_matrix) | (g != 0)
    g = np.where(idx < n_matrix, g + l2_reg * master, g)
    m2 = np.where(active, beta1 * m + (1 - beta1) * g, m)
    v2 = np.where(active, beta2 * v + (1 - beta2) * g * g, v)
    s2 = np.where(active, steps + 1, steps)
    with np.errstate(divide="ignore", invalid="ignore"):
        lr_t = lr * np.sqrt(1 - beta2 ** s2) / (1 - beta1 ** s2)
        new = master - lr_t / (np.sqrt(v2) + eps) * m2
    new = np.where(active, new, master)
    return new, m2, v2, s2
