"""Pins the CPU oracle against hand-computable cases and independent restatements.

The reference has no tests/golden vectors, so these are the
oracle's own known-answer tests: IEEE fp16 vs numpy, the published pcg32 test
vector, an independent pure-Python TEA/LCG, hash-grid sizing of the reference's
example model, and a pure-numpy hash-grid/MLP restatement on tiny cases.
"""
import numpy as np
import pytest


def test_fp16_conversion_matches_ieee(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        rng.normal(size=2000).astype(np.float32),
        (rng.normal(size=2000) * 1e-5).astype(np.float32),
        (rng.normal(size=500) * 3e4).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.9604645e-8, 2.9802322e-8,
                  2.9802326e-8, 6.1035156e-5, 6.0975552e-5, 1.0009766, 1.00048828125, 1.00146484375],
                 dtype=np.float32),
    ])
    got = oracle.f32_to_f16_bits(xs)
    want = xs.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)
    L = oracle.lib()
    back = np.array([L.vnro_f16_to_f32(int(h)) for h in range(0, 0x7c00, 37)], dtype=np.float32)
    assert np.array_equal(back, np.arange(0, 0x7c00, 37, dtype=np.uint16).view(np.float16).astype(np.float32))


def test_pcg32_published_vector(oracle):
    # pcg32 reference demo output for pcg32_srandom(42u, 54u) (pcg-random.org, pcg32-demo)
    import ctypes as C
    r = oracle.pcg32(42, 54)
    L = oracle.lib()
    got = [L.vnro_pcg32_next_uint(C.byref(r)) for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    # advance(n) == n draws
    a = oracle.pcg32(1337, 0xda3e39cb94b95bdb)
    b = oracle.pcg32(1337, 0xda3e39cb94b95bdb)
    for _ in range(1000):
        L.vnro_pcg32_next_uint(C.byref(a))
    L.vnro_pcg32_advance(C.byref(b), C.c_int64(1000))
    assert a.state == b.state
    f = L.vnro_pcg32_next_float(C.byref(a))
    assert 0.0 <= f < 1.0


def _tea_lcg_py(v0, v1, n):
    M = 0xFFFFFFFF
    s0 = 0
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c & M) ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d & M) ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    st = v0
    out = []
    for _ in range(n):
        st = (1664525 * st + 1013904223) & M
        out.append(np.float32(st & 0x00FFFFFF) / np.float32(0x01000000))
    return np.array(out, dtype=np.float32)


@pytest.mark.parametrize("v0,v1", [(1, 0), (1, 12345), (7, 1048575), (0, 0)])
def test_tea_lcg_matches_independent_python(oracle, v0, v1):
    got = oracle.lcg_floats(v0, v1, 4)
    assert np.array_equal(got, _tea_lcg_py(v0, v1, 4))
    assert np.all((got >= 0) & (got < 1))


def test_grid_layout_of_example_model(oracle):
    # example-model.json:19-25 -> L=8, F=8, T=2^19, base 16; SURVEY §8(a6): 2 920 448 entries
    cfg = oracle.grid_config(8, 8, 19, 16)
    lay = oracle.grid_layout(cfg)
    assert lay["total_entries"] == 2920448
    assert list(lay["resolution"]) == [16, 32, 64, 128, 256, 512, 1024, 2048]
    assert list(np.diff(lay["offsets"].astype(np.int64))) == [4096, 32768, 262144] + [524288] * 5
    assert np.allclose(lay["scale"], [15, 31, 63, 127, 255, 511, 1023, 2047])
    # MLP (in 64, 4 hidden layers => 3 hidden matmuls) = 17408 fp16  
    assert oracle.mlp_n_params(64, 64, 3) == 17408
    assert oracle.n_params(cfg, 64, 4) == 17408 + 2920448 * 8 == 23380992


def test_grid_index_dense_and_hash(oracle):
    # dense: stride never exceeds hashmap size
    assert oracle.grid_index(4096, 16, (3, 2, 1)) == 3 + 2 * 16 + 1 * 256
    # hashed: prime-XOR hash (tcnn fast_hash) mod T
    T = 1 << 19
    p = (1000, 2000, 3000)
    want = ((p[0] * 1) ^ ((p[1] * 2654435761) & 0xFFFFFFFF) ^ ((p[2] * 805459861) & 0xFFFFFFFF)) % T
    assert oracle.grid_index(T, 1024, p) == want


def _encode_numpy(cfg, lay, table, coords):
    """independent restatement with numpy scalars (fp32 positions, fp16 accumulate)"""
    L, F = cfg.n_levels, cfg.n_features
    out = np.zeros((coords.shape[0], ((L * F + 15) // 16) * 16), dtype=np.float16)
    for i, x in enumerate(coords.astype(np.float32)):
        for l in range(L):
            off, size = int(lay["offsets"][l]), int(lay["offsets"][l + 1] - lay["offsets"][l])
            scale, res = np.float32(lay["scale"][l]), int(lay["resolution"][l])
            # fma(x, scale, 0.5) evaluated exactly in float64 then rounded once
            pos = (x.astype(np.float64) * np.float64(scale) + 0.5).astype(np.float32)
            g = np.floor(pos)
            w = (pos - g).astype(np.float32)
            g = g.astype(np.int64)
            acc = np.zeros(F, dtype=np.float16)
            for idx in range(8):
                wt = np.float32(1)
                pl = []
                for d in range(3):
                    if idx & (1 << d):
                        wt = np.float32(wt * w[d]); pl.append(int(g[d]) + 1)
                    else:
                        wt = np.float32(wt * (np.float32(1) - w[d])); pl.append(int(g[d]))
                stride, index, d = 1, 0, 0
                while d < 3 and stride <= size:
                    index += pl[d] * stride; stride *= res; d += 1
                index &= 0xFFFFFFFF
                if size < stride:
                    index = (pl[0] ^ ((pl[1] * 2654435761) & 0xFFFFFFFF) ^ ((pl[2] * 805459861) & 0xFFFFFFFF))
                index %= size
                for f in range(F):
                    data = np.float32(table[(off + index) * F + f])
                    acc[f] = np.float16(np.float32(acc[f]) + np.float32(np.float16(np.float32(wt * data))))
            out[i, l * F:(l + 1) * F] = acc
    return out


@pytest.mark.parametrize("L,F,log2T,base", [(2, 2, 4, 2), (4, 2, 10, 4), (3, 8, 8, 4), (5, 4, 12, 8), (2, 1, 6, 3)])
def test_grid_encode_matches_independent_numpy(oracle, L, F, log2T, base):
    cfg = oracle.grid_config(L, F, log2T, base)
    lay = oracle.grid_layout(cfg)
    rng = np.random.default_rng(L * 100 + F)
    table = rng.uniform(-1, 1, lay["total_entries"] * F).astype(np.float16)
    coords = rng.uniform(0, 1, (40, 3)).astype(np.float32)
    coords[0] = (0, 0, 0); coords[1] = (1, 1, 1); coords[2] = (0.5, 0.25, 0.75)
    got = oracle.grid_encode(cfg, table.view(np.uint16), coords).view(np.float16)
    want = _encode_numpy(cfg, lay, table, coords)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_grid_encode_hand_case_index_ramp(oracle):
    """L=1, base 2 -> scale 1, res 2, 8 dense entries; table[e] = e (feature 0), -e (feature 1).
    At x = (0.5, 0.5, 0.5): pos = 1.0 -> cell (1,1,1), frac 0 -> only corner 0 has weight 1:
    index = (1 + 2 + 4) % 8 = 7."""
    cfg = oracle.grid_config(1, 2, 4, 2)
    lay = oracle.grid_layout(cfg)
    assert lay["total_entries"] == 8 and lay["resolution"][0] == 2
    table = np.zeros(16, dtype=np.float16)
    table[0::2] = np.arange(8); table[1::2] = -np.arange(8)
    out = oracle.grid_encode(cfg, table.view(np.uint16), np.array([[0.5, 0.5, 0.5]], np.float32)).view(np.float16)
    assert out.shape == (1, 16)
    assert out[0, 0] == 7 and out[0, 1] == -7 and np.all(out[0, 2:] == 0)
    # x = (0.25, 0.5, 0.5): pos.x = 0.75 -> cell 0, frac .75: 0.25*table[6] + 0.75*table[7] = 6.75
    out = oracle.grid_encode(cfg, table.view(np.uint16), np.array([[0.25, 0.5, 0.5]], np.float32)).view(np.float16)
    assert out[0, 0] == np.float16(6.75) and out[0, 1] == np.float16(-6.75)


def _mlp_numpy(w, in_w, W, nh, x):
    """sequential fp32 accumulation, fp16 activations (matches VNRO_ACC_F32)"""
    o = 0
    W1 = w[o:o + W * in_w].reshape(W, in_w).astype(np.float32); o += W * in_w
    Wh = [w[o + i * W * W:o + (i + 1) * W * W].reshape(W, W).astype(np.float32) for i in range(nh)]; o += nh * W * W
    Wl = w[o:o + 16 * W].reshape(16, W).astype(np.float32)

    def dense(M, v):
        out = np.zeros(M.shape[0], dtype=np.float32)
        for r in range(M.shape[0]):
            s = np.float32(0)
            for k in range(M.shape[1]):
                s = np.float32(s + np.float32(M[r, k] * v[k]))
            out[r] = s
        return out

    ys = []
    for row in x.astype(np.float32):
        a = np.maximum(dense(W1, row).astype(np.float16), np.float16(0)).astype(np.float32)
        for M in Wh:
            a = np.maximum(dense(M, a).astype(np.float16), np.float16(0)).astype(np.float32)
        ys.append(np.float32(np.float16(dense(Wl[:1], a)[0])))
    return np.array(ys, dtype=np.float32)


@pytest.mark.parametrize("in_w,W,nh", [(16, 16, 0), (32, 64, 2), (64, 64, 1), (16, 32, 3)])
def test_mlp_forward_matches_independent_numpy(oracle, in_w, W, nh):
    rng = np.random.default_rng(in_w + W + nh)
    n = oracle.mlp_n_params(in_w, W, nh)
    w = rng.uniform(-0.4, 0.4, n).astype(np.float16)
    x = rng.uniform(-1, 1, (12, in_w)).astype(np.float16)
    got = oracle.mlp_forward(w.view(np.uint16), in_w, W, nh, x.view(np.uint16))
    want = _mlp_numpy(w, in_w, W, nh, x)
    assert np.array_equal(got, want)
    # fp16-accumulate emulation (the reference's half accumulator) stays within the stated tolerance
    got16 = oracle.mlp_forward(w.view(np.uint16), in_w, W, nh, x.view(np.uint16), acc_mode=1)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.max(np.abs(got16 - want)) <= 2.0 ** -6 * scale


def test_network_inference_is_encode_then_mlp(oracle):
    cfg = oracle.grid_config(4, 4, 10, 4)
    W, H = 32, 3
    lay = oracle.grid_layout(cfg)
    n_mlp = oracle.mlp_n_params(16, W, H - 1)
    rng = np.random.default_rng(5)
    params = np.concatenate([rng.uniform(-0.4, 0.4, n_mlp), rng.uniform(-1, 1, lay["total_entries"] * 4)]).astype(np.float16)
    coords = rng.uniform(0, 1, (64, 3)).astype(np.float32)
    out = oracle.network_inference(cfg, W, H, params.view(np.uint16), coords)
    enc = oracle.grid_encode(cfg, params[n_mlp:].view(np.uint16), coords)
    out2 = oracle.mlp_forward(params[:n_mlp].view(np.uint16), 16, W, H - 1, enc)
    assert np.array_equal(out, out2)


def test_mssim_known_answers_and_explicit_loop(oracle):
    """get_mssim (network.cu:474-549): identical volumes -> 1; two constant volumes a, b -> (2ab + C1) / (a^2 + b^2 + C1)
    (all variances vanish, the C2 terms cancel); and the cumulative-sum box filter equals an explicit 7^3 loop."""
    rng = np.random.default_rng(4)
    a = rng.random((9, 10, 11))
    assert oracle.mssim(a, a) == pytest.approx(1.0, abs=1e-12)
    c1 = 0.01 ** 2
    assert oracle.mssim(np.full((8, 8, 8), 0.3), np.full((8, 8, 8), 0.7)) == pytest.approx((2 * 0.3 * 0.7 + c1) / (0.09 + 0.49 + c1), abs=1e-12)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1)
    n_p, c2 = 343, 0.03 ** 2
    acc = []
    for z in range(a.shape[0] - 6):
        for y in range(a.shape[1] - 6):
            for x in range(a.shape[2] - 6):
                wx, wy = a[z:z + 7, y:y + 7, x:x + 7].ravel(), b[z:z + 7, y:y + 7, x:x + 7].ravel()
                ux, uy = wx.mean(), wy.mean()
                vx, vy = wx.var(ddof=1), wy.var(ddof=1)                  # sample covariance = NP / (NP - 1) x population
                vxy = ((wx - ux) * (wy - uy)).sum() / (n_p - 1)
                acc.append((2 * ux * uy + c1) * (2 * vxy + c2) / ((ux * ux + uy * uy + c1) * (vx + vy + c2)))
    assert oracle.mssim(b, a) == pytest.approx(np.mean(acc), abs=1e-10)     # mssim(pred, ref); SSIM is symmetric
    assert oracle.mssim(b, a) < 0.99


def test_nearest_quantize_and_max_level_hand_cases(oracle):
    """the three encoding options of tcnn_impl_decoder.cu the oracle restates beyond Linear / Smoothstep, on the index-ramp table of the
    hand case above (L = 2 here: level 0 has 8 entries, table[e] = e / -e; level 1: base 2 x scale 2 - 1 = 3 -> res 4, 64 entries).
    Nearest (:73-94): the entry of the LOWER corner, no blend: x = (0.25, 0.5, 0.5) -> cell (0, 1, 1) -> index 6.
    quantize_threshold (:120): a corner value below it in magnitude counts as 0 in the blend (and Nearest ignores it).
    max_level (:17): levels l >= max_level + 1e-3 encode to zero."""
    x = np.array([[0.25, 0.5, 0.5]], np.float32)

    def table_for(cfg):
        lay = oracle.grid_layout(cfg)
        t = np.zeros(lay["total_entries"] * 2, np.float16)
        t[0:16:2] = np.arange(8); t[1:16:2] = -np.arange(8)
        t[16::2] = 0.5; t[17::2] = -0.25
        return t

    near = oracle.grid_config(2, 2, 8, 2, interpolation=2)
    out = oracle.grid_encode(near, table_for(near).view(np.uint16), x).view(np.float16)
    assert out[0, 0] == 6 and out[0, 1] == -6 and out[0, 2] == np.float16(0.5) and out[0, 3] == np.float16(-0.25)
    # Linear at the same point blends entries 6 and 7 (0.25 / 0.75); a threshold of 6.5 removes entry 6 from the blend
    lin = oracle.grid_config(2, 2, 8, 2)
    q = oracle.grid_config(2, 2, 8, 2, quantize_threshold=6.5)
    assert oracle.grid_encode(lin, table_for(lin).view(np.uint16), x).view(np.float16)[0, 0] == np.float16(6.75)
    out = oracle.grid_encode(q, table_for(q).view(np.uint16), x).view(np.float16)
    assert out[0, 0] == np.float16(0.75 * 7) and out[0, 1] == np.float16(-0.75 * 7)
    assert out[0, 2] == 0 and out[0, 3] == 0          # level 1: every value (0.5, -0.25) is below the threshold
    nq = oracle.grid_config(2, 2, 8, 2, interpolation=2, quantize_threshold=6.5)
    assert oracle.grid_encode(nq, table_for(nq).view(np.uint16), x).view(np.float16)[0, 0] == 6   # Nearest does not quantise
    # max_level: 1.0 masks level 1 only (1 >= 1.001 is false for level 0 ... and true for level 1? 1 >= 1.001 is FALSE: level 1 stays)
    m1 = oracle.grid_config(2, 2, 8, 2, max_level=1.0)
    assert oracle.grid_encode(m1, table_for(m1).view(np.uint16), x).view(np.float16)[0, 2] == np.float16(0.5)
    m0 = oracle.grid_config(2, 2, 8, 2, max_level=0.5)    # 1 >= 0.501: level 1 is masked, level 0 (0 >= 0.501 false) is not
    out = oracle.grid_encode(m0, table_for(m0).view(np.uint16), x).view(np.float16)
    assert out[0, 0] == np.float16(6.75) and out[0, 2] == 0 and out[0, 3] == 0
    mz = oracle.grid_config(2, 2, 8, 2, max_level=-1.0)   # everything masked
    assert not oracle.grid_encode(mz, table_for(mz).view(np.uint16), x).any()


def test_threaded_network_inference_equals_the_scalar_call(oracle):
    """oracle.network_inference_mt (the C4 band test's value function) cuts the batch over a thread pool: same values, bit for bit"""
    cfg = oracle.grid_config(6, 2, 12, 4, 1.5)
    n_p = oracle.n_params(cfg, 64, 2)
    rng = np.random.default_rng(3)
    params = (rng.normal(size=n_p) * 0.3).astype(np.float16).view(np.uint16)
    coords = rng.random((20000, 3), dtype=np.float32)
    a = oracle.network_inference(cfg, 64, 2, params, coords)
    b = oracle.network_inference_mt(cfg, 64, 2, params, coords, n_threads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.abs(a).max() > 0


def test_dense_and_tiled_grids_hand_cases(oracle):
    """round 4: grid types beside Hash (tcnn_impl_decoder.cu:68-69 passes grid_type to grid_index; EXTERNAL tcnn ctor sizes the levels).
    base 4, 4 levels: resolutions 4, 8, 16, 32.  Dense keeps every level whole (64, 512, 4096, 32768 entries); Tiled caps every level at
    base^3 = 64 entries and indexes with the partial stride walk modulo 64; Hash with T = 2^10 hashes levels 2 and 3."""
    dense = oracle.grid_layout(oracle.grid_config(4, 2, 10, 4, grid_type="Dense"))
    assert list(dense["offsets"]) == [0, 64, 64 + 512, 64 + 512 + 4096, 64 + 512 + 4096 + 32768]
    tiled = oracle.grid_layout(oracle.grid_config(4, 2, 10, 4, grid_type="Tiled"))
    assert list(tiled["offsets"]) == [0, 64, 128, 192, 256]
    hashed = oracle.grid_layout(oracle.grid_config(4, 2, 10, 4))
    assert list(hashed["offsets"]) == [0, 64, 64 + 512, 64 + 512 + 1024, 64 + 512 + 2048]
    import ctypes as C
    L = oracle.lib()
    L.vnro_grid_index_typed.restype = C.c_uint32

    def idx(t, size, res, p):
        return int(L.vnro_grid_index_typed(C.c_uint32(t), C.c_uint32(size), C.c_uint32(res), (C.c_uint32 * 3)(*p)))

    # Dense, res 8 (512 entries): x + 8 y + 64 z
    assert idx(1, 512, 8, (3, 2, 1)) == 3 + 16 + 64
    # Tiled, res 8, 64 entries: the walk takes x (stride 8 <= 64), y (stride 64 <= 64), z (stride 512 > 64 afterwards): x + 8 y + 64 z mod 64
    assert idx(2, 64, 8, (3, 2, 1)) == (3 + 16 + 64) % 64 == 19
    # Tiled, res 16: x (stride 16), y (stride 256 > 64: the walk stops): z does not enter the index
    assert idx(2, 64, 16, (3, 2, 9)) == (3 + 32) % 64 == 35
    assert idx(2, 64, 16, (3, 2, 9)) == idx(2, 64, 16, (3, 2, 0))
    # Tiled, res 128: x only (stride 128 > 64 after the first dimension)
    assert idx(2, 64, 128, (70, 5, 9)) == 70 % 64
    # the same level under Hash hashes instead
    assert idx(0, 64, 128, (70, 5, 9)) == ((70 ^ (5 * 2654435761) ^ (9 * 805459861)) & 0xFFFFFFFF) % 64
    # a modulus that is not a power of two (base 5: 125 entries after next_multiple(125, 8) = 128 is capped at 125)
    t5 = oracle.grid_layout(oracle.grid_config(3, 1, 10, 5, grid_type="Tiled"))
    assert list(np.diff(t5["offsets"].astype(np.int64))) == [125, 125, 125]
    assert idx(2, 125, 10, (9, 9, 9)) == (9 + 90 + 900) % 125
    # encode of a Tiled level with a ramp table: value = index, Nearest picks table[index of the lower corner]
    cfg = oracle.grid_config(2, 1, 10, 4, interpolation=2, grid_type="Tiled")
    table = np.arange(128, dtype=np.float32).astype(np.float16)
    c = np.array([[0.5, 0.25, 0.75]], np.float32)
    enc = oracle.grid_encode(cfg, table.view(np.uint16), c).view(np.float16)[0]
    # level 0: scale 3, pos = 0.5*3+0.5 = 2 -> g = (2, 1, 2): 2 + 4 + 32 = 38;  level 1: scale 7: g = (4, 2, 5): 4 + 16 + 320 = 340 % 64 = 20, + offset 64
    assert enc[0] == 38 and enc[1] == 84


def test_activations_hand_cases_and_derivatives(oracle):
    """round 4: the activations the reference dispatches (tcnn_impl.cu:405-415): known values at 0 and 1 through the C MLP (one
    1 x 16 layer chain with identity-like weights), and the backward formulas of oracle/train_oracle.py (EXTERNAL tcnn
    warp_activation_backward, from the OUTPUT value) against central differences of the forward functions in float64"""
    from oracle import train_oracle as T
    fw = {"None": lambda x: x, "ReLU": lambda x: np.maximum(x, 0), "Exponential": np.exp, "Sigmoid": lambda x: 1 / (1 + np.exp(-x)),
          "Squareplus": lambda x: 0.5 * (10 * x + np.sqrt(100 * x * x + 4)) / 10, "Softplus": lambda x: np.log(np.exp(10 * x) + 1) / 10}
    # forward through the C oracle: W = 16, in = 16, one hidden layer: first layer = identity on feature 0, last layer row 0 picks neuron 0
    W, in_w = 16, 16
    w1 = np.zeros((W, in_w), np.float16); w1[0, 0] = 1
    wl = np.zeros((16, W), np.float16); wl[0, 0] = 1
    weights = np.concatenate([w1.ravel(), wl.ravel()]).view(np.uint16)
    xs = np.array([0.0, 1.0, -1.0, 0.25], np.float32)
    x = np.zeros((xs.size, in_w), np.float16); x[:, 0] = xs
    for name, f in fw.items():
        y = oracle.mlp_forward(weights, in_w, W, 0, x.view(np.uint16), activation=oracle.act_code(name))
        want = f(xs.astype(np.float64)).astype(np.float16).astype(np.float32)
        assert np.allclose(y, want, rtol=2e-3, atol=1e-6), (name, y, want)
        # ... and as the OUTPUT activation on a linear hidden layer
        y2 = oracle.mlp_forward(weights, in_w, W, 0, x.view(np.uint16), activation=oracle.act_code("None", name))
        assert np.allclose(y2, want, rtol=2e-3, atol=1e-6), (name, y2, want)
    assert oracle.mlp_forward(weights, in_w, W, 0, x.view(np.uint16), activation=oracle.act_code("Sigmoid"))[0] == 0.5
    assert oracle.mlp_forward(weights, in_w, W, 0, x.view(np.uint16), activation=oracle.act_code("Exponential"))[0] == 1.0
    assert abs(oracle.mlp_forward(weights, in_w, W, 0, x.view(np.uint16), activation=oracle.act_code("Squareplus"))[0] - 0.1) < 1e-4
    assert abs(oracle.mlp_forward(weights, in_w, W, 0, x.view(np.uint16), activation=oracle.act_code("Softplus"))[0] - np.log(2) / 10) < 1e-4
    # backward: d * f'(x) expressed through y = f(x)
    pts = np.linspace(-0.4, 0.4, 33)
    pts = pts[np.abs(pts) > 1e-3]      # (ReLU's kink)
    for name, f in fw.items():
        y = f(pts).astype(np.float16).astype(np.float32)
        got = T.act_backward(np.ones_like(y), y, name).astype(np.float64)
        h = 1e-5
        want = (f(pts + h) - f(pts - h)) / (2 * h)
        assert np.allclose(got, want, rtol=1e-2, atol=3e-3), (name, np.abs(got - want).max())
