"""GPU parity: fused hash-grid encode + MLP kernel (through the C-ABI) vs the CPU oracle.

Bar: the encode is fp16 bit-exact; the network output is within 2^-8 absolute of the oracle's
fp32-accumulate MLP (the gap between fp16-accumulate and fp32-accumulate MMA)."""
import numpy as np
import pytest

from instantvnr_amd import api
from instantvnr_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL_ABS = 2.0 ** -8

CONFIGS = [
    # (L, F, log2T, base, per_level_scale, hidden_layers)
    (8, 8, 19, 16, None, 2),      # BASELINE C2
    (16, 2, 19, 16, 1.3195, 3),   # BASELINE C4 shape (T=2^19 variant)
    (8, 8, 19, 16, None, 4),      # example-model.json
    (4, 4, 12, 8, None, 1),
    (16, 1, 14, 4, 1.5, 2),
    (5, 2, 10, 3, None, 2),       # padded width 16 with 10 real features
    (12, 4, 15, 8, 1.4, 3),       # padded width 48
    (2, 8, 8, 2, None, 2),
]


def make(oracle, L, F, log2T, base, pls, H, seed=0, interpolation="Linear"):
    cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=log2T, base_resolution=base,
                           n_hidden_layers=H, per_level_scale=pls)
    cfg["encoding"]["interpolation"] = interpolation
    vol = api.vnrCreateNeuralVolume(cfg, (32, 32, 32))
    info = api.neural_info(vol)
    ocfg = oracle.grid_config(L, F, log2T, base, 2.0 if pls is None else pls, 1 if interpolation == "Smoothstep" else 0)
    assert info["n_params"] == oracle.n_params(ocfg, 64, H)
    assert info["padded_width"] == oracle.padded_width(ocfg)
    n_mlp = oracle.mlp_n_params(info["padded_width"], 64, H - 1)
    params = syn.random_params(info["n_params"], n_mlp, seed=seed)
    api.neural_set_params_fp16(vol, params)
    return vol, ocfg, params, n_mlp


def coords_for(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    c[0] = (0, 0, 0); c[1] = (1, 1, 1); c[2] = (0.5, 0.5, 0.5); c[3] = (1, 0, 0.999999)
    return c


@pytest.mark.parametrize("cfg", CONFIGS)
def test_encode_bit_exact(oracle, cfg):
    L, F, log2T, base, pls, H = cfg
    vol, ocfg, params, n_mlp = make(oracle, L, F, log2T, base, pls, H)
    coords = coords_for(3000, 1)  # ragged: not a multiple of 64 or 256
    got = api.neural_encode(vol, coords)
    want = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords).view(np.float16)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("log2T", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("F", [1, 2, 4, 8])
def test_encode_of_tiny_hashed_tables_is_bit_exact(oracle, log2T, F):
    """hashed levels of 1, 2, 4, 8, 32 entries: the x-neighbour of a corner is found inside the aligned group of four entries that holds it
    (grid_device.h gather_corners, F = 2), which a table smaller than a group has to survive; coordinates outside [0, 1] included"""
    vol, ocfg, params, n_mlp = make(oracle, 5, F, log2T, 3, 1.7, 2, seed=40 + log2T)
    coords = coords_for(2000, 41)
    coords[10:60] = np.random.default_rng(42).uniform(-0.3, 1.3, (50, 3)).astype(np.float32)
    got = api.neural_encode(vol, coords)
    want = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords).view(np.float16)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    for _ in range(30):                                  # ... and from the de-hashed image of those tables (built by rows of bricks)
        api.neural_inference(vol, coords[:64])
    assert api.neural_brick_image(vol)["in_use"]
    ok = ~((coords < 0) | (coords > 1)).any(axis=1)      # (outside the unit cube a wave reads the blob again: covered above)
    assert np.array_equal(api.neural_encode(vol, coords).view(np.uint16)[ok], want.view(np.uint16)[ok])


def test_encode_smoothstep_bit_exact(oracle):
    vol, ocfg, params, n_mlp = make(oracle, 6, 2, 12, 4, None, 2, interpolation="Smoothstep")
    coords = coords_for(1000, 2)
    got = api.neural_encode(vol, coords)
    want = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords).view(np.float16)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("cfg", CONFIGS)
def test_inference_within_tolerance(oracle, cfg):
    L, F, log2T, base, pls, H = cfg
    vol, ocfg, params, n_mlp = make(oracle, L, F, log2T, base, pls, H, seed=3)
    coords = coords_for(4097, 4)
    got = api.neural_inference(vol, coords)
    want = oracle.network_inference(ocfg, 64, H, params.view(np.uint16), coords)
    err = np.abs(got - want)
    assert np.isfinite(got).all()
    assert np.abs(want).max() > 0.05          # the test is not vacuous
    assert err.max() <= TOL_ABS * max(1.0, np.abs(want).max()), (err.max(), np.abs(want).max())
    # the bulk is far tighter than the bound (only fp32 summation order + one fp16 rounding differ)
    assert np.median(err) <= 2.0 ** -11 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 8191])
def test_ragged_batch_sizes(oracle, n):
    vol, ocfg, params, n_mlp = make(oracle, 8, 2, 14, 8, None, 2, seed=5)
    coords = coords_for(max(n, 4), 6)[:n]
    got = api.neural_inference(vol, coords)
    want = oracle.network_inference(ocfg, 64, 2, params.view(np.uint16), coords)
    assert got.shape == (n,)
    assert np.abs(got - want).max() <= TOL_ABS * max(1.0, np.abs(want).max())


def test_empty_batch_is_a_noop(oracle):
    vol, *_ = make(oracle, 4, 2, 10, 4, None, 2)
    assert api.neural_inference(vol, np.zeros((0, 3), np.float32)).shape == (0,)


def test_out_of_domain_and_nan_coords_do_not_fault(oracle):
    vol, ocfg, params, n_mlp = make(oracle, 8, 2, 12, 8, None, 2, seed=7)
    coords = np.array([[-0.5, 0.5, 0.5], [1.5, 1.5, 1.5], [np.nan, 0.1, 0.2], [1e9, -1e9, 0.0]], np.float32)
    got = api.neural_encode(vol, coords)
    want = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords[:2]).view(np.float16)
    assert np.array_equal(got[:2].view(np.uint16), want.view(np.uint16))  # exact modulo semantics kept


def test_linearity_of_last_layer_at_scale(oracle):
    """size-independent property at BASELINE scale (1M samples, C4-shaped model): scaling the output layer by
    2 doubles every output exactly (power-of-two scaling commutes with fp16/fp32 rounding)."""
    L, F, log2T, base, pls, H = 16, 2, 19, 16, 1.3195, 3
    vol, ocfg, params, n_mlp = make(oracle, L, F, log2T, base, pls, H, seed=8)
    coords = np.random.default_rng(9).uniform(0, 1, (1 << 20, 3)).astype(np.float32)
    y1 = api.neural_inference(vol, coords)
    p2 = params.copy()
    last = slice(n_mlp - 16 * 64, n_mlp)
    p2[last] = (p2[last].astype(np.float32) * 2).astype(np.float16)
    api.neural_set_params_fp16(vol, p2)
    y2 = api.neural_inference(vol, coords)
    normal = np.abs(y1) >= 2.0 ** -13          # fp16-subnormal outputs do not scale exactly
    assert normal.mean() > 0.99
    assert np.array_equal(y2[normal], 2 * y1[normal])
    assert np.allclose(y2[~normal], 2 * y1[~normal], atol=2.0 ** -22)
    # and a sampled subset agrees with the oracle
    idx = np.random.default_rng(10).choice(coords.shape[0], 2048, replace=False)
    want = oracle.network_inference(ocfg, 64, H, params.view(np.uint16), coords[idx])
    assert np.abs(y1[idx] - want).max() <= TOL_ABS * max(1.0, np.abs(want).max())


def test_params_roundtrip_and_bson(oracle, tmp_path):
    vol, ocfg, params, n_mlp = make(oracle, 4, 4, 12, 8, None, 2, seed=11)
    back = api.neural_get_params_fp16(vol)
    assert np.array_equal(back.view(np.uint16), params.view(np.uint16))
    path = str(tmp_path / "params.json")
    api.vnrNeuralVolumeSerializeParams(vol, path)
    vol2 = api.vnrCreateNeuralVolume(path)   # vnrCreateNeuralVolume(params) overload, BSON file
    coords = coords_for(512, 12)
    assert np.array_equal(api.neural_inference(vol, coords), api.neural_inference(vol2, coords))
    bson = pytest.importorskip("bson")
    doc = bson.decode(open(path, "rb").read())
    assert doc["volume"]["dims"] == {"x": 32, "y": 32, "z": 32}
    assert doc["parameters"]["params_type"] == "__half" and doc["parameters"]["n_params"] == params.size
    assert bytes(doc["parameters"]["params_binary"]) == params.tobytes()
    assert doc["macrocell"]["dims"] == {"x": 2, "y": 2, "z": 2} and len(doc["macrocell"]["data"]) == 8 * 8
    assert doc["model"]["encoding"]["n_levels"] == 4


def test_params_json_of_a_general_model_round_trips(oracle, tmp_path):
    """a params.json written for a model outside the common kind (32 neurons, Squareplus, Sigmoid output, Tiled grid, Nearest) carries its
    model JSON (network.cu:827-857 stores what tcnn_network.h:172-174 kept) and comes back as the same network: same kind, same bits"""
    cfg = syn.model_config(n_levels=6, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=3)
    cfg["network"].update({"n_neurons": 32, "activation": "Squareplus", "output_activation": "Sigmoid"})
    cfg["encoding"].update({"type": "Tiled", "interpolation": "Nearest"})
    vol = api.vnrCreateNeuralVolume(cfg, (32, 32, 32))
    info = api.neural_info(vol)
    n_mlp = oracle.mlp_n_params(info["padded_width"], 32, 2)
    params = syn.random_params(info["n_params"], n_mlp, seed=17)
    api.neural_set_params_fp16(vol, params)
    path = str(tmp_path / "params.json")
    api.vnrNeuralVolumeSerializeParams(vol, path)
    vol2 = api.vnrCreateNeuralVolume(path)
    info2 = api.neural_info(vol2)
    for k in ("n_neurons", "n_hidden_layers", "activation", "output_activation", "grid_type", "interpolation", "n_params"):
        assert info2[k] == info[k], k
    assert (info2["activation"], info2["output_activation"], info2["grid_type"], info2["interpolation"]) == (4, 3, 2, 2)
    coords = coords_for(777, 18)
    assert np.array_equal(api.neural_inference(vol, coords).view(np.uint32), api.neural_inference(vol2, coords).view(np.uint32))
    bson = pytest.importorskip("bson")
    doc = bson.decode(open(path, "rb").read())
    assert doc["model"]["network"]["activation"] == "Squareplus" and doc["model"]["encoding"]["type"] == "Tiled"
    ocfg = oracle.grid_config(6, 2, 12, 4, interpolation=2, grid_type="Tiled")
    want = oracle.network_inference(ocfg, 32, 3, params.view(np.uint16), coords, activation=oracle.act_code("Squareplus", "Sigmoid"))
    assert np.abs(api.neural_inference(vol2, coords) - want).max() <= TOL_ABS


@pytest.mark.parametrize("cfg", [(10, 2, 12, 8, 1.5, 2), (6, 8, 10, 4, None, 2), (7, 4, 11, 4, 1.6, 3), (9, 1, 12, 4, 1.5, 2)])
def test_brick_image_is_lazy_exact_and_dropped_when_parameters_change(oracle, cfg):
    """the de-hashed inference copy of the hashed levels (csrc/network.h) is a cache: built once the parameters have been
    left alone for 24 launches, bit-identical results (coordinates outside [0, 1] and NaN included: those waves read the
    parameter blob), gone as soon as the parameters change"""
    L, F, log2T, base, pls, H = cfg
    vol, ocfg, params, n_mlp = make(oracle, L, F, log2T, base, pls, H, seed=3)
    coords = coords_for(5000, 4)
    coords[10:20] = np.random.default_rng(5).uniform(-0.5, 1.5, (10, 3)).astype(np.float32)   # outside the unit cube
    coords[20] = (np.nan, 0.5, 0.5)
    assert not api.neural_brick_image(vol)["in_use"]
    enc0 = api.neural_encode(vol, coords).view(np.uint16)
    y0 = api.neural_inference(vol, coords).view(np.uint32)
    want = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords)
    ok = ~np.isnan(coords).any(axis=1)
    assert np.array_equal(enc0[ok], want[ok])
    for _ in range(30):
        if api.neural_brick_image(vol)["in_use"]:
            break
        api.neural_inference(vol, coords[:64])
    state = api.neural_brick_image(vol)
    assert state["in_use"] and state["bytes"] > 0
    assert np.array_equal(api.neural_encode(vol, coords).view(np.uint16)[ok], enc0[ok])
    assert np.array_equal(api.neural_inference(vol, coords).view(np.uint32)[ok], y0[ok])
    # new parameters: the image is stale at once, and the next launches read the blob again
    params2 = syn.random_params(len(params), n_mlp, seed=9)
    api.neural_set_params_fp16(vol, params2)
    assert not api.neural_brick_image(vol)["in_use"]
    enc2 = api.neural_encode(vol, coords).view(np.uint16)
    assert np.array_equal(enc2[ok], oracle.grid_encode(ocfg, params2[n_mlp:].view(np.uint16), coords)[ok])
    # (the first image was dropped before it had served 64 launches, so the next one waits twice as long: network.h, the application that trains
    # after every frame)
    assert api.neural_brick_image(vol)["launches_before_next_build"] == 48
    for _ in range(30):
        api.neural_inference(vol, coords[:64])
    assert not api.neural_brick_image(vol)["in_use"]
    for _ in range(30):
        api.neural_inference(vol, coords[:64])
    assert api.neural_brick_image(vol)["in_use"] and api.neural_brick_image(vol)["builds"] == 2
    assert np.array_equal(api.neural_encode(vol, coords).view(np.uint16)[ok], enc2[ok])


def test_free_temporary_gpu_memory_drops_the_caches(oracle):
    """vnrFreeTemporaryGPUMemory (api.h:188; tcnn's free_all_gpu_memory_arenas in the reference): the brick image and the
    training workspace go, results and training state stay"""
    vol, ocfg, params, n_mlp = make(oracle, 10, 2, 12, 8, 1.5, 2, seed=11)
    coords = coords_for(2000, 12)
    y0 = api.neural_inference(vol, coords).view(np.uint32)
    for _ in range(30):
        api.neural_inference(vol, coords[:64])
    assert api.neural_brick_image(vol)["in_use"]
    used = api.vnrMemoryQuery()
    api.vnrFreeTemporaryGPUMemory()
    after = api.vnrMemoryQuery()
    assert not api.neural_brick_image(vol)["in_use"] and api.neural_brick_image(vol)["bytes"] == 0
    assert sum(after) < sum(used)
    assert np.array_equal(api.neural_inference(vol, coords).view(np.uint32), y0)
    for _ in range(30):
        api.neural_inference(vol, coords[:64])
    assert api.neural_brick_image(vol)["in_use"]          # and it comes back


# ------------------------------------------------------------------------------------------------ every shape the reference accepts
# (L, F, log2T, base, pls, H, n_neurons, interpolation, quantize_threshold, max_level[, extras])
# extras: activation / output_activation (tcnn_impl.cu:405-415, tcnn_device_api.h:274-285), grid type (tcnn_impl_decoder.cu:68-69)
ACCEPTED = [
    (8, 2, 12, 4, None, 2, 16, "Linear", 0.0, None),       # FullyFusedMLP WIDTH 16 / 32 / 128 (tcnn_impl.cu:315-347)
    (8, 2, 12, 4, None, 3, 32, "Linear", 0.0, None),
    (6, 4, 12, 4, None, 2, 128, "Smoothstep", 0.0, None),
    (8, 8, 14, 8, None, 2, 128, "Linear", 0.0, None),
    (8, 2, 12, 4, None, 2, 64, "Nearest", 0.0, None),      # tcnn_impl_decoder.cu:73-94
    (5, 1, 10, 3, None, 2, 32, "Nearest", 0.0, None),
    (8, 2, 12, 4, None, 2, 64, "Linear", 0.02, None),      # quantize_threshold (:120): with random fp16 parameters ~ a fifth of the entries are below it
    (8, 4, 12, 4, None, 2, 64, "Smoothstep", 0.05, None),
    (8, 2, 12, 4, None, 2, 64, "Linear", 0.0, 5.0),        # max_level (:17): levels l >= 5.001, i.e. 6 and 7, encode to zero; on the MFMA kernels
    (8, 2, 12, 4, None, 2, 64, "Linear", 0.0, 2.5),        # a fractional bound: levels >= 2.501, i.e. 3 ..
    (8, 2, 12, 4, None, 2, 16, "Nearest", 0.0, 4.0),
    # round 4: the rest of the reference's dispatch
    (8, 2, 12, 4, None, 4, 128, "Linear", 0.0, None),      # 128 neurons, 3 hidden matmuls: a 104 KB weight image, one block of 8 waves per CU
    (12, 8, 14, 4, 1.5, 2, 32, "Linear", 0.0, None),       # encoded width 96 into 32 neurons
    (8, 2, 12, 4, None, 3, 64, "Linear", 0.0, None, {"activation": "Sigmoid"}),
    (8, 2, 12, 4, None, 2, 64, "Linear", 0.0, None, {"activation": "Exponential"}),
    (8, 4, 12, 4, None, 2, 32, "Linear", 0.0, None, {"activation": "Squareplus"}),
    (6, 2, 12, 4, None, 2, 128, "Linear", 0.0, None, {"activation": "Softplus"}),
    (8, 2, 12, 4, None, 2, 16, "Smoothstep", 0.0, None, {"activation": "None"}),
    (8, 2, 12, 4, None, 2, 64, "Linear", 0.0, None, {"output_activation": "Sigmoid"}),
    (8, 2, 12, 4, None, 2, 32, "Linear", 0.0, None, {"activation": "Squareplus", "output_activation": "Exponential"}),
    (6, 2, 12, 4, None, 2, 64, "Linear", 0.0, None, {"output_activation": "ReLU"}),
    (4, 2, 12, 4, None, 2, 64, "Linear", 0.0, None, {"grid_type": "Dense"}),       # levels of 4^3 .. 32^3 grid points, none hashed, no cap
    (5, 4, 12, 4, None, 2, 32, "Smoothstep", 0.0, None, {"grid_type": "Dense"}),
    (8, 2, 12, 4, None, 2, 64, "Linear", 0.0, None, {"grid_type": "Tiled"}),       # every level 4^3 entries, finer levels wrap (2 or 3 index dimensions)
    (6, 8, 12, 8, 1.5, 2, 128, "Nearest", 0.0, None, {"grid_type": "Tiled"}),
    (8, 1, 12, 5, None, 2, 16, "Linear", 0.0, None, {"grid_type": "Tiled"}),       # 5^3 = 125 entries: a modulus that is not a power of two
    # weight images beyond the 160 KiB of LDS (5 hidden matmuls x 32 KB): the A operands come from global memory
    (6, 2, 12, 4, None, 6, 128, "Linear", 0.0, None),
    (8, 2, 12, 4, None, 7, 128, "Linear", 0.0, None, {"activation": "Sigmoid"}),
    (8, 4, 12, 4, None, 2, 32, "Linear", 0.03, None, {"activation": "Squareplus"}),  # quantize_threshold with everything else GENERAL
    (8, 2, 12, 4, None, 21, 64, "Linear", 0.0, None),      # 64 neurons, 20 hidden matmuls x 8 KB: beyond the LDS at this width too (tests/test_gpu_fuzz.py: 32 / 16 neurons)
]
INTERP = {"Linear": 0, "Smoothstep": 1, "Nearest": 2}


@pytest.mark.parametrize("case", ACCEPTED)
def test_models_the_reference_accepts_load_and_evaluate(oracle, case):
    """encode bit-exact, output within 2^-8 of the oracle, gradients within 3 % of the numpy restatement and 60 steps that converge, for
    every FullyFusedMLP width, interpolation, activation, output activation and grid type the reference's dispatch accepts, incl.
    quantize_threshold and 128-neuron models whose weight image exceeds the LDS: all on the MFMA kernels since round 4."""
    L, F, log2T, base, pls, H, W, interp, qt, max_level = case[:10]
    extra = case[10] if len(case) > 10 else {}
    act, out_act, gtype = extra.get("activation", "ReLU"), extra.get("output_activation", "None"), extra.get("grid_type", "Hash")
    cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=log2T, base_resolution=base, n_hidden_layers=H, per_level_scale=pls)
    cfg["encoding"]["interpolation"] = interp
    cfg["network"]["n_neurons"] = W
    cfg["network"]["activation"] = act
    cfg["network"]["output_activation"] = out_act
    if gtype != "Hash":
        cfg["encoding"]["type"] = gtype
    if qt:
        cfg["encoding"]["quantize_threshold"] = qt
    if max_level is not None:
        cfg["encoding"]["max_level"] = max_level
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(16))
    vol = api.vnrCreateNeuralVolume(cfg, sv)
    info = api.neural_info(vol)
    assert info["n_neurons"] == W
    assert info["mfma_kernels"] == 1 and info["mfma_training_kernels"] == 1     # since round 4 there is no other kind of kernel
    ocfg = oracle.grid_config(L, F, log2T, base, 2.0 if pls is None else pls, INTERP[interp], qt, 1000.0 if max_level is None else max_level, gtype)
    assert info["n_params"] == oracle.n_params(ocfg, W, H)
    n_mlp = oracle.mlp_n_params(info["padded_width"], W, H - 1)
    # Exponential / Softplus (x 10 inside) grow fast: smaller weights keep the activations in fp16's range, as a trained model's are
    params = syn.random_params(info["n_params"], n_mlp, seed=11, mlp_scale=0.35 if act in ("Exponential", "Softplus") or out_act == "Exponential" else 1.0)
    api.neural_set_params_fp16(vol, params)
    coords = coords_for(2049, 12)
    enc = api.neural_encode(vol, coords)
    want_enc = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords)
    assert np.array_equal(enc.view(np.uint16), want_enc)
    if max_level is not None:   # the masked levels really are zero, the others are not
        first_masked = int(np.ceil(max_level + 1e-3 - 1e-9))
        assert not enc[:, first_masked * F:].any() and enc[:, :first_masked * F].any()
    got = api.neural_inference(vol, coords)
    code = oracle.act_code(act, out_act)
    want = oracle.network_inference(ocfg, W, H, params.view(np.uint16), coords, activation=code)
    assert np.isfinite(got).all() and np.isfinite(want).all() and np.abs(want).max() > 0.01
    assert np.abs(got - want).max() <= TOL_ABS * max(1.0, np.abs(want).max()), (np.abs(got - want).max(), np.abs(want).max())
    # ... and trains: gradients of one batch against the numpy restatement at the bar of tests/test_gpu_train.py
    from oracle import train_oracle as T
    rng = np.random.default_rng(21)
    B = 700
    tc = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    tt = rng.uniform(0, 1, B).astype(np.float32)
    grads = api.neural_forward_backward(vol, tc, tt).astype(np.float64)
    ref = T.training_gradients(ocfg, W, H, params.view(np.uint16), tc, tt, loss="L1", activation=act, output_activation=out_act)
    assert np.isclose(api.vnrNeuralVolumeGetTrainingLoss(vol), ref["loss"], rtol=2e-3)
    for name, sl in [("mlp", slice(0, n_mlp)), ("grid", slice(n_mlp, None))]:
        g, w = grads[sl], ref["grads"][sl]
        assert np.abs(w).max() > 0
        rel = np.linalg.norm(g - w) / np.linalg.norm(w)
        assert rel < 3e-2, (name, rel)
        assert np.abs(g - w).max() < 6e-2 * np.abs(w).max(), name
    last = grads[n_mlp - 16 * W:n_mlp].reshape(16, W)
    assert np.all(last[1:] == 0) and np.any(last[0] != 0)
    if max_level is not None:   # masked levels get no gradient
        lay = oracle.grid_layout(ocfg)
        first_masked = int(np.ceil(max_level + 1e-3 - 1e-9))
        assert not grads[n_mlp + int(lay["offsets"][first_masked]) * F:].any()
    # and an optimizer step moves the parameters; 60 steps on the analytic volume bring the loss down
    api.neural_train_end(vol)
    assert not np.array_equal(api.neural_get_params_fp16(vol).view(np.uint16), params.view(np.uint16))
    api.vnrNeuralVolumeTrain(vol, 1, True)
    first = api.vnrNeuralVolumeGetTrainingLoss(vol)
    api.vnrNeuralVolumeTrain(vol, 60, True)
    assert np.isfinite(api.vnrNeuralVolumeGetTrainingLoss(vol)) and api.vnrNeuralVolumeGetTrainingLoss(vol) < first


def test_reconfiguring_a_model_does_not_reuse_the_old_models_training_scratch(oracle):
    """the tile lists of the dense levels' LDS scatter and the zeroed part of the weight-gradient slab are cached per
    network; SetModel to a shape with the same batch and level count (n_features 4 -> 2, n_neurons 64 -> 32) must give the gradients of a
    freshly created model of that shape"""
    from oracle import train_oracle as T
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(16))
    cfg_a = syn.model_config(n_levels=6, n_features=4, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
    cfg_b = syn.model_config(n_levels=6, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
    cfg_b["network"]["n_neurons"] = 32
    vol = api.vnrCreateNeuralVolume(cfg_a, sv)
    rng = np.random.default_rng(5)
    tc = rng.uniform(0, 1, (4096, 3)).astype(np.float32)
    tt = rng.uniform(0, 1, 4096).astype(np.float32)
    api.neural_forward_backward(vol, tc, tt)
    api.neural_train_end(vol)
    api.vnrNeuralVolumeSetModel(vol, cfg_b)
    info = api.neural_info(vol)
    n_mlp = oracle.mlp_n_params(info["padded_width"], 32, 1)
    params = syn.random_params(info["n_params"], n_mlp, seed=3)
    api.neural_set_params_fp16(vol, params)
    got = api.neural_forward_backward(vol, tc, tt).astype(np.float64)
    fresh = api.vnrCreateNeuralVolume(cfg_b, sv)
    api.neural_set_params_fp16(fresh, params)
    want = api.neural_forward_backward(fresh, tc, tt).astype(np.float64)
    # packed fp16 atomics arrive in any order on the hashed levels: equal up to that, and equal to the restatement at the usual bar
    assert np.linalg.norm(got - want) <= 2e-3 * np.linalg.norm(want)
    assert np.array_equal(got[n_mlp - 16 * 32 + 32:n_mlp], np.zeros(15 * 32))      # the padded rows of the last layer
    ocfg = oracle.grid_config(6, 2, 12, 4)
    ref = T.training_gradients(ocfg, 32, 2, params.view(np.uint16), tc, tt, loss="L1")["grads"]
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 3e-2


@pytest.mark.parametrize("cfg", [(16, 2, 22, 16, 1.3195, "Hash"), (16, 2, 19, 16, float(np.exp(np.log(1024 / 16.0) / 15)), "Hash"), (8, 8, 19, 16, 2.0, "Hash"),
                                 (5, 4, 12, 4, 1.5, "Dense"), (9, 2, 12, 6, 1.7, "Tiled")])
def test_the_level_table_the_library_reports_is_the_oracle_s_layout(oracle, cfg):
    """vnrAmdNeuralVolumeLevelTable (what bench.py prices the training scatter with): resolution, entries and offset of every level equal the
    oracle's restatement of tcnn's level sizing (EXTERNAL), for a per_level_scale whose fp32 power lands within an ulp of an integer too
    (1024 / 16)^(1/15): level 5 is 65 grid points, not 64)"""
    L, F, log2T, base, pls, gtype = cfg
    c = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=log2T, base_resolution=base, n_hidden_layers=2, per_level_scale=pls)
    if gtype != "Hash": c["encoding"]["type"] = gtype
    vol = api.vnrCreateNeuralVolume(c, (32, 32, 32))
    got = api.neural_level_table(vol)
    lay = oracle.grid_layout(oracle.grid_config(L, F, log2T, base, pls, 0, 0.0, 1000.0, gtype))
    assert len(got) == L
    assert [g["offset"] for g in got] == [int(x) for x in lay["offsets"][:L]]
    assert [g["entries"] for g in got] == [int(x) for x in np.diff(lay["offsets"].astype(np.int64))]
    assert [g["res"] for g in got] == [int(x) for x in lay["resolution"][:L]] if "resolution" in lay else True
    assert all((g["kind"] == 0) == (g["res"] ** 3 <= g["entries"] or gtype == "Dense") for g in got if gtype != "Tiled")
    # vnrAmdNeuralVolumeGridBackwardPlan (what bench.py's train_roofline is priced with, from the library itself): the levels scattered through
    # LDS tiles are the leading dense levels of at most 64 tiles of 24 KB; every other level costs one 64-byte request per sample and yz row
    plan = api.neural_grid_backward_plan(vol, 65536)
    tile = (24 * 1024 // (4 * F)) & ~15
    lds = 0
    while lds < L and got[lds]["kind"] == 0 and -(-got[lds]["entries"] // tile) <= 64 and (got[lds]["offset"] * F) % 2 == 0:
        lds += 1
    assert plan["n_levels"] == L and plan["tile_entries"] == tile and plan["lds_levels"] == lds, (plan, lds)
    assert plan["atomic_requests"] == 65536 * 4 * (L - lds)
    assert 0 <= plan["flush_requests_at_most"] <= sum(128 * (g["entries"] * F * (4 if F == 1 else 2) // 64 + 1) for g in got[:lds])


def test_two_shapes_with_the_same_mlp_size_do_not_share_the_weight_gradient_slab():
    """64 neurons x 1 hidden layer and 32 neurons x 2 hidden layers on a 16-wide encoding both have 2 048 MLP parameters, with
    other layouts.  The slab's never-written elements (rows 1 .. 15 of the padded last layer) were zeroed per n_mlp only, so after the
    re-configuration they held the first shape's hidden-layer sums and were added into the gradient of the padded rows"""
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(16))
    cfg_a = syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=1, n_neurons=64)
    cfg_b = syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2, n_neurons=32)
    vol = api.vnrCreateNeuralVolume(cfg_a, sv)
    assert api.neural_info(vol)["padded_width"] == 16
    rng = np.random.default_rng(11)
    tc = rng.uniform(0, 1, (2048, 3)).astype(np.float32)
    tt = rng.uniform(0, 1, 2048).astype(np.float32)
    ga = api.neural_forward_backward(vol, tc, tt)
    n_mlp = 64 * 16 + 16 * 64
    assert np.abs(ga[:n_mlp]).max() > 0 and np.all(ga[n_mlp - 15 * 64:n_mlp] == 0)
    api.neural_train_end(vol)
    api.vnrNeuralVolumeSetModel(vol, cfg_b)
    info = api.neural_info(vol)
    assert 32 * 16 + 32 * 32 + 16 * 32 == n_mlp                               # the same MLP size, another layout
    params = syn.random_params(info["n_params"], n_mlp, seed=4)
    api.neural_set_params_fp16(vol, params)
    got = api.neural_forward_backward(vol, tc, tt)
    fresh = api.vnrCreateNeuralVolume(cfg_b, sv)
    api.neural_set_params_fp16(fresh, params)
    want = api.neural_forward_backward(fresh, tc, tt)
    assert np.array_equal(got[:n_mlp], want[:n_mlp])                             # the MLP's gradient is summed in block order: bit for bit
    assert np.all(got[n_mlp - 15 * 32:n_mlp] == 0)                              # the padded rows of the last layer


def test_repeated_gradient_batches_and_reconfigurations_never_miss():
    """the hunt for round 4's once-in-28 000 grid-gradient transient as a regression test (tests/diag/grad_hammer.py, short form): the sweep's
    draw and four other models, every repetition compared ON THE DEVICE with the first one -- one volume per model, a fresh volume per
    repetition, one volume re-configured between the models.  3.25 M such checks ran clean in round 5 (profiles/r05_grad_hammer.txt)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "diag", "grad_hammer.py"), "20000", "600", "3000", "/tmp/vnr_grad_hammer_test.txt"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "TOTAL events: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


RENDERED = [
    # (n_neurons, interpolation, activation, output_activation, grid type, rendering mode)
    (32, "Linear", "ReLU", "None", "Hash", 5),
    (128, "Linear", "ReLU", "None", "Hash", 5),          # blocks of 8 waves: the renderer launches the ray packing itself
    (16, "Nearest", "ReLU", "None", "Hash", 5),
    (64, "Linear", "Sigmoid", "None", "Hash", 5),        # a GENERAL instance behind the renderer's sample queue
    (64, "Linear", "ReLU", "Sigmoid", "Tiled", 5),
    (32, "Linear", "ReLU", "None", "Hash", 8),           # gradient shading: four evaluations per sample
    (128, "Smoothstep", "Squareplus", "None", "Dense", 14),   # path tracing: not a 64-neuron common-kind model, so the streaming path tracer
]


@pytest.mark.parametrize("case", RENDERED)
def test_a_rendered_frame_of_every_kind_of_model_equals_the_oracle(oracle, case):
    """the renderer's sample queue goes through the same dispatch as vnrNeuralVolumeInference: frames of models of every width / kind
    equal the oracle's marcher driven by the oracle's network (PSNR > 40 dB on random parameters; modes 5 / 8 / 14)"""
    W, interp, act, out_act, gtype, mode = case
    cfg = syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
    cfg["network"]["n_neurons"] = W
    cfg["network"]["activation"] = act
    cfg["network"]["output_activation"] = out_act
    cfg["encoding"]["interpolation"] = interp
    if gtype != "Hash":
        cfg["encoding"]["type"] = gtype
    vol = syn.analytic_volume(32)
    sv = api.vnrCreateSimpleVolume(vol)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    info = api.neural_info(nv)
    assert info["mfma_kernels"] == 1
    n_mlp = oracle.mlp_n_params(info["padded_width"], W, 1)
    params = syn.random_params(info["n_params"], n_mlp, seed=13)
    api.neural_set_params_fp16(nv, params)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas); api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera((32, 32, 32))
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetTransferFunction(ren, tfn); api.vnrRendererSetCamera(ren, camera); api.vnrRendererSetFramebufferSize(ren, (64, 48))
    api.vnrRendererSetMode(ren, mode)
    api.vnrRender(ren)
    img = api.vnrRendererMapFrame(ren).copy()
    ocfg = oracle.grid_config(4, 2, 12, 4, interpolation=INTERP[interp], grid_type=gtype)
    mo = api.volume_macrocell(nv)["max_opacity"]
    code = oracle.act_code(act, out_act)
    net = lambda c: oracle.network_inference(ocfg, W, 2, params.view(np.uint16), c, activation=code)   # noqa: E731
    kw = {}
    if mode == 8:
        kw = {"shading_mode": 1}
    sc = oracle.SceneHolder(64, 48, (32, 32, 32), oracle.TfnHolder(colors, alphas), mo, cam["from"], cam["at"], cam["up"], cam["fovy"], **kw)
    if mode == 14:
        ref, _, _ = oracle.render_pathtracing(sc, net)
    else:
        ref, _, _ = oracle.render_streaming(sc, net)
    if mode == 14:   # individual paths differ where the two networks' values differ in the last bits (a tracking decision flips): compare coarsely
        assert (img[..., 3] == 1.0).all()
        assert abs(float(img[..., :3].mean()) - float(ref[..., :3].mean())) < 0.15 * float(ref[..., :3].mean())
        assert np.corrcoef(img[..., :3].reshape(-1), ref[..., :3].reshape(-1))[0, 1] > 0.8
        return
    mse = float(((img - ref) ** 2).mean())
    assert img[..., 3].max() > 0.002 and 10 * np.log10(1.0 / max(mse, 1e-20)) > 40.0, (case, 10 * np.log10(1.0 / max(mse, 1e-20)))   # (not vacuous: something is visible)


@pytest.mark.parametrize("L, F", [(17, 8), (33, 4), (33, 2)])
def test_encoding_shapes_without_a_kernel_instance_are_refused_when_the_model_is_set(oracle, L, F):
    """the fused kernels are instantiated for every padded encoded width up to 32 levels of 1, 2 or 4 features and 16 levels of 8
    (infer_kernel.h); anything wider fails by name when the model is set, not at the first launch, and never falls back to another path"""
    cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=10, base_resolution=2, n_hidden_layers=2, per_level_scale=1.1)
    with pytest.raises(RuntimeError, match="unsupported encoding shape|too many hash-grid levels"):
        api.vnrCreateNeuralVolume(cfg, (16, 16, 16))


@pytest.mark.parametrize("F,budget,want", [(8, 4_800_000, [2, 3]), (2, 1_450_000, [2, 4])])
def test_which_levels_the_inference_cache_takes_when_not_all_fit(oracle, F, budget, want):
    """network_host.hip build_brick_image: big bricks (F <= 2: 32 / 64 entries) take the FINEST hashed levels that fit, small bricks (F >= 4:
    2 x 2 x 2 cells for F = 8) the COARSEST (round 6: the reference's example model on a 1024^3 volume spent 17.6 GB on its finest level for
    nothing).  Hashed levels 2..5 of resolution 16 / 32 / 64 / 128; the budget holds level 4 and one small level, or levels 2 and 3.  Bit-exact
    either way."""
    from instantvnr_amd._lib import check, lib
    vol, ocfg, params, n_mlp = make(oracle, 6, F, 10, 4, 2.0, 2, seed=50 + F)
    coords = coords_for(1500, 51)
    enc0 = api.neural_encode(vol, coords).view(np.uint16)
    api.neural_set_brick_budget(vol, budget)
    check(lib().vnrAmdNeuralVolumeSetBrickImageMode(vol.h, 1))
    api.neural_inference(vol, coords[:64])
    st = api.neural_brick_image(vol)
    assert st["in_use"] and st["bytes"] <= budget
    assert [l for l in range(16) if st["levels"] >> l & 1] == want, st
    assert np.array_equal(api.neural_encode(vol, coords).view(np.uint16), enc0)
