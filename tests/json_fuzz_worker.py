"""child process of tests/test_json_fuzz.py: feeds mutated JSON text and BSON bytes to vnrAmdJsonConvert (the library's own parser and
writer, csrc/json.cpp -- the reference uses nlohmann::json for its scene / model / params.json files).  A parser that reads out of
bounds, recurses without a limit or loops dies here, and the parent sees the exit code.  With a third argument every input is also appended to that file (u8 format, u32 length, bytes):
the corpus tests/test_json_fuzz.py feeds to a build of csrc/json.cpp under AddressSanitizer / UBSan.  usage: json_fuzz_worker.py <seed> <n> [corpus]"""
import ctypes as C
import json
import struct
import sys

import numpy as np

from instantvnr_amd import _lib, api

L = _lib.lib()
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
out, size = C.c_void_p(), C.c_size_t()


CORPUS = open(sys.argv[3], "wb") if len(sys.argv) > 3 else None


def convert(data, fin, fout):
    if CORPUS is not None and len(data) < (1 << 17):
        CORPUS.write(struct.pack("<BI", fin, len(data)) + bytes(data))
    rc = L.vnrAmdJsonConvert(data, len(data), fin, fout, C.byref(out), C.byref(size))
    if rc != 0:
        assert L.vnrAmdGetLastError(), "an error without a message"
        return None
    b = C.string_at(out, size.value)
    L.vnrAmdFreeHost(out)
    return b


def random_key(j):
    return "k%d" % j + "".join(chr(int(c)) for c in rng.choice([34, 92, 32, 65, 0xe9, 0x4e2d, 46, 36], int(rng.integers(0, 3))))


def random_value(depth=0):
    k = rng.integers(0, 9 if depth < 5 else 6)
    if k == 0: return int(rng.integers(-2**31, 2**31))
    if k == 1: return int(rng.integers(-2**62, 2**62))
    if k == 2: return float(np.float64(rng.normal()) * 10.0 ** float(rng.integers(-30, 30)))
    if k == 3: return bool(rng.integers(0, 2))
    if k == 4: return None
    if k == 5: return "".join(chr(int(c)) for c in rng.choice([34, 92, 47, 8, 9, 10, 13, 32, 65, 97, 0xe9, 0x4e2d, 0x1f600, 127, 1], int(rng.integers(0, 12))))
    if k in (6, 7): return {random_key(j): random_value(depth + 1) for j in range(int(rng.integers(0, 5)))}
    return [random_value(depth + 1) for _ in range(int(rng.integers(0, 5)))]


def equal(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return isinstance(a, (int, float)) and isinstance(b, (int, float)) and (a == b or abs(a - b) <= 1e-15 * max(abs(a), abs(b)))
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(equal(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return isinstance(b, list) and len(a) == len(b) and all(equal(x, y) for x, y in zip(a, b))
    return type(a) is type(b) and a == b


SEED_TEXT = b"""{ // scene
  "volume": {"dims": {"x": 64, "y": 64, "z": 64}, "data": [{"format": "REGULAR_GRID_RAW_BINARY", "fileName": "a.raw", "endian": "LITTLE_ENDIAN"}]},
  "model": {"loss": {"otype": "L1"}, "optimizer": {"otype": "ExponentialDecay", "decay_base": 0.33, "nested": {"otype": "Adam", "learning_rate": 1e-2}},
            "encoding": {"otype": "HashGrid", "n_levels": 8, "per_level_scale": 2.0}, "network": {"n_neurons": 64, "activation": "ReLU"}},
  /* c */ "arr": [1, -2.5e+3, "x\\u00e9\\n\\"", null, true, false, [], {}], "big": 5000000000 }"""
SEED_BSON = convert(SEED_TEXT, api.JSON_TEXT, api.JSON_BSON)
assert SEED_BSON
TOKENS = [b"{", b"}", b"[", b"]", b",", b":", b'"', b"\\", b"\\u", b"\\ud800", b"//", b"/*", b"*/", b"-", b"+", b"e", b"E", b".", b"0", b"1e999", b"-0",
          b"nul", b"tru", b"NaN", b"Infinity", b"\0", b"\xff", b"\xc3", b"\xf0\x9f", b"99999999999999999999999999"]
counts = {"text ok": 0, "text error": 0, "bson ok": 0, "bson error": 0, "round trips": 0}
for i in range(n):
    # 1. valid random documents: text -> BSON -> text keeps the value; BSON -> BSON is the identity
    if i % 4 == 0:
        doc = {"k%d" % j: random_value() for j in range(int(rng.integers(0, 6)))}
        text = json.dumps(doc, ensure_ascii=bool(rng.integers(0, 2))).encode()
        b = convert(text, api.JSON_TEXT, api.JSON_BSON)
        assert b is not None, (text, L.vnrAmdGetLastError())
        back = convert(b, api.JSON_BSON, api.JSON_TEXT)
        assert back is not None and equal(json.loads(back), doc), (doc, back)
        assert convert(b, api.JSON_BSON, api.JSON_BSON) == b
        counts["round trips"] += 1
    # 2. mutated text
    t = bytearray(SEED_TEXT)
    for _ in range(int(rng.integers(1, 6))):
        k = rng.integers(0, 5)
        p = int(rng.integers(0, len(t) + 1))
        if k == 0 and len(t): t[min(p, len(t) - 1)] = int(rng.integers(0, 256))
        elif k == 1: t[p:p] = TOKENS[int(rng.integers(0, len(TOKENS)))]
        elif k == 2: del t[p:p + int(rng.integers(1, 20))]
        elif k == 3: t = t[:p]
        else: t[p:p] = (b"[" if rng.integers(0, 2) else b'{"a":') * int(rng.choice([3, 70, 3000, 200000]))
    r = convert(bytes(t), api.JSON_TEXT, api.JSON_BSON)
    counts["text ok" if r is not None else "text error"] += 1
    if r is not None:                                    # whatever it accepted must survive its own writer and reader
        assert convert(r, api.JSON_BSON, api.JSON_TEXT) is not None, (bytes(t)[:3000], L.vnrAmdGetLastError())
    # 3. mutated BSON
    b = bytearray(SEED_BSON)
    for _ in range(int(rng.integers(1, 5))):
        k = rng.integers(0, 4)
        p = int(rng.integers(0, len(b)))
        if k == 0: b[p] = int(rng.integers(0, 256))
        elif k == 1 and p + 4 <= len(b): struct.pack_into("<i", b, p, int(rng.choice([-1, 0, 1, 4, 5, 2**31 - 1, -2**31, len(b), len(b) + 1, int(rng.integers(-1000, 100000))])))
        elif k == 2: b = b[:max(1, p)]
        else: b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
    r = convert(bytes(b), api.JSON_BSON, api.JSON_TEXT)
    counts["bson ok" if r is not None else "bson error"] += 1
    if r is not None:
        assert convert(r, api.JSON_TEXT, api.JSON_BSON) is not None, (bytes(b), r)
print(json.dumps(counts))
