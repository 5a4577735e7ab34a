"""leon_host_anchor_dict_decode alone: ns per symbol on a uniform random dictionary (LEON_LIB = another build to compare with)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leon_amd import capi
if os.environ.get("LEON_LIB"):
    capi.lib_path = lambda: os.environ["LEON_LIB"]
capi.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
for k in (31, 63):
    rng = np.random.default_rng(1)
    km = rng.integers(0, 1 << 62, n, dtype=np.uint64) if k < 32 else np.stack([rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + 1, rng.integers(0, 1 << 62, n, dtype=np.uint64)], axis=1).reshape(-1)
    stream = capi.host_anchor_dict_encode(km, k)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); out = capi.anchor_dict_decode(stream, n, k); best = min(best, time.perf_counter() - t)
    print("k=%d: %d symbols, decode best %.3f s = %.2f ns per symbol, equal to the input: %s" % (k, n * k, best, best / (n * k) * 1e9, bool(np.array_equal(out, km))))
