"""One stream of 2^30 + 2^20 symbols on ONE model through the device range coder (leon_rc_encode_streams): from total 2^30 on
the coder divides exactly instead of multiply-high + 32-bit fix-up.  Compared with the oracle's coder byte for byte.  A one-off
(~3 min of one wave's serial chain): not part of the suite."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import leon_amd  # noqa: E402
import oracle_lib as O  # noqa: E402

n = (1 << 30) + (1 << 20)
rng = np.random.default_rng(3)
model = 1                                     # a 5-symbol model (model sizes as in tests/test_gpu_parity.py)
vals = rng.choice(4, size=n, p=[0.55, 0.25, 0.15, 0.05]).astype(np.uint8)
syms = np.empty(2 * n, dtype=np.uint8)
syms[0::2] = model
syms[1::2] = vals
begin = np.array([0, n], dtype=np.uint64)
ctx = leon_amd.DnaEncodeContext(kmer_size=31, reads_per_block=1000, bloom_tai=1000)
t0 = time.time()
got = ctx.rc_encode_streams(syms, begin)[0]
t1 = time.time()
sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
exp = O.rc_encode_stream(syms[0::2], syms[1::2], sizes)
t2 = time.time()
print("symbols %d: device %.1f s (%d bytes), oracle %.1f s (%d bytes), equal: %s" % (n, t1 - t0, len(got), t2 - t1, len(exp), got == exp))
assert got == exp
