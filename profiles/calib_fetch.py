"""Calibration of the FETCH_SIZE PMC counter for random dword gathers on gfx950 (MI355X_MICROARCH.md, HBM section:
'other access widths are uncalibrated: calibrate on a known byte count in your own access pattern').
Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python profiles/calib_fetch.py
Kernel A streams 2 GiB (known bytes); kernel B gathers 2^27 random dwords from a 8 GiB table (each lands in its own
64-byte sector with overwhelming probability, far beyond L2 and the Infinity Cache)."""
import torch

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
table = torch.empty(1 << 31, dtype=torch.int32, device=dev)           # 8 GiB
table.fill_(3)
n = 1 << 27
idx = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
torch.cuda.synchronize()
stream_src = table[: 1 << 29]                                          # 2 GiB
a = stream_src.sum()                                                   # kernel A: reduce (coalesced stream)
torch.cuda.synchronize()
b = torch.take(table, idx)                                             # kernel B: 2^27 random dword gathers (+ 1 GiB idx stream)
torch.cuda.synchronize()
print("stream bytes", (1 << 29) * 4, "gathers", n, "idx bytes", n * 8, int(a.item()), int(b[0].item()))
