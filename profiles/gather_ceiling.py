"""Random 64-byte-sector gather ceiling of one MI355X: every lane reads one dword from a pseudo-random sector of a table
the size of the bench's bloom filter (752 MiB) or of 8 GiB, independent loads (dep=0) or each address depending on
the value loaded before (dep=1, like the bloom walk).  `python profiles/gather_ceiling.py` builds the kernel with hipcc
and prints G sectors/s; profiles/README.md quotes the result next to k_walk's PMC traffic."""
import ctypes, os, subprocess, tempfile

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.mkdtemp(), "gather_ceiling.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                       os.path.join(here, "gather_ceiling.hip")])
lib = ctypes.CDLL(so)
lib.run_gather.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
tables = [int(x) << 20 for x in os.environ.get("LEON_GATHER_TABLES_MIB", "752,8192").split(",")]
for table in tables:
    for dep in (0, 1):
        for blocks in (2048, 8192, 32768):
            iters = 256 if dep == 0 else 64
            ms = ctypes.c_float()
            rc = lib.run_gather(table, blocks, iters, dep, ctypes.byref(ms))
            acc = blocks * 256 * iters
            print(f"table={table >> 20:6d} MiB dep={dep} blocks={blocks:6d} rc={rc} ms={ms.value:8.3f}  "
                  f"{acc / ms.value / 1e6:8.2f} G sectors/s  {acc * 64 / ms.value / 1e6:8.1f} GB/s (64 B/sector)", flush=True)
