"""-m "not gpu": AddressSanitizer + UBSan over the host-side decoders of the C-ABI (no GPU sanitizer exists on this pool, so
the host code is what a sanitizer can see): host_streams.cpp is rebuilt with -fsanitize=address,undefined and fed valid,
corrupted, truncated and random header / quality payloads in a child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_decoders_under_asan_ubsan(tmp_path):
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no libasan / libubsan in this toolchain")
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    stub = tmp_path / "stub.cpp"            # the two symbols host_streams.cpp takes from the rest of the library
    stub.write_text('#include <string>\nnamespace leon { thread_local std::string g; void set_create_error(const std::string& m) { g = m; } }\n'
                    'extern "C" const char* leon_last_error(const void*) { return leon::g.c_str(); }\n')
    so = str(tmp_path / "libhoststreams_asan.so")
    subprocess.check_call(["g++"] + flags + ["-shared", "-I" + os.path.join(ROOT, "include"), "-o", so,
                                             os.path.join(ROOT, "leon_amd", "csrc", "host_streams.cpp"), str(stub), "-lz", "-lpthread"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_fuzz_host_streams.py"), so], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "header decoder under ASan/UBSan: ok" in r.stdout and "quality decoder under ASan/UBSan: ok" in r.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_dictionary_chain_worker_under_sanitizers(tmp_path, san):
    """the dictionary chain of round 3 is a small pipeline of threads (helpers that prepare per-symbol records in a ring of
    buffers, the chain that consumes them; host_rc.h): ThreadSanitizer and ASan/UBSan over the whole worker, one- and two-word
    k-mers, streams long enough to go round the ring several times"""
    if not _lib("libtsan.so" if san == "thread" else "libasan.so"):
        pytest.skip("no sanitizer runtime in this toolchain")
    exe = str(tmp_path / "chain_san")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", exe,
                           os.path.join(ROOT, "profiles", "scripts", "chain_ab", "ab_new.cpp"), "-lpthread"])
    seen = {}
    # ~1.9 M symbols each: 58 segments round a ring of 8 buffers several times, inside the default 512, and with one helper whose five
    # spares do most of the work (the look-ahead never leaves the low-water zone of a ring of 64)
    for n, k, ring, helpers, spares in ((60000, 31, "8", "3", "3"), (30000, 63, "8", "3", "3"), (60000, 31, "512", "3", "3"), (60000, 31, "64", "1", "5")):
        r = subprocess.run([exe, str(n), str(k)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, LEON_CHAIN_RING=ring, LEON_CHAIN_HELPERS=helpers, LEON_CHAIN_SPARES=spares,
                                    TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
        out = r.stdout + r.stderr
        assert r.returncode == 0 and "Sanitizer" not in out and "runtime error" not in out, out[-3000:]
        fnv = {l.split("fnv")[1].strip() for l in r.stdout.splitlines() if "fnv" in l}
        assert len(fnv) == 1, "three runs of the same stream must give the same bytes: %s" % fnv
        seen.setdefault((n, k), set()).update(fnv)
    assert all(len(v) == 1 for v in seen.values()), "the same stream under another ring / helper configuration: other bytes %s" % seen
