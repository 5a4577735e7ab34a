"""Builds leon_amd/lib/libleon_dna.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_library(jobs=4, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(jobs)]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    so = os.path.join(_HERE, "lib", "libleon_dna.so")
    if not os.path.exists(so):
        raise RuntimeError("build did not produce " + so)
    return so
