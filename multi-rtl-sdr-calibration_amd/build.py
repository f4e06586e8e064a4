"""Build libgsmcal.so (HIP, gfx950) in-tree with hipcc.  No torch, no cmake: one translation unit."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gsmcal.hip")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.environ.get("GSMCAL_LIB") or os.path.join(LIB_DIR, "libgsmcal.so")   # GSMCAL_LIB: load another build (tools/devtiming.py)


def _deps():
    d = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))]
    d.append(os.path.join(HERE, "..", "include", "gsmcal.h"))
    return d


def csrc_hash():
    """SHA-256 over the library's sources (csrc/*, include/gsmcal.h, in name order): recorded by tools/profile.sh next to the
    committed PMC / SQ summaries, compared by bench.py and tests/test_abi_cpu.py -- a profile that describes other kernels than
    the ones in the tree is reported as stale, never quoted."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(HERE, "csrc")
    for f in sorted(os.listdir(src)):
        with open(os.path.join(src, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(HERE, "..", "include", "gsmcal.h"), "rb") as fh:
        h.update(b"gsmcal.h\0" + fh.read())
    return h.hexdigest()


def needs_build():
    if os.environ.get("GSMCAL_LIB"):                 # another build was asked for by name: it is loaded as it is, never rebuilt
        if not os.path.exists(LIB):
            raise RuntimeError(f"GSMCAL_LIB={LIB} does not exist")
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False):
    """Compile csrc/gsmcal.hip -> lib/libgsmcal.so for gfx950.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libgsmcal.so cannot be built (there is no CPU fallback)")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", SRC, "-o", LIB + ".tmp", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=HERE)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
