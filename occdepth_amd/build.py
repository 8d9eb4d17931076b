"""Build libocc_hip.so (gfx950 only) in-tree with hipcc.

    python -m occdepth_amd.build [--force]

The shared object lands next to this file so it travels with the source tree
(it is git-ignored, not gpurun-ignored).  No torch headers are involved: the
library exposes the C ABI of include/occdepth_amd.h only.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv3d_igemm.hip", "conv3d_c32p.hip", "lift.hip", "nchw2d.hip", "loss.hip", "conv3d_wgrad.hip", "conv3d_bf16.hip", "bn.hip", "bneck3d.hip", "rows_gemm.hip", "gemm_x3.hip", "wino2d.hip", "wino_conv2d.hip", "pw_gemm.hip", "se2d.hip", "ipc_allreduce.hip", "graph_fix.hip", "prof.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "occdepth_amd.h")]
LIB = os.path.join(HERE, "libocc_hip.so")
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libocc_hip.so cannot be built")
    return exe


def source_digest():
    """sha256 over the names and CONTENTS of every source and header the library is built from (+ the compiler flags)."""
    import hashlib
    h = hashlib.sha256()
    h.update(f"arch={ARCH};flags=-O3 -std=c++17 -fPIC".encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


STAMP = LIB + ".srchash"


def needs_build():
    """The shipped libocc_hip.so is trusted only when the digest recorded beside it equals the digest of the sources in the
    tree (the old mtime test compiled nothing whenever the .so travelled next to fresher-looking sources, so a
    driver-side build() proved little).  The stamp travels with the .so (both git-ignored, neither gpurun-ignored)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_digest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(bdir, os.path.splitext(s)[0] + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
               "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_digest() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
