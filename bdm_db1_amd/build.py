"""Builds libdb1_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

The built library stays next to this file (git-ignored, but it travels to the GPU box).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdb1_hip.so")
OBJ = os.path.join(CSRC, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("DB1_EXTRA_HIPCC_FLAGS", "").split()   # experiments only (tools/exp): e.g. -DFW2_STOP=2; part of the build digest


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    for hdr in ("db1_hip.h", "db1_hip_test.h"):
        h.update(open(os.path.join(HERE, "..", "include", hdr), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_lib(force: bool = False, verbose: bool = False) -> str:
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, src[:-4] + ".o")
        cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        if verbose:
            print("compiled", src, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


DATA_SRC = os.path.join(HERE, "csrc_host", "db1_data.cpp")
DATA_LIB = os.path.join(HERE, "libdb1_data.so")


def build_data_lib(force: bool = False) -> str:
    """libdb1_data.so: the host-side data ingest (include/db1_data.h) -- plain C++17, g++."""
    hdr = os.path.join(os.path.dirname(HERE), "include", "db1_data.h")
    newest = max(os.path.getmtime(DATA_SRC), os.path.getmtime(hdr))
    if not force and os.path.exists(DATA_LIB) and os.path.getmtime(DATA_LIB) >= newest:
        return DATA_LIB
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", DATA_LIB, DATA_SRC], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on db1_data.cpp:\n{r.stderr[-4000:]}")
    return DATA_LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(build_data_lib(force="--force" in sys.argv))
