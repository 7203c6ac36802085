"""Build libbdetr.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

Usage: ``python -m boosted_detr_amd.build [--force]``.  hipcc cross-compiles without a
GPU, so this also runs in the authoring container.  The built ``.so`` is git-ignored but
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libbdetr.so"
OBJ_DIR = CSRC / "_obj"
SOURCES = ["common.cpp", "igemm.hip", "sgemm.hip", "hconv.hip", "hwgrad.hip", "p16.hip", "attention.hip", "rowchain.hip", "augment.hip", "norm.hip", "elementwise.hip", "panoptic.hip", "matcher.hip", "optim.hip"]
ARCH = "gfx950"
COMMON_FLAGS = ["-O3", "-fPIC", f"--offload-arch={ARCH}", "-std=c++20", "-Wall", "-Wno-unused-function"] + os.environ.get("BDETR_CXXFLAGS", "").split()
# the matcher must not contract a*b+c into fma (scipy / numpy evaluate unfused); see matcher.hip
PER_FILE_FLAGS = {"matcher.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.h", CSRC / "gemm_common.h", CSRC / "p16.h", CSRC.parent.parent / "include" / "bdetr.h"]
    stamp = CSRC / "_obj" / "stamp"
    dig = _digest(deps) + "|" + " ".join(COMMON_FLAGS)          # a build with other flags (BDETR_CXXFLAGS) is a different library
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    OBJ_DIR.mkdir(exist_ok=True)
    hipcc = _hipcc()

    headers = [d for d in deps if d.suffix == ".h"]

    def compile_one(src: str) -> Path:
        # one stamp per object: a change to one translation unit recompiles that unit only (a header change recompiles all)
        obj = OBJ_DIR / (src.rsplit(".", 1)[0] + ".o")
        flags = COMMON_FLAGS + PER_FILE_FLAGS.get(src, [])
        ostamp = OBJ_DIR / (src + ".stamp")
        odig = _digest([CSRC / src] + headers) + "|" + " ".join(flags)
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text() == odig:
            return obj
        cmd = [hipcc, *flags, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        ostamp.write_text(odig)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
