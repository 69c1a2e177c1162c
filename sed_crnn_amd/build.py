"""Build libsedcrnn.so (hand-written HIP, gfx950 only) in-tree with hipcc.

`python -m sed_crnn_amd.build` or `sed_crnn_amd.build.build()`.  hipcc cross-compiles without a GPU;
the .so is git-ignored but travels to the GPU box with the working tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsedcrnn.so")
SOURCES = ["api.cpp", "conv.hip", "conv1.hip", "bnpool.hip", "gemm.hip", "gru.hip", "misc.hip", "logmel.hip", "data.hip", "net.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extras.  logmel: the SLP vectoriser packs the FFT's scalar adds into v_pk_add_f32 and pays for it with ~600
# v_mov per frame pair to build the register pairs (packed f32 is no faster than scalar on gfx950)
EXTRA_FLAGS = {"logmel.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "sedcrnn.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
