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
SOURCES = ["api.cpp", "conv.hip", "wino.hip", "conv1.hip", "bnpool.hip", "gemm.hip", "gru.hip", "misc.hip", "logmel.hip", "data.hip", "net.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extras.  logmel: the SLP vectoriser packs the FFT's scalar adds into v_pk_add_f32 and pays for it with ~600
# v_mov per frame pair to build the register pairs (packed f32 is no faster than scalar on gfx950)
EXTRA_FLAGS = {"logmel.hip": ["-fno-slp-vectorize"],
               "conv.hip": ["-Rpass-analysis=kernel-resource-usage"],
               "wino.hip": ["-Rpass-analysis=kernel-resource-usage"]}
# Kernels whose inline-asm loads are consumed after HAND-COUNTED s_waitcnt vmcnt(N) immediates (conv.hip: the fp32 and the
# bf16x3 conv forward).  The counts are only right while hipcc keeps the load destinations in registers between the asm load
# and its use: a spill (scratch store/reload) would insert memory operations the counts do not know about and the MFMAs
# could read stale fragments without any test necessarily noticing.  The build therefore FAILS if one of them spills — or uses
# AGPRs at all: under register pressure hipcc parks VGPR values in AGPRs with v_accvgpr_write right after the instruction
# that defined them, which for an asm load is BEFORE its data arrives (stale copy; the returning load then overwrites a
# register that has been given to something else — seen as a GPU memory fault in a round-3 experiment kernel).
NO_SPILL_KERNELS = {"conv.hip": ("conv3x3_mfma_fwd2_k", "conv3x3_mfma_fwd_bf16x3_k")}
# Kernels that hold their accumulators in AGPRs by design (one wave per SIMD, 256 accumulator registers) and leave all waits to
# hipcc: a spill there is not a correctness hazard, it is a performance cliff nobody would notice — scratch must stay 0.
NO_SCRATCH_KERNELS = {"wino.hip": ("conv3x3_wino_k",)}


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


def check_no_spill(remarks, kernels, allow_agprs=False):
    """parse -Rpass-analysis=kernel-resource-usage output; returns the offending lines (empty = fine).  Every guarded kernel
    must appear at least once, so a rename cannot silently disable the guard."""
    import re
    bad, seen, cur = [], set(), None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = next((k for k in kernels if k in m.group(1)), None)
            if cur:
                seen.add(cur)
            continue
        if cur:
            m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|AGPRs): (\d+)", line)
            if m and int(m.group(2)) != 0 and not m.group(1).startswith("SGPRs") and not (allow_agprs and m.group(1) == "AGPRs"):
                bad.append(f"{cur}: {m.group(1)} = {m.group(2)}")
    bad += [f"{k}: no resource-usage remark found (kernel renamed? update NO_SPILL_KERNELS)" for k in kernels if k not in seen]
    return bad


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_shared.h"), os.path.join(HERE, "..", "include", "sedcrnn.h")]
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
        src = next((a for a in cmd if a.endswith((".hip", ".cpp"))), "")
        guarded = NO_SPILL_KERNELS.get(os.path.basename(src)) or NO_SCRATCH_KERNELS.get(os.path.basename(src))
        if guarded:
            bad = check_no_spill(r.stderr, guarded, allow_agprs=os.path.basename(src) in NO_SCRATCH_KERNELS)
            if bad:
                os.remove(cmd[cmd.index("-o") + 1])
                raise RuntimeError("hand-counted-waitcnt kernels must not spill:\n" + "\n".join(bad))
            # the resource remarks are not warnings: drop them (and their source-context lines), keep genuine diagnostics
            keep, skip = [], 0
            for line in r.stderr.splitlines():
                if "remark:" in line:
                    skip = 2                                   # a remark is followed by the quoted source line and a caret line
                    continue
                if skip and (line.strip().startswith(("|", "^")) or " | " in line[:12]):
                    skip -= 1
                    continue
                skip = 0
                keep.append(line)
            return "\n".join(l for l in keep if l.strip() and "remarks generated" not in l)
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
