"""Build librpg_ramnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every csrc/*.hip is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked
into one shared library with a plain C ABI (include/ramnet_hip.h) and no torch dependency."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
HEADER = os.path.join(PKG, "..", "include", "ramnet_hip.h")
LIB = os.path.join(PKG, "librpg_ramnet_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# per-file additions.  conv_wino6.hip: its main loop places single scalar fp32 operations between MFMAs by hand; the SLP vectoriser
# pairs neighbours into 2-wide vectors that the backend scalarises again through v_mov shuffles (4 moves per 2 v_fma)
# (the SLP vectorizer pairs the fp32 transform arithmetic of these kernels into <2 x float> values that the backend takes apart again with
# v_mov shuffles: 29 of 303 instructions per 32 MFMAs in the backward-weights loop)
# conv_wino.hip: 24 v_mov per chunk pair in the dense instantiations (181 instead of 196 instructions per 32 MFMAs at 32 channels), +1 % in the sparse ones
EXTRA_FLAGS = {"conv_wino6.hip": ["-fno-slp-vectorize"], "conv_wgrad_wino.hip": ["-fno-slp-vectorize"], "conv_wino.hip": ["-fno-slp-vectorize"],
               "conv_wgrad_wino6.hip": ["-fno-slp-vectorize"], "conv_wino6s.hip": ["-fno-slp-vectorize"],
               "conv_wgrad_dsplit.hip": ["-fno-slp-vectorize"]}          # (its split arithmetic: 576 instead of 595 instructions per batch)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [HEADER]
    return max(os.path.getmtime(h) for h in hs)


def _compile(hipcc, src, obj, verbose):
    tmp = "%s.%d.tmp" % (obj, os.getpid())        # concurrent build() calls must not see each other's half-written objects
    cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", tmp]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode == 0:
        os.replace(tmp, obj)
    elif os.path.exists(tmp):
        os.remove(tmp)
    return src, r


def build(force=False, verbose=False):
    """Compile every HIP source and link the shared library; returns its path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hm = _headers_mtime()
    jobs, objs = [], []
    for f in sources():
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            jobs.append((src, obj))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for src, r in ex.map(lambda j: _compile(hipcc, j[0], j[1], verbose), jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed on %s" % src)
                if verbose:
                    sys.stderr.write(r.stderr)
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed linking %s" % LIB)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv))
