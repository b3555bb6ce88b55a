"""Build libidealnerf.so (gfx950) in-tree with hipcc.  No torch dependency.

    python ideal-nerf_amd/build.py [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libidealnerf.so")
SOURCES = ["mlp_f32.hip", "mlp_bf16x3.hip", "mlp_fp16x3.hip", "mlp_bf16x6.hip", "render_fused.hip", "mlp_f32_bwd.hip", "prep.hip", "composite.hip", "audio.hip", "train.hip", "capi.hip"]
# -ffp-contract=off: sample positions feed index decisions and must round like the
# reference's chain of eager ops (DESIGN.md "numerics"); MFMA is unaffected.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_diag(verbose=True, define="-DIDN_DIAG", tag="diag"):
    """Variant builds for diagnostics and same-box A/Bs: libidealnerf_<tag>.so compiled with an
    extra -D define (default: cycle stamps in the fp32 MLP kernel).  Never loaded by the package
    unless IDN_LIB points at them (tools/ab_bench.sh)."""
    obj = os.path.join(OBJ, tag)
    os.makedirs(obj, exist_ok=True)
    objs = []
    for src in SOURCES:
        o = os.path.join(obj, src.replace(".hip", ".o"))
        cmd = [hipcc()] + FLAGS + define.split() + ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(o)
    lib = os.path.join(HERE, f"libidealnerf_{tag}.so")
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def build_variant(tag, define, sources, verbose=True):
    """libidealnerf_<tag>.so for same-box A/Bs (tools/ab_bench.sh): only `sources` are recompiled with the
    extra defines, every other object comes from the regular build (run build() first)."""
    obj = os.path.join(OBJ, tag)
    os.makedirs(obj, exist_ok=True)

    def one(src):
        o = os.path.join(obj, src.replace(".hip", ".o"))
        cmd = [hipcc()] + FLAGS + define.split() + ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return o

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        changed = dict(zip(sources, ex.map(one, sources)))
    objs = [changed.get(s, os.path.join(OBJ, s.replace(".hip", ".o"))) for s in SOURCES]
    lib = os.path.join(HERE, f"libidealnerf_{tag}.so")
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "idealnerf.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc()] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag())
    elif "--variant" in sys.argv:   # --variant <tag> "<-D...>" src.hip [src.hip ...]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3:]))
    else:
        print(build(force="--force" in sys.argv))
