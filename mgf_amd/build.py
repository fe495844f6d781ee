"""Build the HIP extension in-tree: mgf_amd/libmgf_hip.so (gfx950, no torch dependency).

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps every f32 operation un-fused so
the kernels reproduce the reference's IEEE sequence bit for bit.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmgf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP pass packs the solver's scalar f32 arithmetic into v_pk_* pairs glued together with v_mov shuffles; the
# serving loop of k_solve_flow6 is one lane's dependent chain, where the plain form is 2-5 % faster (profiles/r04_noslp.txt)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
UNITS = ["prims.hip", "mgf_hip.hip"]


def _deps(unit):
    srcs = [os.path.join(CSRC, unit)]
    srcs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    srcs.append(os.path.join(HERE, "..", "include", "mgf_hip.h"))
    return srcs


def _stale(target, srcs):
    return (not os.path.exists(target)) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in srcs)


def _compile(unit, force):
    obj = os.path.join(CSRC, unit.replace(".hip", ".o"))
    deps = _deps(unit) if unit != "prims.hip" else [os.path.join(CSRC, "prims.hip"), os.path.join(CSRC, "common.h")]
    if force or _stale(obj, deps):
        subprocess.check_call([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, unit), "-o", obj])
    return obj


def build(force=False, verbose=False):
    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(lambda u: _compile(u, force), UNITS))
    if force or _stale(LIB, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
