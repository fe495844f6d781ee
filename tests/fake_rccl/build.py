"""Builds tests/fake_rccl/libmgf_fake_rccl.so (test infrastructure: see fake_rccl.cpp's header)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.cpp")
OUT = os.path.join(HERE, "libmgf_fake_rccl.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", SRC, "-o", OUT,
           f"-L{rocm}/lib", "-lamdhip64", "-lrt", "-lpthread", f"-Wl,-rpath,{rocm}/lib"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
