"""Build helper for the RCCL test double (tests/fake_rccl/fake_rccl.cpp): TEST INFRASTRUCTURE, host-only C++ over the HIP runtime API."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "fake_rccl.cpp")
LIB = os.path.join(_HERE, "lib", "librccl.so.1")   # the SONAME the product's dlopen asks for


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC,
                           "-o", LIB, "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return LIB
