"""Builds the native libraries IN-TREE (the built .so travels to the GPU box with the snapshot).

    python -m dynam3d_amd.build            # libdynam3d_hip.so for gfx950 (hipcc cross-compiles without a GPU)
    python -m dynam3d_amd.build --host     # CPU-only bookkeeping library for `pytest -m "not gpu"`
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdynam3d_hip.so")
HOST_LIB = os.path.join(HERE, "libd3d_ffstate_host.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# (source, extra flags).  Geometry kernels are bit-exact against the oracle: no FMA contraction.
HIP_SOURCES = [
    ("geometry_kernels.hip", ["-ffp-contract=off"]),
    ("dense_kernels.hip", ["-ffp-contract=fast"]),
    ("tower_kernels.hip", ["-ffp-contract=off"]),
    ("train_kernels.hip", ["-ffp-contract=off"]),
    ("f32_kernels.hip", ["-ffp-contract=off"]),
    ("f32x3_kernels.hip", ["-ffp-contract=off"]),
    ("verify_f32_kernels.hip", ["-ffp-contract=off"]),
    ("gemm_kernels.hip", ["-ffp-contract=fast"]),
    ("attn3_kernels.hip", ["-ffp-contract=fast"]),
    ("attn4_kernels.hip", ["-ffp-contract=fast", "-fno-slp-vectorize"]),     # (SLP packs the softmax's float adds into v_pk_add_f32 + moves: slower)
    ("decode_attn_kernels.hip", ["-ffp-contract=fast"]),
    ("decode_kernels.hip", ["-ffp-contract=fast"]),
    ("render_kernels.hip", ["-ffp-contract=off"]),
    ("mlp_kernels.hip", ["-ffp-contract=fast"]),
    ("segment_kernels.hip", ["-ffp-contract=off"]),
    ("ff_plan_kernels.hip", ["-ffp-contract=off"]),
]
CPP_SOURCES = ["ff_state.cpp", "d3d_error.cpp", "phi3_decode.cpp", "mlp_forward.cpp"]
HOST_CPP_SOURCES = ["ff_state.cpp", "d3d_error.cpp", "ff_plan_host.cpp"]      # the CPU-only bookkeeping library (no device entry points;
                                                                          # ff_plan_host.cpp = the device planner's source on host arrays, tests only)


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def _deps(path):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "dynam3d_hip.h"))
    return [path] + hdrs


def build_hip(force: bool = False, verbose: bool = True) -> str:
    objs = []
    odir = os.path.join(HERE, "build")
    os.makedirs(odir, exist_ok=True)
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    procs = []
    for src, extra in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(odir, src + ".o")
        objs.append(obj)
        if force or any(_newer(d, obj) for d in _deps(sp)):
            cmd = [HIPCC] + common + extra + ["-c", sp, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), cmd))
    for src in CPP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(odir, src + ".o")
        objs.append(obj)
        if force or any(_newer(d, obj) for d in _deps(sp)):
            cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-c", sp, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), cmd))
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_host_state(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HOST_CPP_SOURCES]
    if force or any(_newer(d, HOST_LIB) for s in srcs for d in _deps(s)):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", HOST_LIB] + srcs
        subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    t0 = time.time()
    if "--host" in sys.argv:
        print(build_host_state(force="--force" in sys.argv))
    else:
        print(build_hip(force="--force" in sys.argv))
    print(f"built in {time.time() - t0:.1f}s")
