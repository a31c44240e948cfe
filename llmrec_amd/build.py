"""Builds libllmrec_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m llmrec_amd.build [--force]

hipcc cross-compiles without a GPU. The .so is git-ignored but travels to the GPU box with the
repo snapshot; ``llmrec_amd._lib`` refuses to run without it (there is no CPU fallback)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libllmrec_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")
SOURCES = ["graph.hip", "spmm.hip", "dense.hip", "rowops.hip", "bpr.hip", "topk.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build llmrec_amd/lib/libllmrec_hip.so)")


def _deps_mtime() -> float:
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    files.append(os.path.join(os.path.dirname(HERE), "include", "llmrec_hip.h"))
    return max(os.path.getmtime(f) for f in files)


def build(force: bool = False, verbose: bool = True, tools: bool = False) -> str:
    """tools=True: the instrumented variant for tools/ (-DLLMREC_TOOLS_BUILD: cycle buckets, ablation launches) as
    lib/libllmrec_hip_tools.so; LLMREC_LIB=<path> makes llmrec_amd._lib load it instead of the product library."""
    lib = LIB.replace(".so", "_tools.so") if tools else LIB
    objdir = OBJDIR + ("_tools" if tools else "")
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _deps_mtime():
        return lib
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + (["-DLLMREC_TOOLS_BUILD"] if tools else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = lib + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, lib)
    if verbose:
        print("[llmrec_amd.build] built", lib)
    return lib


HOST_LIB = os.path.join(LIBDIR, "libllmrec_host.so")


def build_host(force: bool = False, verbose: bool = True) -> str:
    """libllmrec_host.so: the plain-C host helper of the drop-in's Data.sample() (csrc/host_sampler.c, gcc). Optional: without it the
    same draws run as a Python loop."""
    src = os.path.join(CSRC, "host_sampler.c")
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= os.path.getmtime(src):
        return HOST_LIB
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for llmrec_amd/csrc/host_sampler.c")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = HOST_LIB + ".tmp"
    r = subprocess.run([cc, "-O2", "-shared", "-fPIC", "-std=c99", "-Wall", "-o", tmp, src], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed on host_sampler.c:\n" + r.stderr)
    os.replace(tmp, HOST_LIB)
    if verbose:
        print("[llmrec_amd.build] built", HOST_LIB)
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tools="--tools" in sys.argv)
    build_host(force="--force" in sys.argv)
