"""Build liblzgpu.so (hand-written HIP for gfx950) in-tree: python -m lastz_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("LZGPU_CSRC") or os.path.join(HERE, "csrc")      # (LZGPU_CSRC: a patched copy of the sources, tools/build_variant.sh)
OUT = os.path.join(HERE, "liblzgpu.so")
SOURCES = ["lzgpu_api.hip", "seed_kernels.hip", "dp_kernels.hip", "dp_kernels_narrow.hip", "window_kernels.hip", "lz_share.hip", "lz_host.cpp", "lz_gapped_host.cpp", "lz_dp_pieces.cpp", "lz_chain_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"] + os.environ.get("LZGPU_CXXFLAGS", "").split()


def _stale(obj, srcs):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = os.environ.get("LZGPU_BUILD_TAG", "")          # variant builds (profiling switches): liblzgpu_<tag>.so, picked up with LZGPU_LIB
    out = OUT if not tag else os.path.join(HERE, "liblzgpu_%s.so" % tag)
    objdir = os.path.join(HERE, "build" if not tag else "build_" + tag)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "lzgpu.h"))
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            flags = FLAGS if src.endswith(".hip") else ["-x", "c++"] + [f for f in FLAGS if not f.startswith("--offload")]   # host-only sources: no device pass
            cmd = [hipcc] + flags + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
