"""Build recipe for libvhap_hip.so (gfx950 only).  hipcc cross-compiles without a GPU.

    python -m vhap_amd.build [--force]

The shared library is written IN-TREE (vhap_amd/lib/libvhap_hip.so): it is git-ignored but
travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libvhap_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith((".h", ".hip")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdr_m = max(os.path.getmtime(os.path.join(d, f)) for d in (CSRC, INCLUDE) for f in os.listdir(d) if f.endswith(".h"))
    objs, rebuilt = [], False
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            cmd = [HIPCC, *FLAGS, "-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
            rebuilt = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if rebuilt or not os.path.exists(SO):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(SO)
