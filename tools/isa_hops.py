"""Dependent round trips per kernel, read off the ISA: for every kernel of the given csrc/*.hip files, the number of load GROUPS -- runs of
global / buffer / flat loads with no `s_waitcnt vmcnt` between them -- and the number of loads.  A kernel whose groups ~ its loads issues
load -> wait -> use one at a time (an optional input behind its own `if`, a run-time trip count, a table of the kernel-argument segment
indexed per lane, a `#pragma unroll` that did not unroll); a latency-bound kernel pays a round trip per group (the two
"round 4, late" lessons).  Static counts: loops and branches not taken are not weighed.

    python tools/isa_hops.py [frame.hip flame.hip ...]        (default: every vhap_amd/csrc/*.hip; hipcc cross-compiles without a GPU)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def isa(src, out_dir):
    s = os.path.join(out_dir, os.path.basename(src)[:-4] + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", f"-I{ROOT}/include", f"-I{ROOT}/vhap_amd/csrc", "-S",
                    "--cuda-device-only", src, "-o", s], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return s


def hops(asm_path):
    """{mangled kernel name: (load groups, loads)}"""
    out, name, groups, pending, loads = {}, None, 0, 0, 0
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, groups, pending, loads = m.group(1), 0, 0, 0
            continue
        if name is None:
            continue
        if re.search(r"\b(global|buffer|flat)_load", line):
            pending += 1
            loads += 1
        elif "s_waitcnt" in line and "vmcnt" in line:
            if pending:
                groups += 1
            pending = 0
        elif ".end_amdhsa_kernel" in line:
            out[name] = (groups, loads)
            name = None
    return out


def demangle(names):
    r = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.split("\n")
    return [n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for n in r]


def table(files):
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(max_workers=8) as ex:
        res = {}
        for asm in ex.map(lambda f: isa(f, d), files):
            h = hops(asm)
            for k, n in zip(h, demangle(h)):
                res[n] = h[k]
    return res


def main():
    files = [os.path.join(ROOT, "vhap_amd", "csrc", a) if not os.path.isabs(a) else a for a in sys.argv[1:]] or \
        sorted(glob.glob(os.path.join(ROOT, "vhap_amd", "csrc", "*.hip")))
    res = table(files)
    print("load groups  loads  kernel")
    for n, (g, l) in sorted(res.items(), key=lambda kv: -kv[1][0]):
        print(f"{g:11d}  {l:5d}  {n}")


if __name__ == "__main__":
    main()
