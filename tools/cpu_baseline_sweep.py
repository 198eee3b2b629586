#!/usr/bin/env python3
"""How many host threads give the CPU restatement its best step time?  (bench.py used 16 of the box's 256 on an
unmeasured claim.)  One warm-up + two timed whole steps of bench.py's cpu_baseline per thread count, on the GPU box's host:

    python tools/cpu_baseline_sweep.py [--threads 16 64 256] > profiles/r05_cpu_baseline_thread_sweep.txt
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[16, 64, 256])
    ap.add_argument("--config", type=int, default=2)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    C = bench.CONFIGS[a.config]
    tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
    tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.1)
    sample = tr.get_sample(own, device_index=True)
    print(f"host: {os.cpu_count()} hardware threads, load average {os.getloadavg()}; BASELINE config {a.config}: {C['name']}")
    best = None
    for n in a.threads:
        r = bench.cpu_baseline(C, tr, sample, model, topo, n_timed=2, budget_s=200.0, cores=n)
        print(f"threads {r['cores']:4d}: {r['value']:.3f} frames/s   ({r['sample'][:70]}...)", flush=True)
        if best is None or r["value"] > best[1]:
            best = (r["cores"], r["value"])
    print(f"best: {best[0]} threads, {best[1]:.3f} frames/s")


if __name__ == "__main__":
    main()
