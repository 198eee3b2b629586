"""Per-kernel timeline of the captured step from the plan executor's own timing events (no profiler attached).

    python tools/plan_timeline.py [--config 2] [--reps 9] [--out gpurun_out/plan_timeline.txt]

Prints the plan (stream layout, edges), then per node the median start (relative to the head of the replay) and duration over `reps`
timed replays (vhap_plan_launch_timed: every node bracketed by two events -- a few microseconds of overhead per node, so the sum is
longer than an untimed step; the untimed step time is measured separately and printed next to it).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plan_timeline.txt"))
    ap.add_argument("--two-pass-tex", action="store_true", help="A/B: texture gradient finish and its Adam update as two passes")
    ap.add_argument("--debug-flags", type=int, default=0, help="library debug flags (65536: plan edge events with the system-scope fence)")
    args = ap.parse_args()
    import bench
    from vhap_amd import step as vstep
    from vhap_amd.tracker import GraphedStep
    if args.two_pass_tex:
        vstep.FUSE_TEX_ADAM = False
    if args.debug_flags:
        from vhap_amd import _lib
        _lib.debug_set_flags(args.debug_flags)
    C = bench.CONFIGS[args.config]
    torch.cuda.set_device(0)
    tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
    opt = tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.1)
    sample = tr.get_sample(own, device_index=True)
    st = GraphedStep(tr, sample, opt, bench.STAGE)
    lines = [f"config {args.config}: {C['name']}", "---- plan ----", st.gF.describe()]
    with st.replay_stream():
        for _ in range(20):
            st()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        lines.append(f"untimed: {dt * 1e6:.1f} us per step ({C['B'] / dt:.0f} frames/s)")
        # host cost of one replay call (enqueue only)
        t0 = time.perf_counter()
        for _ in range(50):
            st()
        host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        lines.append(f"host enqueue time per replay (50 back-to-back calls, queue permitting): {host * 1e6:.1f} us")
        recs = []
        for _ in range(args.reps):
            recs.append(st.gF.timed())
            st.tr.global_step += 1
    names = [r[0].split("(")[0] for r in recs[0]]
    start = np.median(np.array([[x[1] for x in r] for r in recs]), axis=0)
    dur = np.median(np.array([[x[2] for x in r] for r in recs]), axis=0)
    lines.append("---- timed replays (median): start us, duration us, kernel ----")
    order = np.argsort(start)
    for i in order:
        lines.append(f"{start[i]:8.1f} {dur[i]:7.1f}  n{i:<3d} {names[i][:70]}")
    lines.append(f"sum of durations {dur.sum():.1f} us; last end {float((start + dur).max()):.1f} us")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
