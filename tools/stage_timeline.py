"""Where does a step of the stage loop go?  GlobalTracker.optimize_stage('rgb_global_tracking') over shuffled batches of a resident
FrameStore (what bench.py's stage_fps times), with the host side of every step timed piece by piece and the GPU side by events per epoch.

    python tools/stage_timeline.py [--frames 256] [--epochs 5] [--out FILE]
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
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from vhap_amd import tracker as T
    from vhap_amd.ingest import FrameStore
    from vhap_amd.tracker import GlobalTracker, GraphedStep, ShuffledBatches
    C = bench.CONFIGS[2]
    tr0, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
    B, n_frames = C["B"], args.frames
    reps = (n_frames + B - 1) // B
    rgb = tr0.dataset["rgb"][:B]
    u8 = (rgb.permute(0, 2, 3, 1).clamp(0, 1) * 255).round().to(torch.uint8).repeat(reps, 1, 1, 1)[:n_frames].contiguous()
    data = {"frames": FrameStore(u8, device="cuda:0"), "lmk2d": tr0.dataset["lmk2d"][:B].repeat(reps, 1, 1)[:n_frames].contiguous()}
    tr = GlobalTracker(tr0.cfg, model, topo, tr0.flame_tex_painted()[0].cpu().numpy(), data)
    with torch.no_grad():
        for name in ("shape", "lights", "tex_extra", "static_offset", "focal_length"):
            getattr(tr, name).copy_(getattr(tr0, name))
        for name in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
            p, q = getattr(tr, name), getattr(tr0, name)[:B]
            p.copy_(q.repeat(reps, *([1] * (q.dim() - 1)))[:n_frames])
    del tr0
    stage = bench.STAGE
    cfg = tr.cfg
    loader = ShuffledBatches(tr, B, device_index=True, generator=torch.Generator().manual_seed(0))
    keep = cfg.pipeline[stage].num_epochs
    # ---- host-side instrumentation: wrap the pieces of the per-step work ----
    acc = {}

    def wrap(obj, name, key):
        fn = getattr(obj, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
            acc[key + "#"] = acc.get(key + "#", 0) + 1
            return r
        setattr(obj, name, w)
    wrap(GlobalTracker, "get_train_parameters", "get_train_parameters")
    for nm in ("update_timesteps", "feed_epoch", "_replay", "join"):
        if hasattr(GraphedStep, nm):
            wrap(GraphedStep, nm, nm)
    wrap(T.NV.HipAdam, "sync_lr", "sync_lr")
    made = []
    init0 = GraphedStep.__init__

    def init1(self, *a, **k):
        init0(self, *a, **k)
        made.append(self)
    GraphedStep.__init__ = init1
    cfg.pipeline[stage].num_epochs = 1
    tr.optimize_stage(stage, dataloader=loader, lr_scale=0.1)           # capture + warm-up pass
    torch.cuda.synchronize()
    acc.clear()
    cfg.pipeline[stage].num_epochs = args.epochs
    import gc
    gcs = {"t": 0.0, "n": [0, 0, 0], "t0": 0.0}

    def gc_cb(phase, info):                                   # time spent in the cyclic collector during the timed stage
        if phase == "start":
            gcs["t0"] = time.perf_counter()
        else:
            gcs["t"] += time.perf_counter() - gcs["t0"]
            gcs["n"][info["generation"]] += 1
    gc.callbacks.append(gc_cb)
    if os.environ.get("STAGE_GC") == "0":
        gc.disable()
    t0 = time.perf_counter()
    tr.optimize_stage(stage, dataloader=loader, lr_scale=0.1)
    t_host = time.perf_counter() - t0
    gc.enable()
    gc.callbacks.remove(gc_cb)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    cfg.pipeline[stage].num_epochs = keep
    steps = args.epochs * len(loader)
    lines = [f"optimize_stage({stage}): {n_frames} frames, batches of {B}, {args.epochs} epochs = {steps} steps",
             f"wall {t_all * 1e3:.2f} ms = {t_all / steps * 1e3:.4f} ms/step = {n_frames * args.epochs / t_all:.0f} frames/s; "
             f"host loop returned after {t_host * 1e3:.2f} ms ({t_host / steps * 1e3:.4f} ms/step of host time)"]
    lines.append(f"  cyclic collector during the timed stage: {gcs['t'] * 1e3:.2f} ms in {gcs['n']} collections (generation 0 / 1 / 2); {len(gc.get_objects())} tracked objects"
                 + ("  [STAGE_GC=0: collector off]" if os.environ.get("STAGE_GC") == "0" else ""))
    for st in made:
        rep = getattr(st, "defer_report", None) or []
        lines.append(f"  captured step: deferred join {'ON' if getattr(st, 'defer_join', False) else 'OFF'}, self-feeding {'ON' if getattr(st, 'feed', None) is not None else 'OFF'}"
                     + (f"; {str(rep[-1])[:160]}" if rep else ""))
    for k in sorted(k for k in acc if not k.endswith("#")):
        lines.append(f"  host time in {k}: {acc[k] * 1e3:.2f} ms total, {acc[k] / max(acc[k + '#'], 1) * 1e6:.1f} us per call x {acc[k + '#']}")
    rest = t_host - sum(v for k, v in acc.items() if not k.endswith("#") and k not in ("sync_lr",))
    lines.append(f"  rest of the host loop (Python, batch index bookkeeping, uploads): {rest * 1e3:.2f} ms total, {rest / steps * 1e6:.1f} us per step")
    print("\n".join(lines))
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
