"""Debugging aid for a ROCm 7 hipGraphLaunch crash (null dereference inside libamdhip64's GraphExec stream set-up): a graph with parallel
branches segfaults when the LAUNCH stream shares its hardware queue with one of the graph's internal parallel streams -- which depends on
how many streams the process created before.  This script captures one small step, then replays it on a series of freshly created
streams (kept alive), printing the index before each replay: the process dies at the first colliding stream.

    python tools/dbg/graph_stream_collision.py [normal|high|null] [n_streams]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "normal"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_flame_model, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker, GraphedStep
    H = W = 96
    N, T = 2, 128
    model, topo = make_flame_model(seed=0)
    cfg = BaseTrackingConfig()
    cfg.device = "cuda:0"
    cfg.model.tex_resolution = T
    gt = make_scene_params(N, seed=3, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (H, W), "cuda:0", seed=3, tex=make_texture(3, T))
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T), data)
    with torch.no_grad():
        tr.translation[:, 2] = 0.45
    stage = "rgb_global_tracking"
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
    st = GraphedStep(tr, tr.get_sample(np.arange(N), device_index=True), opt, stage)
    print("captured; single =", st.single, flush=True)
    keep = []
    for j in range(n):
        if mode == "high":
            s = torch.cuda.Stream(priority=-1)
        elif mode == "null":
            s = torch.cuda.default_stream()
        else:
            s = torch.cuda.Stream()
        keep.append(s)
        print(f"replay on stream {j} ({mode}) ...", end=" ", flush=True)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            st._replay()
        torch.cuda.synchronize()
        print("ok E =", float(st.E), flush=True)
    print("ALL OK", flush=True)


if __name__ == "__main__":
    main()
