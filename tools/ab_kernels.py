"""A/B timing of kernel variants selected by vhap_debug_set_flags: runs the bench workload eagerly, a few steps per flag value.
Run under `rocprofv3 --kernel-trace --output-format csv`, then `python tools/ab_kernels.py --report <trace.csv> f0,f1,...`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STEPS = 3

if "--report" in sys.argv:
    import collections, csv, re
    path, flags = sys.argv[2], sys.argv[3].split(",")
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "frame_prep_fwd" in r["Kernel_Name"]]
    starts = starts[-len(flags) * STEPS:] + [len(rows)]
    want = ("texture_bwd", "gbuffer_bwd", "aa_", "bin_", "shade_", "disturb_", "frame_prep", "tex_prep", "offset_reg", "landmark", "raster_kernel", "flame_skin_fwd",
            "mip_fold_tail", "adam_kernel", "shade_", "photo_", "texture_fwd", "aa_bwd")
    for fi, f in enumerate(flags):
        agg = collections.defaultdict(list)
        for s in range(1, STEPS):          # skip the first step of each phase
            a, b = starts[fi * STEPS + s], starts[fi * STEPS + s + 1]
            for r in rows[a:b]:
                n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
                if any(w in n for w in want):
                    agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("flags=%s  " % f + "  ".join("%s=%.0f" % (k[:22], sum(v) / len(v)) for k, v in sorted(agg.items())))
    sys.exit(0)

import torch
import bench
from vhap_amd import _lib
flags = [int(x) for x in sys.argv[1].split(",")]
tr, own, model, topo, gt = bench.build_tracker(0, 1, "cuda:0")
opt = tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.0)
sample = tr.get_sample(own, device_index=True)
for f in flags:
    _lib.debug_set_flags(f)
    for _ in range(STEPS):
        tr.optimize_iter(dict(sample), opt, bench.STAGE)
    torch.cuda.synchronize()
_lib.debug_set_flags(0)
