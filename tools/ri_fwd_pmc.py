"""HBM traffic of the fused rasterize+interpolate forward pass from the memory-side PMC counters.
Run twice on the GPU box (the counters do not fit one pass, /opt/skills/guides/MI355X_MICROARCH.md):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/ri_fwd_pmc.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/ri_fwd_pmc.py
then   python tools/ri_fwd_pmc.py --report gpurun_out/pmc_fetch gpurun_out/pmc_write  > profiles/r01_ri_fwd_pmc.json
The run also launches two calibration kernels with known byte counts (a 256 MiB fill and a 256 MiB copy)."""
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NREP = 5
CAL_BYTES = 256 << 20

if "--report" in sys.argv:
    import csv

    def load(d, counter):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        return per
    fetch, write = load(sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], "WRITE_SIZE")
    pick = lambda per, key: [v for k, vs in per.items() if key in k for v in vs]
    # calibration: the fill writes CAL_BYTES, the copy reads and writes CAL_BYTES (counters are in KiB-like units; derive the scale)
    cal_w = max(pick(write, "vhap_zero_words_kernel")) if pick(write, "vhap_zero_words_kernel") else None
    cal_r = max(pick(fetch, "vhap_copy_words_kernel")) if pick(fetch, "vhap_copy_words_kernel") else None
    out = {"calibration": {"bytes": CAL_BYTES, "WRITE_SIZE_of_256MiB_fill": cal_w, "FETCH_SIZE_of_256MiB_copy": cal_r}}
    ws = CAL_BYTES / cal_w if cal_w else None
    rs = CAL_BYTES / cal_r if cal_r else None
    tot = 0.0
    for name in ("bin_build_kernel", "raster_kernel<1>", "raster_kernel<2>"):
        f, w = pick(fetch, name), pick(write, name)
        fm, wm = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)
        out[name] = {"FETCH_SIZE_raw": fm, "WRITE_SIZE_raw": wm, "read_bytes": fm * rs if rs else None, "write_bytes": wm * ws if ws else None,
                     "launches": len(f)}
        if rs and ws and name != "raster_kernel<2>":
            tot += fm * rs + wm * ws
    out["traffic_bytes_per_launch"] = tot if tot else None         # the RI-fwd pass: bin_build + raster_kernel<1>
    if rs and ws and out["raster_kernel<2>"]["launches"]:
        d2 = out["raster_kernel<2>"]
        out["traffic_bytes_per_launch_deferred"] = out["bin_build_kernel"]["read_bytes"] + out["bin_build_kernel"]["write_bytes"] + d2["read_bytes"] + d2["write_bytes"]
    out["note"] = ("raw counters scaled by the in-run calibration kernels (a 256 MiB fill for WRITE_SIZE, a 256 MiB coalesced copy for FETCH_SIZE), "
                   "as the guide prescribes for gfx950 where the absolute units are uncalibrated")
    print(json.dumps(out, indent=1))
    sys.exit(0)

import ctypes

import numpy as np
import torch
import bench
from vhap_amd import _lib, ops
from vhap_amd.ops import _p, _stream
from vhap_amd.step import NativeStep
C = bench.CONFIGS[2]
H, W = C["H"], C["W"]
tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
sample = tr.get_sample(own, device_index=True)
tr.get_train_parameters(bench.STAGE)
with torch.no_grad():
    s = dict(sample)
    tr.fill_cam_params_into_sample(s)
    verts, *_ = tr.forward_flame(s["timestep_index"])
    rd = tr.render.rasterize(verts, tr.flame.faces, s["extrinsic"], s["intrinsic"], (H, W), defer=True)
    vn = tr.render.compute_v_normals(verts, tr.flame.faces)
    tri, tri_uv = tr.render._tri32(tr.flame.faces), tr.render._tri32(tr.flame.textures_idx)
    pos = rd["verts_clip"].contiguous()
    for _ in range(NREP):                                   # the RI-fwd op: bin_build + raster_kernel<1>
        ops.raster_interp_fwd(tr.render.glctx, pos, tri, vn, tr._verts_uv_flipped, tri_uv, (H, W))
    # the deferred-shading pass the captured step runs: bin_build + raster_kernel<2> (rasterise + interpolate + texture + shade + composite)
    ns = NativeStep(tr, sample, bench.STAGE)
    ns.overlap = False
    for _ in range(NREP):
        ns.forward()
    # calibration through the library's own fill / copy kernels
    a = torch.empty(CAL_BYTES // 4, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    L = ctypes.CDLL(_lib.SO_PATH)
    L.vhap_debug_fill.restype = ctypes.c_int
    L.vhap_debug_fill(ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(CAL_BYTES), ctypes.c_void_p(0))
    L.vhap_debug_copy(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(CAL_BYTES), ctypes.c_void_p(0))
    torch.cuda.synchronize()
