"""HBM traffic of ONE WHOLE optimiser step (stage rgb_global_tracking, BASELINE config 2), per kernel, from the memory-side PMC counters.
Run twice on the GPU box (the counters do not fit one pass, /opt/skills/guides/MI355X_MICROARCH.md):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/step_pmc.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/step_pmc.py
then   python tools/step_pmc.py --report gpurun_out/pmc_fetch gpurun_out/pmc_write  > profiles/rNN_step_pmc.json
MFMA utilisation of the FLAME contractions (north_star asks for it), a third pass over the same script:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python tools/step_pmc.py
    python tools/step_pmc.py --report-mfma gpurun_out/pmc_mfma > profiles/rNN_flame_mfma_pmc.json
The step is issued as the NativeStep's eager launch sequence (the same kernels the captured graph replays; counter collection serialises
kernels anyway), NREP times after a warm-up, between two marker launches; two calibration kernels with known byte counts (a 256 MiB fill and
a 256 MiB copy) give the scale of the raw counters."""
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NREP = 4
CAL_BYTES = 256 << 20

if "--report-mfma" in sys.argv:
    import csv
    d = sys.argv[sys.argv.index("--report-mfma") + 1]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "flame_" not in k and "verts_bwd_fused" not in k:
            continue
        e = acc.setdefault(k, {})
        e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {"note": "per launch (mean over the launches of the run).  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs); "
                   "GRBM_GUI_ACTIVE under counter collection includes the profiler's per-dispatch serialisation, so this is a LOWER bound -- "
                   "the FLAME contractions are skinny fp32 GEMMs (M = 16 frames, [16,436] x [436,15429]) bound by the 27 MB read of the "
                   "blendshape basis, not by the matrix cores: MFMA is used for its exact-fp32 16x16x4 tile"}
    for k, e in acc.items():
        m = {c: sum(v) / len(v) for c, v in e.items()}
        m["launches"] = max(len(v) for v in e.values())
        if m.get("GRBM_GUI_ACTIVE"):
            m["MfmaUtil_percent_lower_bound"] = 100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] * 1024)
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
            m["mfma_flops"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
        out[k] = m
    print(json.dumps(out, indent=1))
    sys.exit(0)

if "--report" in sys.argv:
    import csv
    import re

    def load(d, counter):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        return rows

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
        return n.split("(")[0][:60]

    def summarise(rows, cal_kernel):
        cal = [float(r["Counter_Value"]) for r in rows if cal_kernel in r["Kernel_Name"]]
        # the markers: vhap_copy_words_kernel launches of 1 KiB bracket the NREP measured steps
        idx = [i for i, r in enumerate(rows) if "vhap_copy_words_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) <= 4096]
        per = {}
        if len(idx) >= 2:
            lo, hi = idx[-2], idx[-1]               # marker (a 1 KiB copy), steps ..., marker, (calibration fill, calibration copy)
            for r in rows[lo + 1:hi]:
                per[short(r["Kernel_Name"])] = per.get(short(r["Kernel_Name"]), 0.0) + float(r["Counter_Value"]) / NREP
        return per, (max(cal) if cal else None)
    frows, wrows = load(sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], "WRITE_SIZE")
    fper, cal_r = summarise(frows, "vhap_copy_words_kernel")
    wper, cal_w = summarise(wrows, "vhap_zero_words_kernel")
    rs = CAL_BYTES / cal_r if cal_r else None
    ws = CAL_BYTES / cal_w if cal_w else None
    out = {"calibration": {"bytes": CAL_BYTES, "FETCH_SIZE_of_256MiB_copy": cal_r, "WRITE_SIZE_of_256MiB_fill": cal_w}, "per_kernel_MB_per_step": {}}
    tr_, tw_ = 0.0, 0.0
    for k in sorted(set(fper) | set(wper), key=lambda k: -(fper.get(k, 0) * (rs or 0) + wper.get(k, 0) * (ws or 0))):
        r = fper.get(k, 0.0) * rs if rs else None
        w = wper.get(k, 0.0) * ws if ws else None
        out["per_kernel_MB_per_step"][k] = {"read": round(r / 1e6, 2) if r is not None else None, "write": round(w / 1e6, 2) if w is not None else None}
        tr_ += r or 0.0
        tw_ += w or 0.0
    out["step_read_MB"], out["step_write_MB"], out["step_total_MB"] = round(tr_ / 1e6, 1), round(tw_ / 1e6, 1), round((tr_ + tw_) / 1e6, 1)
    out["note"] = ("HBM bytes per optimiser step (16 x 512^2, T = 2048, rgb_global_tracking: forward + backward + Adam), raw FETCH_SIZE / WRITE_SIZE "
                   "scaled by the in-run calibration kernels (256 MiB copy / fill), averaged over %d eagerly issued steps" % NREP)
    print(json.dumps(out, indent=1))
    sys.exit(0)

import ctypes

import torch
import bench
from vhap_amd import _lib
from vhap_amd.step import NativeStep
C = bench.CONFIGS[2]
tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
sample = tr.get_sample(own, device_index=True)
opt = tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.1)
for p in [p for g in opt.param_groups for p in g["params"]]:
    p.grad = torch.zeros_like(p)
opt._build()
opt.sync_lr()
ns = NativeStep(tr, sample, bench.STAGE)
# the step AS SHIPPED since round 6: the whole Adam update issued inside the step, the texture carried (the finish pass writes the next step's
# albedo / pyramid levels 1-2; no tex_prep_fwd at the head) -- VHAP_TEX_CARRY=0: the round-5 form
CARRY = os.environ.get("VHAP_TEX_CARRY", "1") != "0"
if CARRY:
    ns.step_optimizer = opt
    assert ns.enable_carry()
    ns.tex_prime()
ns.overlap = False                                          # one stream: the counters are per kernel, the order does not matter
L = ctypes.CDLL(_lib.SO_PATH)
L.vhap_debug_fill.restype = L.vhap_debug_copy.restype = ctypes.c_int
a = torch.empty(CAL_BYTES // 4, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
fill = lambda n: L.vhap_debug_fill(ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(0))
copy = lambda n: L.vhap_debug_copy(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(0))


def step():                                                 # the captured step's kernels, one chain: the texture's Adam update rides in its gradient pass
    ns.forward()
    ns.backward(1, optimizer=opt)
    if not CARRY:                                           # (carried: step_optimizer is set -- backward() issues the rest of the update itself)
        opt.step(skip=(tr.tex_extra,))


for _ in range(2):
    step()
torch.cuda.synchronize()
copy(1024)                                                  # marker
for _ in range(NREP):
    step()
copy(1024)                                                  # marker
fill(CAL_BYTES)
copy(CAL_BYTES)
torch.cuda.synchronize()
print("energy", float(ns.log[15]))
