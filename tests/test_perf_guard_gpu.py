"""Performance guards ("tests never assert performance -- nothing fails if raster_kernel<1> regresses to 120 us").
Boxes differ by +-4 % and the suite shares a host with other jobs, so each guard takes the best of several measurements and sits just
under the worst box on record: isolated RI-fwd 0.36-0.41 of the peak over rounds 3-6 (guard 0.36: a 10 % slower kernel fails on every
box), the captured step 0.76-0.79 ms since the shading backward walks the list of covered pixels (guard 0.85 ms: a step that loses its
deferred join, its plan executor or the carried texture lands at 0.87-1.0 ms).  Round-5 review, weak 11: the first version (0.30 / 1.10 ms)
let a 20 % regression through."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench_state():
    spec = importlib.util.spec_from_file_location("bench_for_guard", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    C = bench.CONFIGS[2]
    tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
    opt = tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.1)
    sample = tr.get_sample(own, device_index=True)
    return bench, C, tr, opt, sample


def test_ri_fwd_pass_isolated_fraction_of_hbm_peak(bench_state):
    """bin_build + raster_kernel<1> (vhap_raster_interp_fwd) alone on the chip at 16 x 512^2: >= 0.36 of 8 TB/s against the 292 MB of
    SURVEY 8(d) (measured 0.36-0.41 over rounds 3-6)."""
    bench, C, tr, opt, sample = bench_state
    best = 0.0
    for _ in range(5):                                         # (the best of five: the guard is about the kernel, not about the neighbours)
        t, cov = bench.time_ri_isolated(tr, sample, C, torch.cuda.Stream())
        best = max(best, bench.ri_alg_bytes_per_frame(C["H"], C["W"]) * C["B"] / t / bench.HBM_PEAK)
    assert 0.2 < cov < 0.4
    assert best >= 0.36, f"RI-fwd isolated: {best:.3f} of the HBM peak"


def test_captured_step_time_at_the_quoted_config(bench_state):
    """One optimiser step (fwd + bwd + Adam, disturbance on) of BASELINE config 2 as bench.py times it: <= 0.85 ms (measured 0.76-0.79)."""
    import time
    from vhap_amd.tracker import GraphedStep
    bench, C, tr, opt, sample = bench_state
    step = GraphedStep(tr, sample, opt, bench.STAGE)
    assert step.ns is not None and step.gF.plan is not None and step.defer_join, step.defer_report
    best = 1e9
    with step.replay_stream():
        for _ in range(20):
            step()
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100)
    assert best <= 0.85e-3, f"{best * 1e3:.3f} ms per step"
