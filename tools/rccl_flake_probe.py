"""Why does tests/test_dist_gpu.py::test_one_rank_rccl_sharded_step_matches_unsharded[False-False] miss its bound in one run out of three?

The same scene and the same three steps as the test, repeated `--trials` times, with a snapshot of every gradient and every parameter after EACH step
(the device is synchronised for the snapshot: `--no-sync-each` keeps the test's own timing and snapshots the last step only).  Per trial and step:
how far the parameters the step started from were apart, how far the gradients are apart, and HOW MANY elements of the per-vertex gradient
(static_offset) are off -- a handful of vertices is a pixel on the other side of a kink (L1 sign, coverage), thousands are a race.

    python tools/rccl_flake_probe.py --trials 8 [--tex-sharded 0] [--tex-first 0] [--no-sync-each]"""
import argparse
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_dist_gpu import NAMES, _build  # noqa: E402


def _snap(tr):
    return ({k: getattr(tr, k).grad.detach().cpu().clone() for k in NAMES if getattr(tr, k).grad is not None},
            {k: getattr(tr, k).detach().cpu().clone() for k in NAMES})


def _run(tr, n_steps, sync_each):
    from vhap_amd.tracker import GraphedStep
    opt = tr.configure_optimizer(tr.get_train_parameters("rgb_global_tracking"), lr_scale=0.1)
    st = GraphedStep(tr, tr.get_sample(np.arange(4), device_index=True), opt, "rgb_global_tracking", warmup=0)
    E, snaps = [], []
    with st.replay_stream():
        for _ in range(n_steps):
            E.append(float(st()))
            if sync_each:
                st.join()
                torch.cuda.synchronize()
                snaps.append(_snap(tr))
    torch.cuda.synchronize()
    if not sync_each:
        snaps.append(_snap(tr))
    return E, snaps, st


def _worker(rank, world, port, T, ret, n_steps, tex_sharded, tex_first, sync_each):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VHAP_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      VHAP_TEX_SHARDED="1" if tex_sharded else "0", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                      VHAP_SHARD_TEX_FIRST="1" if tex_first else "0", VHAP_TEX_KEEP_GRAD="1")
    from vhap_amd import dist as vdist
    vdist.init_from_env("nccl")
    tr = _build(T)
    vdist.attach(tr)
    E, snaps, st = _run(tr, n_steps, sync_each)
    assert not st.single
    ret[0] = (E, snaps)
    dist.destroy_process_group()


def _cmp(a, b, floor=0.0):
    nrm = float(b.abs().max())
    if nrm == 0:
        return 0.0, 0, []
    d = ((a - b).abs() / nrm).reshape(-1)
    off = torch.nonzero(d > max(10 * floor, 1e-5)).reshape(-1)
    return float(d.max()), int(off.numel()), off[:8].tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=6)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--tex-sharded", type=int, default=0)
    ap.add_argument("--tex-first", type=int, default=0)
    ap.add_argument("--no-sync-each", action="store_true")
    a = ap.parse_args()
    os.environ["VHAP_TEX_KEEP_GRAD"] = "1"
    T, sync_each = 128, not a.no_sync_each
    start = _build(T)
    p0 = {k: getattr(start, k).detach().cpu().clone() for k in NAMES}
    print(f"tex_sharded={a.tex_sharded} tex_first={a.tex_first} sync_each={sync_each}", flush=True)
    for t in range(a.trials):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(1, port, T, ret, a.steps, bool(a.tex_sharded), bool(a.tex_first), sync_each), nprocs=1, join=True)
        E_s, S_s = ret[0]
        E_1, S_1, _ = _run(_build(T), a.steps, sync_each)
        E_2, S_2, _ = _run(_build(T), a.steps, sync_each)
        print(f"--- trial {t}: energies sharded / one-plan / one-plan again: " +
              "  ".join(f"{abs(x - y) / abs(y):.1e}|{abs(z - y) / abs(y):.1e}" for x, y, z in zip(E_s, E_1, E_2)), flush=True)
        for i, ((g_s, p_s), (g_1, p_1), (g_2, p_2)) in enumerate(zip(S_s, S_1, S_2)):
            worst = []
            for k in NAMES:
                if k not in g_1:
                    continue
                fl, _, _ = _cmp(g_2[k], g_1[k])
                rel, n_off, first = _cmp(g_s[k], g_1[k], fl)
                upd = float((p_1[k] - p0[k]).abs().max())
                dp = float((p_s[k] - p_1[k]).abs().max()) / upd if upd else 0.0
                dp2 = float((p_2[k] - p_1[k]).abs().max()) / upd if upd else 0.0
                flag = "  <-- OFF" if rel > max(3 * fl, 2e-5) else ""
                worst.append(f"    {k:14s} grad {rel:.1e} (self {fl:.1e})  off {n_off:6d}/{g_1[k].numel():7d} {first if n_off else ''}   "
                             f"params after: {dp:.1e} of the update (self {dp2:.1e}){flag}")
            print(f"  step {i if sync_each else a.steps - 1}:", flush=True)
            print("\n".join(worst), flush=True)


if __name__ == "__main__":
    main()
