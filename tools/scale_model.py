#!/usr/bin/env python3
"""What the frame-sharded step should cost on N MI355X of one node -- stated BEFORE it could be measured (no multi-GPU node was available to
this build in six rounds).  A critical-path model of one steady-state period of the sharded step from pieces measured on
ONE GPU (a world-size-1 RCCL group runs the sharded step for real: every overhead but wire time).

    measured, one GPU, 16 x 512^2, round 6 (profiles/r06_final_step_timeline_sharded.txt, r06_call12_kbench_warm_vs_cold.txt)
        head      start of the step -> end of the shading backward (forward, scalar all-reduce, pixel chain)
        tex       tile accumulation + fold of the texture gradient, alone on the chip / beside the G-buffer backward
        geo       geometry plan (G-buffer backward .. per-frame backward) + small all-reduce + Adam of the rest
        rows(N)   row finish + Adam of T / N texture rows (latency-bound: 0.041 ms at N = 8, the whole texture at N = 1)
        next_head the next step's geometry head that may run under the all-gather (per-frame stage, skinning, binning)
        assembly  texture assembly + pyramid of the NEXT step, which wait for the all-gathered rows and which the rasteriser waits for
                  (round 5's model left this term out: 0.07 ms; a sharded form of the carried texture -- all-gather the albedo rows -- would
                  replace it by the pyramid build alone, 0.02 ms)
        one_gpu   the one-plan step every number is compared with
    assumed
        wire      time of the reduce-scatter / of the all-gather of the 50.3 MB level-0 texture: (N - 1) / N x 50.3 MB / bus bandwidth

    period(N) = head + max( texture path , geometry path )
        texture path (texture first)  = tex_alone + RS + rows + AG + assembly - min(next_head, AG + assembly)
        texture path (geometry beside) = tex_beside + RS + rows + AG + assembly - min(next_head, AG + assembly)
        geometry path (texture first)  = tex_alone + geo        (the geometry plan waits for the fold)
        geometry path (geometry beside) = geo

    python tools/scale_model.py [--bus 300 200 150] [--gpus 2 4 8] [--carried]
"""
import argparse

HEAD, TEX_ALONE, TEX_BESIDE, GEO, NEXT_HEAD, ONE_GPU = 0.541, 0.114, 0.215, 0.274, 0.095, 0.777     # (the final tree: the shading backward over the list of covered pixels)
ASSEMBLY, ASSEMBLY_CARRIED = 0.070, 0.020
TEX_MB = 50.3
FRAMES = 16


def rows(n):
    return 0.041 + (0.093 - 0.041) * (8 - min(n, 8)) / 7.0            # (measured at N = 1 and N = 8, linear in between)


def period(n, bus_gbs, tex_first, assembly=ASSEMBLY):
    wire = (n - 1) / n * TEX_MB * 1e-3 / bus_gbs * 1e3                 # ms
    tex = TEX_ALONE if tex_first else TEX_BESIDE
    texture = tex + wire + rows(n) + wire + assembly - min(NEXT_HEAD, wire + assembly)
    geometry = (TEX_ALONE if tex_first else 0.0) + GEO
    return HEAD + max(texture, geometry), wire


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bus", type=float, nargs="+", default=[300.0, 200.0, 150.0], help="bus bandwidth of the collective, GB/s")
    ap.add_argument("--gpus", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--carried", action="store_true", help="with a sharded form of the carried texture (not built): assembly = the pyramid build alone")
    a = ap.parse_args()
    asm = ASSEMBLY_CARRIED if a.carried else ASSEMBLY
    print(f"weak scaling, {FRAMES} frames per GPU, one-GPU step {ONE_GPU} ms = {FRAMES / ONE_GPU * 1e3:.0f} frames/s")
    print(f"{'bus GB/s':>8} {'N':>3} {'wire ms':>8} | {'texture first':>28} | {'geometry beside':>28}")
    for bus in a.bus:
        for n in a.gpus:
            cells = []
            for tf in (True, False):
                p, wire = period(n, bus, tf, asm)
                cells.append(f"{p:.3f} ms {n * FRAMES / p * 1e3:8.0f} f/s {n * ONE_GPU / p:5.2f}x")
            print(f"{bus:8.0f} {n:3d} {wire:8.3f} | {cells[0]:>28} | {cells[1]:>28}")
    print("(the shipped default: texture first for N >= 4, geometry beside below; config 5 -- independent subjects -- has no exchange: N x)")


if __name__ == "__main__":
    main()
