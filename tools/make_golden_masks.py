#!/usr/bin/env python3
"""Golden vectors for the FLAME_masks.pkl path of vhap_amd.topology: the reference's UNMODIFIED FlameMask
(vhap/model/flame.py:719-1054: process_vertex_mask, create_custom_mask, construct_vid_table, process_face_mask,
process_face_clusters, and the update() that FlameHead.add_teeth issues at flame.py:503-504) run on a SYNTHETIC FLAME_masks.pkl --
the 14 part masks FLAME ships, filled with the uv-sampled regions of the topology asset (the licensed file is absent) -- over the
obj topology with the teeth faces appended.

Runs only in the build container (needs /root/reference).  Output: tests/golden/flame_masks_golden.npz; the synthetic pkl itself is
rebuilt from the same asset by the test (tests/test_assets_cpu.py), so it does not travel.
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "flame_masks_golden.npz")


def synthetic_part_masks():
    from vhap_amd.topology import FLAME_PART_NAMES, FlameTopology
    t = FlameTopology(add_teeth=False)
    return {k: np.asarray(t.v_regions[k], np.int64) for k in FLAME_PART_NAMES}


def main():
    from make_golden_energy import load_reference
    from vhap_amd.topology import TEX_CLUSTERS, FlameTopology
    _, _, _, FL = load_reference()
    parts = synthetic_part_masks()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "FLAME_masks.pkl")
        with open(path, "wb") as f:
            pickle.dump(parts, f, protocol=2)
        topo0 = FlameTopology(add_teeth=False)
        faces0 = torch.from_numpy(topo0.faces.astype(np.int64))
        faces0_t = torch.from_numpy(topo0.faces_uv.astype(np.int64))
        fm = FL.FlameMask(flame_parts_path=path, faces=faces0, faces_t=faces0_t, num_verts=topo0.num_verts_orig, num_faces=faces0.shape[0],
                          face_clusters=list(TEX_CLUSTERS))
    out = {"v_names": np.array(sorted(k for k, _ in fm.v))}
    for k, b in fm.v:
        out[f"v/{k}"] = b.numpy()
    out["f_names_noteeth"] = np.array(sorted(k for k, _ in fm.f))
    for k, b in fm.f:
        out[f"f0/{k}"] = b.numpy()
    out["fid2cid_noteeth"] = fm.fid2cid.numpy()
    # what FlameHead.add_teeth does to the mask (flame.py:264-272, 500-504): three new vertex regions, then update() over the grown topology
    topo1 = FlameTopology(add_teeth=True)
    nv = topo0.num_verts_orig
    up = torch.cat([torch.arange(0, 15), torch.arange(30, 45), torch.arange(60, 75), torch.arange(75, 90)]) + nv
    lo = torch.cat([torch.arange(15, 30), torch.arange(45, 60), torch.arange(90, 105), torch.arange(105, 120)]) + nv
    fm.v.register_buffer("teeth_upper", up)
    fm.v.register_buffer("teeth_lower", lo)
    fm.v.register_buffer("teeth", torch.cat([up, lo]))
    fm.num_verts = topo1.num_verts
    fm.update(torch.from_numpy(topo1.faces.astype(np.int64)), torch.from_numpy(topo1.faces_uv.astype(np.int64)))
    out["f_names"] = np.array(sorted(k for k, _ in fm.f))
    for k, b in fm.f:
        out[f"f/{k}"] = b.numpy()
    out["fid2cid"] = fm.fid2cid.numpy()
    for regions in (["hair", "boundary", "neck"], ["hair", "bottomline"], ["left_ear", "right_ear", "neck", "left_eye", "right_eye", "lips_tight"]):
        key = "+".join(regions)
        out[f"vid/{key}"] = fm.get_vid_by_region(regions).numpy()
        if all(r in fm.f.keys() for r in regions):                # (a region without a face has no buffer in the reference: AttributeError)
            out[f"fid/{key}"] = fm.get_fid_by_region(regions).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes;", len(out["v_names"]), "vertex regions,", len(out["f_names"]), "face regions")


if __name__ == "__main__":
    main()
