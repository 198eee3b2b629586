"""Static mesh topology tables for the FLAME head (host side, numpy, built once).

Mirrors the topology bookkeeping of the reference's FlameHead / FlameMask
(vhap/model/flame.py:149-167 load_obj, :206-504 add_teeth, :719-1054 FlameMask) on the asset
`vhap_amd/assets/flame_topology.npz` (derived by tools/make_assets.py).  Because the topology is
fixed for the whole fit, everything the reference rebuilds per call is precomputed here:
the edge -> opposite-vertex table (nvdiffrast rebuilds an edge hash every dr.antialias call),
the vertex -> incident-corner CSR (the reference scatter_adds face normals every step,
render_nvdiffrast.py:297-316) and the sparse uniform Laplacian (the reference stores it dense,
flame.py:196, tracker.py:682-690).
"""
import os

import numpy as np

ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "flame_topology.npz")

TEX_CLUSTERS = ("skin", "hair", "boundary", "lips_tight", "teeth", "sclerae", "irises")  # config/base.py:83


def build_opposite_table(tri):
    """opp[t, i] = vertex opposite edge i (edge i joins v[(i+1)%3], v[(i+2)%3]) in the other triangle
    sharing that edge; -1 on a boundary edge."""
    tri = np.asarray(tri, np.int64)
    F = tri.shape[0]
    a = tri[:, [1, 2, 0]].reshape(-1)
    b = tri[:, [2, 0, 1]].reshape(-1)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = lo * (tri.max() + 1) + hi
    order = np.argsort(key, kind="stable")
    ks = key[order]
    opp = -np.ones(3 * F, np.int32)
    own = tri.reshape(-1)                        # vertex opposite corner slot (t,i) is tri[t,i]
    start = 0
    n = ks.shape[0]
    while start < n:
        end = start + 1
        while end < n and ks[end] == ks[start]:
            end += 1
        if end - start >= 2:
            s0, s1 = order[start], order[start + 1]
            opp[s0] = own[s1]
            opp[s1] = own[s0]
            for s in order[start + 2:end]:
                opp[s] = own[s0]
        start = end
    return opp.reshape(F, 3)


def build_vertex_corner_csr(tri, V):
    """CSR of face corners incident to each vertex: for vertex v, corners[ptr[v]:ptr[v+1]] are flat
    corner ids c = 3*t + i with tri[t, i] == v (sorted, hence deterministic gather order)."""
    flat = np.asarray(tri, np.int64).reshape(-1)
    order = np.argsort(flat, kind="stable").astype(np.int32)
    counts = np.bincount(flat, minlength=V)
    ptr = np.zeros(V + 1, np.int32)
    np.cumsum(counts, out=ptr[1:])
    return ptr, order


def build_uniform_laplacian_csr(tri, V):
    """pytorch3d Meshes.laplacian_packed(): L = D^-1 A - I with -1 on EVERY diagonal entry
    (flame.py:196).  Returns CSR (ptr [V+1], col [nnz], val [nnz]) with the diagonal stored."""
    f = np.asarray(tri, np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    e = np.unique(np.sort(e, 1), axis=0)
    src = np.concatenate([e[:, 0], e[:, 1]])
    dst = np.concatenate([e[:, 1], e[:, 0]])
    deg = np.bincount(src, minlength=V)
    rows = np.concatenate([src, np.arange(V)])
    cols = np.concatenate([dst, np.arange(V)])
    vals = np.concatenate([1.0 / deg[src], -np.ones(V)])
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    ptr = np.zeros(V + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=V), out=ptr[1:])
    return ptr, cols.astype(np.int32), vals.astype(np.float32)


FLAME_PART_NAMES = ("face", "neck", "scalp", "boundary", "right_eyeball", "left_eyeball", "right_ear", "left_ear", "forehead", "eye_region",
                    "nose", "lips", "right_eye_region", "left_eye_region")      # the keys of FLAME_masks.pkl (flame.py:759-763)


def regions_from_flame_parts(part_masks, literal_masks, num_verts):
    """The vertex regions FlameMask builds from FLAME's own part masks (FLAME_masks.pkl) -- flame.py:754-767 process_vertex_mask +
    :769-938 create_custom_mask: the part masks as they are, the hand-picked index tables (`literal_masks`: the integer literals of
    flame.py, shipped in the topology asset as vmask_*), and the derived regions -- hair = scalp minus its intersection with
    face + neck, the unions (ears, eyeballs, irises, left_eye, right_eye, eyelids, lip_inside_ring), sclerae = eyeballs minus irises,
    skin = everything but eyeballs, hair, lips_tight and boundary.  Unions are CONCATENATIONS like the reference's (an index listed
    twice counts twice in the face test of FlameTopology._finalize, flame.py:947-955)."""
    v = {k: np.asarray(m, np.int64).reshape(-1) for k, m in part_masks.items()}
    for k, m in literal_masks.items():
        v[k] = np.asarray(m, np.int64).reshape(-1)
    cat = lambda *names: np.concatenate([v[n] for n in names])
    face_and_neck = np.unique(cat("face", "neck"))
    v["hair"] = np.setdiff1d(np.unique(v["scalp"]), face_and_neck)
    v["ears"] = cat("right_ear", "left_ear")
    v["eyeballs"] = cat("right_eyeball", "left_eyeball")
    v["irises"] = cat("right_iris", "left_iris")
    v["left_eye"] = cat("left_eye_region", "left_eyeball")
    v["right_eye"] = cat("right_eye_region", "right_eyeball")
    v["eyelids"] = cat("left_eyelid", "right_eyelid")
    v["lip_inside_ring"] = np.concatenate([v["lip_inside_ring_upper"], v["lip_inside_ring_lower"], np.array([1594, 2730], np.int64)])
    v["sclerae"] = np.setdiff1d(np.unique(v["eyeballs"]), np.unique(v["irises"]))
    v["skin"] = np.setdiff1d(np.arange(num_verts, dtype=np.int64), np.unique(cat("eyeballs", "hair", "lips_tight", "boundary")))
    return v


def load_flame_masks(path):
    """FLAME_masks.pkl (flame.py:40,757): a pickled dict part name -> vertex indices (latin1: it was written by Python 2)."""
    import pickle
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    return {str(k): np.asarray(m, np.int64).reshape(-1) for k, m in d.items()}


class FlameTopology:
    """Faces / UVs / landmark embedding / region masks of the FLAME head, optionally with teeth.

    `part_masks` = the dict of FLAME_masks.pkl (load_flame_masks): the vertex regions are then built the reference's way
    (regions_from_flame_parts).  Without it (the licensed file is not redistributable) they are approximated by sampling the UV-space
    region masks of asset/flame/uv_masks.npz at the vertex UVs (tools/make_assets.py)."""

    @classmethod
    def from_flame_masks(cls, masks_path, add_teeth=True, tex_clusters=TEX_CLUSTERS):
        return cls(add_teeth=add_teeth, tex_clusters=tex_clusters, part_masks=load_flame_masks(masks_path))

    def __init__(self, add_teeth=True, tex_clusters=TEX_CLUSTERS, part_masks=None):
        d = np.load(ASSET, allow_pickle=False)
        self.v_template_obj = d["v_template"].astype(np.float32)          # [5023,3] (un-centred obj verts)
        self.verts_uvs = d["verts_uvs"].astype(np.float32)                # [5118,2]
        self.faces = d["faces"].astype(np.int32)                          # [9976,3]
        self.faces_uv = d["faces_uv"].astype(np.int32)
        self.lmk_faces_idx = d["lmk_faces_idx"].astype(np.int64)          # [70]
        self.lmk_bary_coords = d["lmk_bary_coords"].astype(np.float32)    # [70,3]
        self.num_verts_orig = self.v_template_obj.shape[0]
        self.num_faces_orig = self.faces.shape[0]
        literal = {k[6:]: d[k].astype(np.int64) for k in d.files if k.startswith("vmask_")}
        if part_masks is not None:
            missing = [k for k in FLAME_PART_NAMES if k not in part_masks]
            if missing:
                raise ValueError(f"FLAME part masks lack {missing}")
            self.v_regions = regions_from_flame_parts(part_masks, literal, self.num_verts_orig)
        else:
            names = [str(x) for x in d["region_names"]]
            bits = np.unpackbits(d["v_region"], axis=1)[:, :len(names)].astype(bool)
            self.v_regions = {n: np.nonzero(bits[:, i])[0] for i, n in enumerate(names)}
            self.v_regions.update(literal)                                # literal masks of the reference win
        T = int(d["uvmask_size"])
        self.uvmasks = {k: np.unpackbits(d[f"uvmask_{k}"])[: T * T].reshape(T, T).astype(bool)
                        for k in ("sclerae", "teeth")}
        self._f_teeth = (d["f_teeth_upper"].astype(np.int32), d["f_teeth_lower"].astype(np.int32))
        self.has_teeth = False
        self.teeth_uv = None
        if add_teeth:
            self._add_teeth_topology()
        self.tex_clusters = tuple(tex_clusters)
        self._finalize()

    # flame.py:292-300, 326-501: 120 teeth vertices / uv vertices, 168 faces appended
    def _add_teeth_topology(self):
        nv, nvt = self.num_verts_orig, self.verts_uvs.shape[0]
        u = np.linspace(0.62, 0.38, 15, dtype=np.float32)
        v = np.linspace(1 - 0.0083, 1 - 0.0425, 7, dtype=np.float32)[[3, 2, 0, 1, 3, 4, 6, 5]]
        uu, vv = np.meshgrid(u, v, indexing="ij")                         # [15,8]
        uv = np.stack([uu, vv], -1).transpose(1, 0, 2).reshape(120, 2)
        self.verts_uvs = np.concatenate([self.verts_uvs, uv], 0)
        fu, fl = self._f_teeth
        self.faces = np.concatenate([self.faces, fu + nv, fl + nv], 0)
        self.faces_uv = np.concatenate([self.faces_uv, fu + nvt, fl + nvt], 0)
        up = np.concatenate([np.arange(0, 15), np.arange(30, 45), np.arange(60, 75), np.arange(75, 90)]) + nv
        lo = np.concatenate([np.arange(15, 30), np.arange(45, 60), np.arange(90, 105), np.arange(105, 120)]) + nv
        self.v_regions["teeth_upper"], self.v_regions["teeth_lower"] = up, lo
        self.v_regions["teeth"] = np.concatenate([up, lo])
        self.has_teeth = True

    def _finalize(self):
        self.num_verts = self.num_verts_orig + (120 if self.has_teeth else 0)
        self.num_faces = self.faces.shape[0]
        V, F = self.num_verts, self.num_faces
        # face regions (flame.py:940-955): every corner adds one count per LISTING of its vertex in the region, a face is in the region
        # at 3 counts -- for a duplicate-free list: iff all 3 of its vertices are; the reference's concatenated unions may list a vertex twice
        self.f_regions = {}
        for name, vid in self.v_regions.items():
            mult = np.bincount(vid[vid < V], minlength=V)
            fid = np.nonzero(mult[self.faces].sum(1) >= 3)[0]
            if fid.size:
                self.f_regions[name] = fid
        # fid2cid (flame.py:965-984): 1 = no cluster, cluster k -> k+2, later clusters overwrite;
        # padded with a leading 0 for "background" like NVDiffRenderer.__init__ (render_nvdiffrast.py:78)
        fid2cid = np.ones(F, np.int32)
        for cid, name in enumerate(self.tex_clusters):
            if name in self.f_regions:
                fid2cid[self.f_regions[name]] = cid + 2
        self.fid2cid = np.concatenate([[0], fid2cid]).astype(np.int32)
        self.opp = build_opposite_table(self.faces)
        self.vc_ptr, self.vc_idx = build_vertex_corner_csr(self.faces, V)
        # Laplacian over the ORIGINAL faces but V rows (flame.py:196): teeth rows are diagonal-only
        self.lap_ptr, self.lap_col, self.lap_val = build_uniform_laplacian_csr(self.faces[: self.num_faces_orig], V)

    def get_vid_by_region(self, regions):
        if isinstance(regions, str):
            regions = [regions]
        if not regions:
            return np.zeros(0, np.int64)
        return np.unique(np.concatenate([self.v_regions[r] for r in regions])).astype(np.int64)

    def get_fid_by_region(self, regions):
        if isinstance(regions, str):
            regions = [regions]
        got = [self.f_regions[r] for r in regions if r in self.f_regions]
        if not got:
            return np.zeros(0, np.int64)
        return np.unique(np.concatenate(got)).astype(np.int64)

    def get_uvmask_by_region(self, regions):
        if isinstance(regions, str):
            regions = [regions]
        m = self.uvmasks[regions[0]].copy()
        for r in regions[1:]:
            m |= self.uvmasks[r]
        return m
