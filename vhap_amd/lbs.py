"""Linear blend skinning for the 5-joint FLAME head -- host-side mirror of vhap/model/lbs.py.

Same public names and argument meaning as the reference module (batch_rodrigues :25,
vertices2landmarks :60, lbs :101, vertices2joints :198, blend_shapes :218, batch_rigid_transform
:254) so callers and tests read the same.  Written for the MI355X path: the blendshape contraction
is one GEMM on a pre-flattened basis, the kinematic chain is unrolled for FLAME's fixed tree, and
`lbs()` never materialises the [B,V,4,4] per-vertex transforms of the reference (it applies the
blended 3x4 directly).  torch is the host language here; the heavy contractions go through
vhap_amd.flame.FlameHead which dispatches to the HIP kernels when enabled.
"""
import torch


def batch_rodrigues(rot_vecs, epsilon=1e-8, dtype=torch.float32):
    """[N,3] axis-angle -> [N,3,3].  angle = ||r + 1e-8||, R = I + sin(a) K + (1 - cos(a)) K K."""
    angle = (rot_vecs + 1e-8).norm(dim=1, keepdim=True)
    ax = rot_vecs / angle
    s, c = torch.sin(angle), torch.cos(angle)
    x, y, z = ax[:, 0], ax[:, 1], ax[:, 2]
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], dim=1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).expand_as(K)
    return eye + s[:, :, None] * K + (1 - c)[:, :, None] * (K @ K)


def blend_shapes(betas, shape_disps):
    """betas [B,NB], shape_disps [V,3,NB] -> [B,V,3]."""
    V = shape_disps.shape[0]
    return (betas @ shape_disps.reshape(V * 3, -1).t()).view(-1, V, 3)


def vertices2joints(J_regressor, vertices):
    """J_regressor [J,V], vertices [B,V,3] -> [B,J,3]."""
    return torch.matmul(J_regressor, vertices)


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary_coords):
    """Barycentric landmark gather.  lmk_faces_idx [L] or [B,L]; lmk_bary_coords [L,3] or [B,L,3]."""
    if lmk_faces_idx.dim() == 2:
        lmk_faces_idx, lmk_bary_coords = lmk_faces_idx[0], lmk_bary_coords[0]
    corner = faces[lmk_faces_idx]                                   # [L,3]
    tri_v = vertices[:, corner]                                     # [B,L,3,3]
    return (tri_v * lmk_bary_coords[None, :, :, None].to(vertices.dtype)).sum(dim=2)


def batch_rigid_transform(rot_mats, joints, parents, dtype=torch.float32):
    """rot_mats [B,J,3,3], joints [B,J,3] -> posed joints [B,J,3], relative transforms [B,J,4,4]."""
    B, J = joints.shape[:2]
    par = parents if isinstance(parents, (list, tuple)) else [int(p) for p in parents]   # pass a list: no device sync
    rel_t = joints.clone()
    for j in range(1, J):
        rel_t[:, j] = joints[:, j] - joints[:, par[j]]
    G_R, G_t = [rot_mats[:, 0]], [rel_t[:, 0]]
    for j in range(1, J):
        G_R.append(G_R[par[j]] @ rot_mats[:, j])
        G_t.append((G_R[par[j]] @ rel_t[:, j, :, None])[..., 0] + G_t[par[j]])
    GR, Gt = torch.stack(G_R, 1), torch.stack(G_t, 1)               # [B,J,3,3], [B,J,3]
    A = torch.zeros(B, J, 4, 4, dtype=joints.dtype, device=joints.device)
    A[:, :, :3, :3] = GR
    A[:, :, :3, 3] = Gt - (GR @ joints[..., None])[..., 0]          # remove the rest-pose joint location
    A[:, :, 3, 3] = 1
    return Gt, A


def lbs(pose, v_shaped, posedirs, J_regressor, parents, lbs_weights, pose2rot=True, dtype=torch.float32):
    """pose [B,15], v_shaped [B,V,3] -> verts [B,V,3], posed joints [B,J,3], neck transform [B,4,4]."""
    B = pose.shape[0]
    J = vertices2joints(J_regressor, v_shaped)
    if pose2rot:
        R = batch_rodrigues(pose.reshape(-1, 3)).view(B, -1, 3, 3)
    else:
        R = pose.view(B, -1, 3, 3)
    eye = torch.eye(3, dtype=pose.dtype, device=pose.device)
    pose_feature = (R[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + (pose_feature @ posedirs).view(B, -1, 3)
    J_posed, A = batch_rigid_transform(R, J, parents)
    T = torch.matmul(lbs_weights, A[:, :, :3, :].reshape(B, -1, 12)).view(B, -1, 3, 4)   # blended 3x4
    verts = (T[..., :3] @ v_posed[..., None])[..., 0] + T[..., 3]
    return verts, J_posed, A[:, 1]
