"""ORACLE (test infrastructure only -- never imported by shapy_amd/).

NumPy float32 restatement of everything between the backbone features and the posed
SMPL-X body on the SHAPY hot path.  Every function cites the reference lines it follows.
All arithmetic is done in float32 (the reference's dtype); reductions use NumPy's
pairwise float32 summation, which is *not* the order of ATen/oneDNN -- comparisons use
the 1e-4 / 1e-5 tolerances stated in the tests, not bit equality.

Pinned by: tests/golden/img_00_pins.npz (the reference's shipped SHAPY_A output: 6-D
decoder, camera projection) and tests/golden/regressor_golden.npz (outputs of the real
reference modules on CPU for seeded synthetic weights / SMPL-X buffers).
"""
import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------------------
# iterative regressor
# ------------------------------------------------------------------------------------------
def mlp_forward(x, layers):
    """MLP.forward (models/common/networks.py:392-400) with activation 'none',
    normalisation 'none', dropout in eval mode (configs/b2a_expose_hrnet_demo.yaml:200-207):
    a chain of affine maps.  ``layers`` = [(W, b), ...] in nn.Linear layout [out, in]."""
    for W, b in layers:
        x = (x @ W.T + b).astype(f32)
    return x


def iterative_regression(features, mean_param, layers, num_stages=3):
    """IterativeRegression.forward (networks.py:536-592): x_i = cat[features, p_{i-1}],
    p_i = p_{i-1} + MLP(x_i), p_{-1} = mean."""
    B = features.shape[0]
    cond = np.broadcast_to(mean_param.reshape(1, -1), (B, mean_param.size)).astype(f32)
    params = []
    prev = cond
    for _ in range(num_stages):
        inp = np.concatenate([features, prev], axis=1).astype(f32)
        delta = mlp_forward(inp, layers)
        prev = (prev[:, :delta.shape[1]] + delta).astype(f32)
        params.append(prev)
    return params


# ------------------------------------------------------------------------------------------
# pose decoders
# ------------------------------------------------------------------------------------------
def _normalize(v, eps=1e-12):
    # F.normalize: v / max(||v||_2, eps)
    n = np.sqrt((v * v).sum(axis=1, keepdims=True, dtype=f32)).astype(f32)
    return (v / np.maximum(n, f32(eps))).astype(f32)


def cont_rot_repr_decode(x):
    """ContinuousRotReprDecoder.forward (models/common/pose_utils.py:138-153).
    x: [B, 6k] viewed as (-1, 3, 2) -> [B, k, 3, 3]; b1,b2,b3 are the *columns*."""
    B = x.shape[0]
    r = x.reshape(-1, 3, 2).astype(f32)
    a1, a2 = r[:, :, 0], r[:, :, 1]
    b1 = _normalize(a1)
    dot = (b1 * a2).sum(axis=1, keepdims=True, dtype=f32)
    b2 = _normalize((a2 - dot * b1).astype(f32))
    b3 = np.cross(b1, b2).astype(f32)
    return np.stack([b1, b2, b3], axis=-1).reshape(B, -1, 3, 3)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """utils/rotation_utils.py:5-37: eps is added to the *vector* before the norm."""
    rv = rot_vecs.astype(f32)
    angle = np.linalg.norm((rv + f32(epsilon)).astype(f32), axis=1, keepdims=True).astype(f32)
    rot_dir = (rv / angle).astype(f32)
    cos = np.cos(angle)[:, None, :].astype(f32)
    sin = np.sin(angle)[:, None, :].astype(f32)
    rx, ry, rz = rot_dir[:, 0], rot_dir[:, 1], rot_dir[:, 2]
    z = np.zeros_like(rx)
    K = np.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], axis=1).reshape(-1, 3, 3).astype(f32)
    ident = np.eye(3, dtype=f32)[None]
    return (ident + sin * K + (f32(1) - cos) * (K @ K)).astype(f32)


def rot_mat_to_euler(R):
    """utils/rotation_utils.py:86-92."""
    sy = np.sqrt(R[:, 0, 0] * R[:, 0, 0] + R[:, 1, 0] * R[:, 1, 0]).astype(f32)
    return np.arctan2(-R[:, 2, 0], sy).astype(f32)


# ------------------------------------------------------------------------------------------
# SMPL-X
# ------------------------------------------------------------------------------------------
def blend_shapes(betas, shape_disps):
    """lbs.py:218-239: einsum('bl,mkl->bmk')."""
    return np.einsum('bl,mkl->bmk', betas.astype(f32), shape_disps.astype(f32)).astype(f32)


def vertices2joints(J_regressor, vertices):
    """lbs.py:199-215: einsum('bik,ji->bjk')."""
    return np.einsum('bik,ji->bjk', vertices, J_regressor).astype(f32)


def batch_rigid_transform(rot_mats, joints, parents):
    """lbs.py:242-295 (+ transform_mat, body_models/utils.py:14-24)."""
    B, N = rot_mats.shape[:2]
    rel = joints.copy()
    rel[:, 1:] -= joints[:, parents[1:]]
    T = np.zeros((B, N, 4, 4), f32)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, N):
        chain.append((chain[parents[i]] @ T[:, i]).astype(f32))
    transforms = np.stack(chain, axis=1)
    posed = transforms[:, :, :3, 3].copy()
    jh = np.concatenate([joints, np.zeros((B, N, 1), f32)], axis=2)[..., None]
    corr = (transforms @ jh).astype(f32)                      # B,N,4,1
    rel_transforms = transforms.copy()
    rel_transforms[:, :, :, 3:4] -= corr
    return posed, rel_transforms.astype(f32)


def lbs(betas, pose_rotmats, v_template, shapedirs, posedirs, J_regressor, parents,
        lbs_weights):
    """lbs() with pose2rot=False (lbs.py:99-196).  posedirs is [P, V*3] as registered at
    body_models.py:150-153."""
    B = betas.shape[0]
    v_shaped = (v_template[None] + blend_shapes(betas, shapedirs)).astype(f32)
    J = vertices2joints(J_regressor, v_shaped)
    ident = np.eye(3, dtype=f32)
    pose_feature = (pose_rotmats[:, 1:] - ident).reshape(B, -1).astype(f32)
    pose_offsets = (pose_feature @ posedirs).reshape(B, -1, 3).astype(f32)
    v_posed = (pose_offsets + v_shaped).astype(f32)
    J_transformed, A = batch_rigid_transform(pose_rotmats, J, parents)
    T = np.einsum('vj,bjmn->bvmn', lbs_weights, A).astype(f32)
    vh = np.concatenate([v_posed, np.ones((B, v_posed.shape[1], 1), f32)], axis=2)
    verts = np.einsum('bvmn,bvn->bvm', T, vh).astype(f32)[:, :, :3]
    return dict(vertices=verts, joints=J_transformed, v_shaped=v_shaped, v_posed=v_posed, A=A)


def find_dynamic_lmk_idx_and_bcoords(pose, dyn_faces_idx, dyn_bcoords, neck_kin_chain):
    """lbs.py:20-49."""
    rot = pose[:, neck_kin_chain]
    rel = np.broadcast_to(np.eye(3, dtype=f32), (pose.shape[0], 3, 3)).copy()
    for idx in range(len(neck_kin_chain)):
        rel = (rot[:, idx] @ rel).astype(f32)
    ang = (-rot_mat_to_euler(rel) * f32(180.0) / f32(np.pi)).astype(f32)
    y = np.round(np.minimum(ang, f32(39))).astype(np.int64)     # torch.round: half-to-even
    neg_mask = (y < 0).astype(np.int64)
    mask = (y < -39).astype(np.int64)
    neg_vals = mask * 78 + (1 - mask) * (39 - y)
    y = neg_mask * neg_vals + (1 - neg_mask) * y
    return dyn_faces_idx[y], dyn_bcoords[y]


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary_coords):
    """lbs.py:52-94."""
    B = vertices.shape[0]
    lmk_faces = faces[lmk_faces_idx.reshape(-1)].reshape(B, -1, 3)
    lmk_vertices = np.stack([vertices[b][lmk_faces[b]] for b in range(B)])   # B,L,3,3
    return (lmk_vertices * lmk_bary_coords[..., None]).sum(axis=2, dtype=f32).astype(f32)


def find_joint_kin_chain(joint_id, parents):
    chain = []
    cur = joint_id
    while cur != -1:
        chain.append(cur)
        cur = parents[cur]
    return chain


def smplx_forward(model, global_rot, body_pose, betas, use_face_contour=True,
                  num_betas=10, num_expr=10):
    """SMPLX.forward (body_models.py:628-767) for the SHAPY_A call: only global_rot,
    body_pose and betas are given; jaw/eyes/hands are identity, expression is zero.
    ``model``: dict with the SMPLX_NEUTRAL.npz keys."""
    B = betas.shape[0]
    parents = model['kintree_table'][0].astype(np.int64).copy()
    parents[0] = -1
    eye = lambda n: np.broadcast_to(np.eye(3, dtype=f32), (B, n, 3, 3))
    full_pose = np.concatenate([global_rot, body_pose, eye(1), eye(1), eye(1),
                                eye(15), eye(15)], axis=1).astype(f32)
    shapedirs = model['shapedirs'][:, :, :num_betas].astype(f32)
    expr_dirs = model['shapedirs'][:, :, 300:300 + num_expr].astype(f32)
    comps = np.concatenate([betas, np.zeros((B, num_expr), f32)], axis=1)
    sdirs = np.concatenate([shapedirs, expr_dirs], axis=-1)
    P = model['posedirs'].shape[-1]
    posedirs = model['posedirs'].reshape(-1, P).T.astype(f32)
    out = lbs(comps, full_pose, model['v_template'].astype(f32), sdirs, posedirs,
              model['J_regressor'].astype(f32), parents, model['weights'].astype(f32))
    faces = model['f'].astype(np.int64)
    lmk_idx = np.broadcast_to(model['lmk_faces_idx'][None], (B, 51))
    lmk_bc = np.broadcast_to(model['lmk_bary_coords'][None].astype(f32), (B, 51, 3))
    if use_face_contour:
        chain = np.array(find_joint_kin_chain(15, list(parents)), np.int64)
        di, db = find_dynamic_lmk_idx_and_bcoords(
            full_pose, model['dynamic_lmk_faces_idx'],
            model['dynamic_lmk_bary_coords'].astype(f32), chain)
        lmk_idx = np.concatenate([lmk_idx, di], axis=1)
        lmk_bc = np.concatenate([lmk_bc, db], axis=1)
    landmarks = vertices2landmarks(out['vertices'], faces, lmk_idx, lmk_bc)
    joints = np.concatenate([out['joints'], landmarks], axis=1).astype(f32)
    v_shaped = (model['v_template'].astype(f32)[None] + blend_shapes(betas, shapedirs)).astype(f32)
    return dict(joints=joints, vertices=out['vertices'], v_shaped=v_shaped, faces=faces,
                full_pose=full_pose)


# ------------------------------------------------------------------------------------------
# camera
# ------------------------------------------------------------------------------------------
def softplus(x):
    """F.softplus, beta=1, threshold=20."""
    x = x.astype(f32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, f32(20))))).astype(f32)


def weak_persp_project(points, scale, translation):
    """WeakPerspectiveCamera.forward, scale_first=False (camera_projection.py:181-213)."""
    return (scale.reshape(-1, 1, 1) * (points[:, :, :2] + translation.reshape(-1, 1, 2))).astype(f32)


# ------------------------------------------------------------------------------------------
# parameter layout (iterative_regressor.py:81-110)
# ------------------------------------------------------------------------------------------
PARAM_SLICES = dict(global_rot=(0, 6), body_pose=(6, 132), betas=(132, 142), camera=(142, 145))


def param_mean(mean_scale=0.9):
    m = np.zeros(145, f32)
    m[0:6] = [1, 0, 0, -1, 0, 0]               # 180 deg about x (body_heads.py:103-108)
    m[6:132] = np.tile(np.array([1, 0, 0, 1, 0, 0], f32), 21)   # pose_utils.py:86-107
    m[142] = np.log(np.exp(mean_scale) - 1)     # camera_projection.py:71-80
    return m


def regressor_head(features, layers, model, num_stages=3):
    """HMRLikeRegressor.forward after the backbone (iterative_regressor.py:638-733), with
    pose_last_stage=True."""
    params = iterative_regression(features.astype(f32), param_mean(), layers, num_stages)
    stages = []
    for p in params:
        d = {}
        for name, (a, b) in PARAM_SLICES.items():
            d[name] = p[:, a:b].copy()
        d['raw_global_rot'] = d['global_rot']
        d['raw_body_pose'] = d['body_pose']
        d['global_rot'] = cont_rot_repr_decode(d['raw_global_rot'])
        d['body_pose'] = cont_rot_repr_decode(d['raw_body_pose'])
        stages.append(d)
    last = stages[-1]
    body = smplx_forward(model, last['global_rot'], last['body_pose'], last['betas'])
    cam = last['camera']
    scale = softplus(cam[:, 0:1])
    proj = weak_persp_project(body['joints'], scale, cam[:, 1:3])
    last.update(body)
    last['proj_joints'] = proj
    return dict(params=params, stages=stages, proj_joints=proj, scale=scale)
