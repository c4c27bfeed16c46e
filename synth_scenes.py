"""Seeded synthetic SUNCG-shaped scenes, NMS box sets and batch collation.

Survey-authored generator, not reference code.  The
voxelisation step restates the reference dataset's host quantisation
(/root/reference/data3d/suncg_utils/suncg_dataset.py:126-188: scale, shift to
the minimum, drop points outside FULL_SCALE, truncate to int64) and the
`trainMerge` collate (/root/reference/data3d/data.py:25-37: concatenate scenes,
append the batch-index column).
"""
import numpy as np

FULL_SCALE = (4096, 4096, 512)


def synth_points(npts, seed, ext=(16.0, 12.0, 2.7), nwalls=14):
    """fp64 [~npts, 3] points on floor + ceiling + wall slabs, min at origin.
    Returns (points, rng) so callers draw the feature stand-ins from the same
    stream the survey's probes used."""
    rng = np.random.default_rng(seed)
    pts = []
    n_fc = npts // 4
    for z in (0.0, ext[2]):
        p = rng.random((n_fc, 3))
        p[:, 0] *= ext[0]
        p[:, 1] *= ext[1]
        p[:, 2] = z
        pts.append(p)
    per = (npts - 2 * n_fc) // nwalls
    for w in range(nwalls):
        c = rng.random(2) * np.array(ext[:2])
        L = 2 + rng.random() * 6
        yaw = rng.choice([0, np.pi / 2]) if w % 5 else rng.random() * np.pi
        t = (rng.random(per) - 0.5) * L
        z = rng.random(per) * ext[2]
        pts.append(np.stack([c[0] + t * np.cos(yaw), c[1] + t * np.sin(yaw), z], 1))
    p = np.concatenate(pts, 0)
    p -= p.min(0)
    return p, rng


def voxelize_scene(xyz, rng, voxel_scale, full_scale=FULL_SCALE, c_extra=6):
    """-> (locs int64 [n,3], feats float32 [n, 3 + c_extra])."""
    a = xyz * voxel_scale
    keep = (a < np.array(full_scale)).all(1)
    locs = np.trunc(a[keep]).astype(np.int64)
    n = locs.shape[0]
    feats = np.concatenate([xyz[keep], rng.random((n, c_extra))], 1).astype(np.float32)
    return locs, feats


def make_scene(npts=80000, seed=0, voxel_scale=20, ext=(16.0, 12.0, 2.7)):
    xyz, rng = synth_points(npts, seed, ext)
    return voxelize_scene(xyz, rng, voxel_scale)


def make_batch(batch_size, npts=80000, base_seed=0, voxel_scale=20, ext=(16.0, 12.0, 2.7)):
    """trainMerge-style batch: locs int64 [sum n, 4] (x,y,z,b), feats [sum n, C]."""
    L, F = [], []
    for b in range(batch_size):
        l, f = make_scene(npts, base_seed + b, voxel_scale, ext)
        L.append(np.concatenate([l, np.full((l.shape[0], 1), b, np.int64)], 1))
        F.append(f)
    return np.concatenate(L, 0), np.concatenate(F, 0)


def make_gt_boxes(n_gt=40, seed=0, ext=(16.0, 12.0)):
    """[n_gt,7] yx_zb ground-truth wall boxes: the same draw `make_nms_boxes` clusters its candidates around"""
    rng = np.random.default_rng(seed)
    gt = np.zeros((n_gt, 7))
    gt[:, 0] = rng.random(n_gt) * ext[0]
    gt[:, 1] = rng.random(n_gt) * ext[1]
    gt[:, 3] = 0.09 + rng.random(n_gt) * 0.21
    gt[:, 4] = 0.5 + rng.random(n_gt) * 5.5
    gt[:, 5] = 2.4 + rng.random(n_gt) * 0.4
    gt[:, 6] = rng.choice([0.0, np.pi / 2, -np.pi / 2], n_gt)
    return gt.astype(np.float32)


def make_nms_boxes(n=2000, seed=0, n_gt=40, ext=(16.0, 12.0)):
    """[n,7] yx_zb boxes (xc,yc,zb,thick,len,h,yaw) clustered around n_gt walls + scores."""
    rng = np.random.default_rng(seed)
    gt = np.zeros((n_gt, 7))
    gt[:, 0] = rng.random(n_gt) * ext[0]
    gt[:, 1] = rng.random(n_gt) * ext[1]
    gt[:, 2] = 0.0
    gt[:, 3] = 0.09 + rng.random(n_gt) * 0.21
    gt[:, 4] = 0.5 + rng.random(n_gt) * 5.5
    gt[:, 5] = 2.4 + rng.random(n_gt) * 0.4
    gt[:, 6] = rng.choice([0.0, np.pi / 2, -np.pi / 2], n_gt)
    idx = rng.integers(0, n_gt, n)
    b = gt[idx].copy()
    b[:, 0:2] += rng.normal(0, 0.15, (n, 2))
    b[:, 2] += rng.normal(0, 0.05, n)
    b[:, 3] = np.clip(b[:, 3] + rng.normal(0, 0.02, n), 0.09, 0.3)
    b[:, 4] = np.clip(b[:, 4] * (1 + rng.normal(0, 0.1, n)), 0.5, 6.0)
    b[:, 5] = np.clip(b[:, 5] + rng.normal(0, 0.05, n), 2.4, 2.8)
    b[:, 6] += rng.normal(0, 0.05, n)
    scores = rng.random(n)
    return b.astype(np.float32), scores.astype(np.float32)
