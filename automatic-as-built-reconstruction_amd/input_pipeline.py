"""Device-side input pipeline: the reference dataset's host quantisation + `trainMerge` collate
(data3d/suncg_utils/suncg_dataset.py:126-188 test-time path, data3d/data.py:25-37) as one HIP
kernel per scene writing straight into the batch tensors the InputLayer consumes."""
import torch

import _hip
from _hip import ptr, stream, check


def quantize_scenes(xyz_list, extra_feats_list, voxel_scale, full_scale):
    """xyz_list: per-scene [n_i,3] float32/float64 DEVICE tensors (metres); extra_feats_list: per-scene
    [n_i, C-3] float32 (colour, normal, ...) or None.  Returns (locs int64 [sum n,4], feats float32
    [sum n, C]) in the trainMerge layout.  Points outside FULL_SCALE carry the (-1,-1,-1) sentinel
    the InputLayer skips (the reference drops them on the host)."""
    lib = _hip.load()
    dev = xyz_list[0].device
    _hip.require_gpu(xyz_list[0])
    ns = [int(x.shape[0]) for x in xyz_list]
    cx = 0 if extra_feats_list is None or extra_feats_list[0] is None else int(extra_feats_list[0].shape[1])
    C_ = 3 + cx
    tot = sum(ns)
    locs = torch.empty((tot, 4), dtype=torch.int64, device=dev)
    feats = torch.empty((tot, C_), dtype=torch.float32, device=dev)
    fs = torch.tensor([int(v) for v in full_scale], dtype=torch.int32, device=dev)
    o = 0
    for b, x in enumerate(xyz_list):
        n = ns[b]
        x = x.contiguous()
        assert x.dtype in (torch.float32, torch.float64) and x.shape[1] == 3
        amin = x.amin(0) if n else torch.zeros(3, dtype=x.dtype, device=dev)
        check(lib.aabr_quantize_points(ptr(x), int(x.dtype == torch.float64), n, float(voxel_scale), ptr(amin),
                                       ptr(fs), b, locs[o:].data_ptr() if n else None,
                                       feats[o:].data_ptr() if n else None, C_, stream()))
        if cx:
            feats[o:o + n, 3:] = extra_feats_list[b]
        o += n
    return locs, feats
