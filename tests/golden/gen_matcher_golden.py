"""Generate tests/golden/matcher_golden.npz by IMPORTING the reference's pure-torch matcher and running it on
seeded inputs (build container only; needs /root/reference):

  * maskrcnn_benchmark/modeling/matcher.py:12-196     Matcher.__call__ incl. yaw_diff_constrain,
    set_low_quality_matches_ and its IGNORE_HIGHEST_MATCH_NEARBY pass (module flags as committed upstream)
  * utils3d/geometric_torch.py:4-21                    limit_period / angle_dif (the |yaw difference| fed to it,
    rpn/loss_3d.py:96-97)

Both files import torch / math only and are loaded from where they lie (no placeholder modules needed).  The
committed fixture is data only: match-quality matrices, yaws, thresholds -> the reference's matched_idxs.
Cases follow make_rpn_loss_evaluator (rpn/loss_3d.py:338-344: FG 0.55 / BG 0.2, allow_low_quality_matches=True,
YAW_THRESHOLD 0.7, config/defaults.py:147-153) plus the variations the other call sites use (yaw threshold > 1.58 =
no mask, s_3c config; allow_low_quality_matches False, ROI heads)."""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    matcher_mod = _load("ref_matcher", "maskrcnn_benchmark/modeling/matcher.py")
    geo = _load("ref_geometric_torch", "utils3d/geometric_torch.py")
    rng = np.random.default_rng(23)
    out = {}
    cases = []
    # (G, N, fg, bg, allow_low, yaw_thr, flavour)
    specs = [(7, 400, 0.55, 0.2, True, 0.7, "criterion6"), (25, 3000, 0.55, 0.2, True, 0.7, "criterion6"),
             (1, 500, 0.55, 0.2, True, 0.7, "criterion6"), (12, 800, 0.55, 0.2, True, 3.0, "criterion6"),
             (9, 600, 0.5, 0.5, False, 0.7, "iou"), (6, 300, 0.55, 0.2, True, 0.7, "row_all_masked"),
             (5, 64, 0.55, 0.2, True, 0.7, "ties")]
    for ci, (G, N, fg, bg, allow, ythr, flavour) in enumerate(specs):
        if flavour == "iou":
            mq = rng.random((G, N)).astype(np.float32) ** 3
        else:
            # criterion-6-like values: 1 - (...)/0.7 is mostly negative, a few anchors near each target score high
            mq = (1.0 - rng.random((G, N)) * 6.0).astype(np.float32)
            for g in range(G):
                hot = rng.choice(N, max(2, N // 60), replace=False)
                mq[g, hot] = rng.random(hot.shape[0]).astype(np.float32)
        tyaw = rng.choice([0.0, np.pi / 2, -np.pi / 2, 0.3, -1.2], G).astype(np.float32) + \
            (rng.standard_normal(G) * 0.05).astype(np.float32)
        ayaw = rng.choice([0.0, -1.57, -0.785, 0.785], N).astype(np.float32)
        if flavour == "row_all_masked":
            tyaw[2] = 0.4            # |diff| to every anchor yaw: 0.4, 1.17(wrapped), 1.185, 0.385 ... keep some
            tyaw[3] = np.float32(0.0)
            ayaw[:] = np.float32(-1.57)   # then target 3 (yaw 0) is masked against EVERY anchor: its row max is 0
        if flavour == "ties":
            mq = np.round(mq * 4) / 4     # quantised: exact ties between targets and between anchors
            mq = mq.astype(np.float32)
        yd = torch.abs(geo.angle_dif(torch.from_numpy(ayaw).view(1, -1), torch.from_numpy(tyaw).view(-1, 1), 0))
        m = matcher_mod.Matcher(fg, bg, allow_low_quality_matches=allow, yaw_threshold=ythr)
        got = m(torch.from_numpy(mq.copy()), yaw_diff=yd, flag="RPN", cendis=None)
        out["c%d_mq" % ci], out["c%d_tyaw" % ci], out["c%d_ayaw" % ci] = mq, tyaw, ayaw
        out["c%d_yaw_diff" % ci] = yd.numpy()
        out["c%d_matches" % ci] = got.numpy().astype(np.int64)
        out["c%d_params" % ci] = np.array([fg, bg, float(allow), ythr], np.float64)
        cases.append(flavour)
        vals, cnt = np.unique(got.numpy(), return_counts=True)
        print(ci, flavour, (G, N), "labels: pos %d ignore %d neg %d" % (int((got >= 0).sum()), int((got == -2).sum()),
                                                                       int((got == -1).sum())))
    out["n_cases"] = np.array(len(specs))
    np.savez_compressed(os.path.join(HERE, "matcher_golden.npz"), **out)


if __name__ == "__main__":
    main()
