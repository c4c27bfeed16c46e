"""The oracle composition of FPN_Net (tests/ref_net.py::FpnOracle) checked on the CPU: structure against the
module tree, and its hand-written reverse pass against a central-difference directional derivative.

The derivative check runs with leakiness = 1 (activation = identity) so that the composed function is smooth
and the finite difference converges: it verifies the WIRING of the tape (residual adds, lateral adds, deconv
column swap, rule-book reuse, BatchNorm train-mode backward).  The ReLU mask itself is pinned at op level,
bit-equal to the reference's BatchNormalization_BackwardPass (tests/test_oracle_ref_kernels.py); with ReLU on,
a 700-point scene has so many activations within the step of 0 that the finite difference is kink noise."""
import numpy as np
import torch

import ref_net
import synth_scenes as S


def _small_fpn():
    import sparseconvnet as scn
    torch.manual_seed(3)
    return scn.FPN_Net([64, 64, 16], 3, ["xyz", "color", "normal"], 1, [8, 8, 16], 8, True, [2, 1], [2, 1],
                       [[[2, 2, 2]] * 2, [[2, 2, 2]] * 2], [[64, 64, 16], [32, 32, 8]], [0, 1, 2, 3], leakiness=0,
                       voxel_scale=4, bn_momentum=0.95)


def _oracle(P, leak=0.0):
    return ref_net.FpnOracle(P, (64, 64, 16), [[2, 2, 2]] * 2, [[2, 2, 2]] * 2, [[64, 64, 16], [32, 32, 8]], (2, 1),
                             (2, 1), (0, 1, 2, 3), leakiness=leak)


def _scene():
    rng = np.random.default_rng(0)
    n = 700
    locs = np.stack([rng.integers(0, 40, n), rng.integers(0, 30, n), rng.integers(0, 10, n),
                     np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
    return locs, rng.standard_normal((n, 9)).astype(np.float32)


def test_param_walk_covers_every_trainable_tensor_of_the_module_tree():
    net = _small_fpn()
    names = ref_net.fpn_param_names(net)
    mine = {id(p) for p in names.values()}
    unused = ("linear", "layers_out")   # owned but never called by forward (fpn_net.py:46-50)
    for n, p in net.named_parameters():
        if n.startswith(unused):
            continue
        assert id(p) in mine, n
    from test_cabi_and_host import default_fpn
    big = default_fpn()
    P = ref_net.fpn_params(big)
    n_par = sum(v.size if isinstance(v, np.ndarray) else v["weight"].size + v["bias"].size for v in P.values())
    assert n_par == 21212660 - (32 * 20 + 20) - 64     # minus linear head and layers_out BN


def test_fpn_oracle_backward_is_the_derivative_of_its_forward():
    net = _small_fpn()
    P = ref_net.fpn_params(net)
    locs, feats = _scene()
    fo = _oracle(P, 1.0)
    rpn, roi = fo.forward(locs, feats)
    assert len(rpn) == 4 and [m.spatial for m in rpn[:2]] == [(64, 64, 16), (32, 32, 8)]
    assert rpn[2].spatial == (64, 64, 1) and rpn[3].spatial == (32, 32, 1)
    assert fo.macs > 0
    rng = np.random.default_rng(1)
    G = [rng.standard_normal(m.v.shape).astype(np.float32) for m in rpn]
    grads = fo.backward(G)
    conv_keys = [k for k, v in P.items() if isinstance(v, np.ndarray)]
    bn_keys = [k for k, v in P.items() if not isinstance(v, np.ndarray)]
    assert all(k in grads for k in conv_keys) and all(k + ".weight" in grads for k in bn_keys)

    def loss(Pq, fq):
        o = _oracle(Pq, 1.0)
        r, _ = o.forward(locs, fq)
        return sum(float((g.astype(np.float64) * m.v).sum()) for g, m in zip(G, r))

    # directional derivative over ALL parameters and the input features at once
    D = {k: rng.standard_normal(P[k].shape).astype(np.float32) for k in conv_keys}
    Db = {k: (rng.standard_normal(P[k]["weight"].shape).astype(np.float32),
              rng.standard_normal(P[k]["bias"].shape).astype(np.float32)) for k in bn_keys}
    Df = rng.standard_normal(feats.shape).astype(np.float32)
    want = sum(float((grads[k].astype(np.float64) * D[k]).sum()) for k in conv_keys)
    want += sum(float((grads[k + ".weight"].astype(np.float64) * Db[k][0]).sum() +
                      (grads[k + ".bias"].astype(np.float64) * Db[k][1]).sum()) for k in bn_keys)
    want += float((grads["d_feats"].astype(np.float64) * Df).sum())

    def shifted(e):
        Pq = {}
        for k in conv_keys:
            Pq[k] = (P[k] + np.float32(e) * D[k]).astype(np.float32)
        for k in bn_keys:
            Pq[k] = dict(P[k], weight=(P[k]["weight"] + np.float32(e) * Db[k][0]).astype(np.float32),
                         bias=(P[k]["bias"] + np.float32(e) * Db[k][1]).astype(np.float32))
        return Pq, (feats + np.float32(e) * Df).astype(np.float32)

    e = 1e-3
    num = (loss(*shifted(e)) - loss(*shifted(-e))) / (2 * e)
    assert abs(num - want) <= 0.01 * abs(want) + 1e-3, (num, want)
