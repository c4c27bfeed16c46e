"""Oracle-side execution of the '2-stage' backbone slice (BASELINE.json configs[1]):
InputLayer(mode 4) -> SubmConv3 C_in->32 -> [Identity | BNReLU -> SubmConv3 -> BNReLU -> SubmConv3] -> add
(reference: SparseConvNet/sparseconvnet/fpn_net.py:42-44,60-75).  Forward and hand-written
backward composed from oracle_lib calls -- test infrastructure only."""
import numpy as np

import oracle_lib as O


def two_stage_forward(locs, feats, W1, W2, W3, bn1, bn2, eps=1e-4, momentum=0.95):
    """bn1/bn2: dict(weight, bias, running_mean, running_var). Returns a cache dict."""
    c = {}
    c["il"] = il = O.input_layer(locs, feats, 4)
    c["rb"] = rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    V = il["V"]
    c["x0"] = il["out"]
    c["x1"], m1 = O.conv_fwd(c["x0"], W1, rb, V)
    c["y1"], c["sm1"], c["si1"], c["rm1"], c["rv1"] = O.bn_fwd(c["x1"], bn1["weight"], bn1["bias"],
                                                             bn1["running_mean"], bn1["running_var"], eps,
                                                             momentum, True, 0.0)
    c["x2"], m2 = O.conv_fwd(c["y1"], W2, rb, V)
    c["y2"], c["sm2"], c["si2"], c["rm2"], c["rv2"] = O.bn_fwd(c["x2"], bn2["weight"], bn2["bias"],
                                                             bn2["running_mean"], bn2["running_var"], eps,
                                                             momentum, True, 0.0)
    c["x3"], m3 = O.conv_fwd(c["y2"], W3, rb, V)
    c["out"] = c["x1"] + c["x3"]
    c["macs"] = m1 + m2 + m3
    return c


def two_stage_backward(c, g_out, W1, W2, W3, bn1, bn2):
    rb = c["rb"]
    d_y2, dW3, _ = O.conv_bwd(c["y2"], g_out, W3, rb)
    d_x2, dbn2w, dbn2b, _ = O.bn_bwd(c["x2"], c["y2"], d_y2, c["sm2"], c["si2"], bn2["weight"], 0.0)
    d_y1, dW2, _ = O.conv_bwd(c["y1"], d_x2, W2, rb)
    d_x1b, dbn1w, dbn1b, _ = O.bn_bwd(c["x1"], c["y1"], d_y1, c["sm1"], c["si1"], bn1["weight"], 0.0)
    d_x1 = g_out + d_x1b
    d_x0, dW1, _ = O.conv_bwd(c["x0"], d_x1, W1, rb)
    d_feats = O.input_layer_bwd(c["il"], d_x0)
    return dict(dW1=dW1, dW2=dW2, dW3=dW3, dbn1w=dbn1w, dbn1b=dbn1b, dbn2w=dbn2w, dbn2b=dbn2b, d_feats=d_feats,
                d_x0=d_x0)
