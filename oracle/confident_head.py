"""ORACLE-SIDE FIXTURE (test infrastructure, NOT product code): a deterministic, CONFIDENT head for the U-Net parity tests.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

With Glorot weights no class probability reaches 0.5 and the sigmoid never reaches 0.8, so f1_m / wr_m / the 0.8 mask compare
0 with 0.  `saturate_head` replaces the two 1x1x1 head layers of an oracle by a ridge-regression
probe of the oracle's own trunk features onto the labels (atom voxels weighted up, the logits scaled by `sharp`): the trunk
stays Glorot, the predictions become confident and mostly right, TP / predicted / possible counts become non-trivial,
sig crosses 0.8.  The weights are inputs like any other -- engine and oracle both receive them."""
import numpy as np

from . import numpy_ref as R


def saturate_head(orc, X, lab, training, sharp=8.0, sharp_sig=3.0, atom_weight=25.0, ridge=1e-3):
    """sharp / sharp_sig: logit scales.  Chosen so that |logits| stay below ~12 (softmax) / ~8 (sigmoid): the engine's trunk
    output carries an fp32 error of up to 1e-5 of its maximum, a logit of magnitude z turns that into an absolute logit
    error ~1e-5 z, and north_star's 1e-5 on the probabilities survives only while z is O(10); at z = 30 (sharp = 12 on both
    heads) the sigmoid sat 1.6e-5 off and -log(1 - p) of p = 1 - 4e-7 is representation noise in fp32 (measured on MI355X)."""
    cache = {}
    orc.forward(X, training=training, cache=cache)
    F = cache["_head"]["c18"].reshape(-1, 128)
    nc = orc.num_classes
    y = R.one_hot(lab, nc).reshape(-1, nc)
    A = np.concatenate([F, np.ones((F.shape[0], 1))], 1)
    w = np.where(lab.reshape(-1) != 0, atom_weight, 1.0)[:, None]
    G = A.T @ (A * w)
    G = G + ridge * np.trace(G) / 129.0 * np.eye(129)
    W = sharp * np.linalg.solve(G, A.T @ ((y - 1.0 / nc) * w))
    t = (lab != 0).astype(np.float64).reshape(-1, 1)
    Wg = sharp_sig * np.linalg.solve(G, A.T @ ((2 * t - 1) * w))
    W, Wg = (a.astype(np.float32).astype(np.float64) for a in (W, Wg))     # what an fp32 engine can hold
    orc.P["soft/kernel"] = W[:128].reshape(1, 1, 1, 128, nc)
    orc.P["soft/bias"] = W[128].copy()
    orc.P["sig/kernel"] = Wg[:128].reshape(1, 1, 1, 128, 1)
    orc.P["sig/bias"] = Wg[128].copy()
    return orc


def metric_counts(lab, soft, margin=0.0):
    """Integer counts behind r_m / p_m / wr_m (unet/unet.py:159-193) in fp64 + how many probabilities lie within
    `margin` of the rounding threshold 0.5 (those may legitimately round the other way in fp32)."""
    nc = soft.shape[-1]
    y = R.one_hot(lab, nc)
    tp = np.round(np.clip(y * soft, 0, 1)).sum()
    predicted = np.round(np.clip(soft, 0, 1)).sum()
    w = np.ones(nc); w[0] = 0.0
    wr_tp = np.round(np.clip(w * y * soft, 0, 1)).sum()
    wr_possible = np.round(np.clip(w * y, 0, 1)).sum()
    near = int((np.abs(soft - 0.5) <= margin).sum())
    return {"tp": tp, "predicted": predicted, "wr_tp": wr_tp, "wr_possible": wr_possible,
            "voxels": float(lab.size)}, near
