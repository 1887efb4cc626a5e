"""Generates tests/golden/unet_vae_d{16,32,64}_b{2,2,1}.npz in THIS container.  The reference's own
Keras/TF path cannot be imported, so the expected values come from oracle/torch_ref.py --
an independent torch-CPU (F.conv3d / F.batch_norm / autograd, fp64) implementation of the same graphs
-- NOT from the numpy oracle the fixtures are used to pin.  Fixtures are data only: seeded inputs
(regenerated from the committed seed) plus expected metrics, BatchNorm batch statistics, per-tensor
gradient checksums and output samples.

    PYTHONPATH=. python tests/golden/make_golden.py d16      # B=2, both pool-tie rules (seconds)
    PYTHONPATH=. python tests/golden/make_golden.py d32      # B=2: the shapes bench.py times (minutes)
    PYTHONPATH=. python tests/golden/make_golden.py d64      # B=1: BASELINE configs[4] grid (~10 min)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import numpy_ref as R      # only for the seeded inputs/weights (init order = the spec)
from oracle import torch_ref as T

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {"d16": (2, 16, ("tf_cpu", "first")), "d32": (2, 32, ("tf_cpu",)), "d64": (1, 64, ("tf_cpu",))}


def checks(g):
    """size-independent per-tensor summaries: sum, abs-sum, and 8 strided samples"""
    f = np.asarray(g, np.float64).ravel()
    idx = np.linspace(0, f.size - 1, 8).astype(int)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])


def inputs(B, d, C=1):
    """the seeded batch every fixture and every test of that size uses"""
    X, lab, cond = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    return X, lab, cond.astype(np.float64), eps


def main(which):
    B, d, tie_rules = CONFIGS[which]
    C = 1
    X, lab, cond, eps = inputs(B, d, C)
    st = max(d // 3, 1)                 # sample stride: 4 points per axis
    out = {"B": B, "d": d, "C": C}
    for ties in tie_rules:
        t0 = time.time()
        shapes = R.unet_param_shapes(C, 95)
        P, S = R.init_params(shapes, 1), R.init_bn_state(shapes)
        m, grads, stats, soft, sig = T.unet_step_grads(P, S, X, lab, ties=ties)
        out["unet_%s_metrics" % ties] = m
        for k, g in grads.items():
            out["unet_%s_grad__%s" % (ties, k.replace("/", "__"))] = checks(g)
        for k, (mean, var, n) in stats.items():
            out["unet_%s_bnmean__%s" % (ties, k)] = mean
            out["unet_%s_bnvar__%s" % (ties, k)] = var
        out["unet_%s_soft_sample" % ties] = soft[0, ::st, ::st, ::st, ::7]
        out["unet_%s_sig_sample" % ties] = sig[0, ::st, ::st, ::st, 0]
        print("unet", ties, "%.0f s" % (time.time() - t0), m, flush=True)
        del grads, soft, sig
    # VAE (perceptual U-Net = the same seed-1 U-Net)
    shapes = R.unet_param_shapes(C, 95)
    Pu, Su = R.init_params(shapes, 1), R.init_bn_state(shapes)
    vs = R.vae_param_shapes(C, 10, (16, 32, 64, 128), 256, d)
    Pv, Sv = R.init_params(vs, 3), R.init_bn_state(vs)
    for training in (True, False):
        t0 = time.time()
        m, grads, stats, recon, zm, zlv = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond, eps, d=d, training=training)
        tag = "train" if training else "eval"
        out["vae_%s_metrics" % tag] = m
        out["vae_%s_recon_sample" % tag] = recon[0, ::st, ::st, ::st, 0]
        out["vae_%s_zmean" % tag] = zm
        out["vae_%s_zlogvar" % tag] = zlv
        for k, g in grads.items():
            out["vae_grad__%s" % k.replace("/", "__")] = checks(g)
        for k, (mean, var, n) in stats.items():
            out["vae_%s_bnmean__%s" % (tag, k)] = mean
            out["vae_%s_bnvar__%s" % (tag, k)] = var
        print("vae", tag, "%.0f s" % (time.time() - t0), m, flush=True)
    path = os.path.join(HERE, "unet_vae_%s_b%d.npz" % (which, B))
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["d16"]):
        main(w)
