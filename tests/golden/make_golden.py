"""Generates tests/golden/*.npz in THIS container.  The reference's own Keras/TF path cannot be
imported, so the expected values come from oracle/torch_ref.py -- an independent
torch-CPU (F.conv3d / F.batch_norm / autograd, fp64) implementation of the same graphs -- NOT from
the numpy oracle the fixtures are used to pin.  Fixtures are data only: seeded inputs (regenerated
from the committed seed) plus expected metrics, per-tensor gradient checksums and output samples.

    PYTHONPATH=. python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import numpy_ref as R      # only for the seeded inputs/weights (init order = the spec)
from oracle import torch_ref as T

HERE = os.path.dirname(os.path.abspath(__file__))


def checks(g):
    """size-independent per-tensor summaries: sum, abs-sum, and 8 strided samples"""
    f = np.asarray(g, np.float64).ravel()
    idx = np.linspace(0, f.size - 1, 8).astype(int)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])


def main():
    B, d, C = 2, 16, 1
    X, lab, cond = R.synthetic_batch(B, d, C, seed=0, dtype=np.float64)
    X = X + 1e-3 * np.random.default_rng(5).uniform(size=X.shape)
    out = {"B": B, "d": d, "C": C}
    for ties in ("tf_cpu", "first"):
        shapes = R.unet_param_shapes(C, 95)
        P, S = R.init_params(shapes, 1), R.init_bn_state(shapes)
        m, grads, stats, soft, sig = T.unet_step_grads(P, S, X, lab, ties=ties)
        out["unet_%s_metrics" % ties] = m
        for k, g in grads.items():
            out["unet_%s_grad__%s" % (ties, k.replace("/", "__"))] = checks(g)
        for k, (mean, var, n) in stats.items():
            out["unet_%s_bnmean__%s" % (ties, k)] = mean
            out["unet_%s_bnvar__%s" % (ties, k)] = var
        out["unet_%s_soft_sample" % ties] = soft[0, ::5, ::5, ::5, ::7]
        out["unet_%s_sig_sample" % ties] = sig[0, ::5, ::5, ::5, 0]
    # VAE (perceptual U-Net = the same seed-1 U-Net)
    shapes = R.unet_param_shapes(C, 95)
    Pu, Su = R.init_params(shapes, 1), R.init_bn_state(shapes)
    vs = R.vae_param_shapes(C, 10, (16, 32, 64, 128), 256, d)
    Pv, Sv = R.init_params(vs, 3), R.init_bn_state(vs)
    eps = np.random.default_rng(2).standard_normal((B, 256))
    for training in (True, False):
        m, grads, stats, recon, zm, zlv = T.vae_step_grads(Pv, Sv, Pu, Su, X, cond.astype(np.float64), eps, d=d,
                                                           training=training)
        tag = "train" if training else "eval"
        out["vae_%s_metrics" % tag] = m
        out["vae_%s_recon_sample" % tag] = recon[0, ::5, ::5, ::5, 0]
        out["vae_%s_zmean" % tag] = zm
        for k, g in grads.items():
            out["vae_grad__%s" % k.replace("/", "__")] = checks(g)
    np.savez_compressed(os.path.join(HERE, "unet_vae_d16_b2.npz"), **out)
    print("wrote", os.path.join(HERE, "unet_vae_d16_b2.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
