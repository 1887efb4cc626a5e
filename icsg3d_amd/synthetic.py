"""Synthetic workload of SURVEY.md 8(d): parameter shapes, Glorot initialisation and Gaussian-blob
grids.  Product-side (bench.py, the training scripts' --synthetic mode); the oracle keeps its own
copy so that tests can check the two agree."""
from __future__ import annotations

import numpy as np

# name, Cin (None = input channels), Cout      (/root/reference/unet/unet.py:276-336)
UNET_CONVS = [
    ("c1", None, 32), ("c2", 32, 64), ("c3", 64, 64), ("c4", 64, 128), ("c5", 128, 128),
    ("c6", 128, 256), ("c9", 256, 512), ("c10", 512, 512), ("c13", 768, 512), ("c14", 512, 256),
    ("c15", 384, 256), ("c16", 256, 128), ("c17", 192, 128), ("c18", 128, 128),
]


def unet_param_shapes(in_ch=1, num_classes=95):
    shapes = []
    for name, cin, cout in UNET_CONVS:
        cin = in_ch if cin is None else cin
        shapes += [(name + "/kernel", (3, 3, 3, cin, cout)), (name + "/bias", (cout,)),
                   (name + "/gamma", (cout,)), (name + "/beta", (cout,))]
    shapes += [("soft/kernel", (1, 1, 1, 128, num_classes)), ("soft/bias", (num_classes,)),
               ("sig/kernel", (1, 1, 1, 128, 1)), ("sig/bias", (1,))]
    return shapes


def vae_param_shapes(in_ch=1, cond=10, filters=(16, 32, 64, 128), latent=256, d=32):
    sh = []
    cin = in_ch + in_ch * cond            # K.tile quirk (vae/lattice_vae.py:167-169)
    for i, f in enumerate(filters):
        n = "e%d" % i
        sh += [(n + "/kernel", (3, 3, 3, cin, f)), (n + "/bias", (f,)), (n + "/gamma", (f,)), (n + "/beta", (f,))]
        cin = f
    sh += [("e4/kernel", (3, 3, 3, cin, 4)), ("e4/bias", (4,))]
    flat = (d // 16) ** 3 * 4
    sh += [("enc_dense/kernel", (flat, latent)), ("enc_dense/bias", (latent,)),
           ("z_mean/kernel", (latent, latent)), ("z_mean/bias", (latent,)),
           ("z_log_var/kernel", (latent, latent)), ("z_log_var/bias", (latent,))]
    seed = (d // 8) ** 3 * 4
    sh += [("dec_dense/kernel", (latent + cond, seed)), ("dec_dense/bias", (seed,))]
    cin = 4
    for i, f in enumerate(filters[::-1]):
        n = "d%d" % i
        sh += [(n + "/kernel", (3, 3, 3, cin, f)), (n + "/bias", (f,)), (n + "/gamma", (f,)), (n + "/beta", (f,))]
        cin = f
    sh += [("dout/kernel", (3, 3, 3, cin, in_ch)), ("dout/bias", (in_ch,)),
           ("dout/gamma", (in_ch,)), ("dout/beta", (in_ch,))]
    return sh


def bn_state_defaults(shapes):
    """Keras BatchNormalization initial moving statistics (mean 0, variance 1) for every layer that has a gamma."""
    out = {}
    for name, shp in shapes:
        if name.endswith("/gamma"):
            out[name[:-5] + "moving_mean"] = np.zeros(shp, np.float32)
            out[name[:-5] + "moving_var"] = np.ones(shp, np.float32)
    return out


def glorot_params(shapes, seed):
    """Glorot-uniform kernels from PCG64(seed) in list order, zero biases, BN gamma 1 / beta 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in shapes:
        if name.endswith("/kernel"):
            if len(shp) == 5:
                rf = shp[0] * shp[1] * shp[2]
                fan_in, fan_out = rf * shp[3], rf * shp[4]
            else:
                fan_in, fan_out = shp
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
        elif name.endswith("/gamma"):
            out[name] = np.ones(shp, np.float32)
        else:
            out[name] = np.zeros(shp, np.float32)
    return out


def synthetic_batch(B, d=32, C=1, seed=0, noise=0.0):
    """Gaussian-blob densities (mimics utils.density_matrix, /root/reference/utils.py:135-143),
    uint8 species labels, one-hot condition (i mod 10).  Returns (X float32, labels uint8, cond)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    X = np.zeros((B, d, d, d, C), np.float32)
    labels = np.zeros((B, d, d, d), np.uint8)
    for b in range(B):
        dens = np.zeros((d, d, d))
        for k in range(int(rng.integers(2, 9))):
            c = rng.uniform(0, d, 3)
            sig = rng.uniform(1.5, 4.0) * d / 32.0
            amp = rng.uniform(0.5, 3.0)
            r2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
            dens += amp * np.exp(-r2 / (2 * sig * sig))
            labels[b][r2 <= sig * sig] = 1 + (k * 13) % 94
        X[b, ..., 0] = np.maximum(dens, 0)
        if C > 1:
            X[b, ..., 1:4] = (np.stack([zz, yy, xx], -1) / float(d))[..., :C - 1]
    if noise:
        X = X + np.float32(noise) * np.random.default_rng(seed + 1000).uniform(size=X.shape).astype(np.float32)
    cond = np.eye(10, dtype=np.float32)[np.arange(B) % 10]
    return X, labels, cond
