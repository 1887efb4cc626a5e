"""Generates tests/golden/loss_golden.npz in THIS container by running the REFERENCE's own loss / metric / sampling
function bodies on seeded inputs.

  ** reference formulas, stand-in backend **  /root/reference/unet/unet.py and vae/lattice_vae.py cannot be imported
  (`keras`, `tensorflow` absent).  Their loss / metric functions are ~40 lines of `K.*` calls.  The FunctionDefs are
  pulled out of the two files with `ast`, compiled and executed with a numpy namespace standing where `keras.backend`
  stands (`K` below: sum, mean, round, clip, log, exp, square, epsilon, variable, flatten, batch_flatten, shape,
  int_shape, random_normal) plus `mse` (keras.losses.mean_squared_error) and a `Model` stand-in for the tap sub-model.
  It is the reference's code that runs -- none of its text is written to the repo; the fixture is data (inputs and the
  values the reference's formulas give).  What this pins: the ORDER and SHAPE of operations the reference spells out
  (renormalise-then-clip, scalar weight, axis of every mean / sum, where K.epsilon() enters, round-half-even at 0.5,
  per-sample PM / KLD then batch mean, global MSE mean).  What it does not pin: the semantics of each `K.*` call --
  those are the shim's (numpy's), checked against Keras 2.3.1's documented behaviour by reading only.

    python tests/golden/make_loss_golden.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_UNET = "/root/reference/unet/unet.py"
REF_VAE = "/root/reference/vae/lattice_vae.py"


class NumpyBackend:
    """The subset of keras.backend (2.3.1, TF backend) the extracted functions call, in numpy.
    K.round = tf.round = round-half-to-even = np.round.  K.epsilon() = 1e-7.  K.variable(v) = the value.
    K.mean / K.sum with axis=None reduce over everything; K.batch_flatten keeps axis 0."""

    def __init__(self, dtype, eps_source=None):
        self.dtype = dtype
        self._eps_source = eps_source

    def epsilon(self):
        return 1e-7

    def variable(self, v):
        return np.asarray(v, self.dtype)

    def sum(self, x, axis=None, keepdims=False):
        return np.sum(x, axis=axis, keepdims=keepdims)

    def mean(self, x, axis=None, keepdims=False):
        return np.mean(x, axis=axis, keepdims=keepdims)

    def round(self, x):
        return np.round(x)

    def clip(self, x, lo, hi):
        return np.clip(x, lo, hi)

    def log(self, x):
        return np.log(x)

    def exp(self, x):
        return np.exp(x)

    def square(self, x):
        return np.square(x)

    def flatten(self, x):
        return np.reshape(x, (-1,))

    def batch_flatten(self, x):
        return np.reshape(x, (x.shape[0], -1))

    def shape(self, x):
        return x.shape

    def int_shape(self, x):
        return tuple(x.shape)

    def random_normal(self, shape):
        e = self._eps_source
        assert tuple(shape) == e.shape
        return e


def keras_mse(y_true, y_pred):
    """keras.losses.mean_squared_error: K.mean(K.square(y_pred - y_true), axis=-1)."""
    return np.mean(np.square(y_pred - y_true), axis=-1)


def _functions(path, names, klass=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if klass is not None:
        body = [n for n in body if isinstance(n, ast.ClassDef) and n.name == klass][0].body
    fns = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(f.name for f in fns) == sorted(names), [f.name for f in fns]
    return fns


def unet_namespace(K):
    """r_m, wr_m, p_m, f1_m, weighted_categorical_crossentropy of unet/unet.py:159-221, executing."""
    fns = _functions(REF_UNET, ["r_m", "wr_m", "p_m", "f1_m", "weighted_categorical_crossentropy"])
    ns = {"K": K, "np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), REF_UNET, "exec"), ns)
    return ns


class _TapModel:
    """Stands where `Model(self.pm.input, outputs)` stands in perceptual_loss (lattice_vae.py:260-261): called on a
    tensor it returns the list of tap tensors.  The taps here are fixed seeded linear maps of the input -- the arithmetic
    AFTER the taps (flatten, square, per-sample mean, weights, sum) is what the fixture pins."""

    def __init__(self, inp, outputs):
        self.maps = outputs

    def __call__(self, x):
        B = x.shape[0]
        xf = np.reshape(x, (B, -1))
        return [np.reshape(xf @ m, (B, 2, 2, -1)) for m in self.maps]


class _Layer:
    def __init__(self, m):
        self.output = m


class _Pm:
    input = None

    def __init__(self, maps):
        self._maps = maps

    def get_layer(self, name):
        return _Layer(self._maps[name])


def vae_namespace(K):
    """sampling (lattice_vae.py:53-66) and the methods mse_loss, kld_loss, _vae_dfc_loss, perceptual_loss
    (:232-270) bound to a bare object carrying the attributes they read."""
    mod_fns = _functions(REF_VAE, ["sampling"])
    meth = _functions(REF_VAE, ["mse_loss", "kld_loss", "_vae_dfc_loss", "perceptual_loss"], klass="LatticeDFCVAE")
    holder = ast.ClassDef(name="RefVae", bases=[], keywords=[], body=meth, decorator_list=[])
    module = ast.Module(body=mod_fns + [holder], type_ignores=[])
    ast.fix_missing_locations(module)
    ns = {"K": K, "np": np, "mse": keras_mse, "Model": _TapModel}
    exec(compile(module, REF_VAE, "exec"), ns)
    return ns


def unet_cases(rng):
    """(labels, p) pairs: (B, 4, 4, 4) uint8 class ids and (B, 4, 4, 4, 95) probabilities."""
    B, S, NC = 3, 4, 95
    lab = rng.integers(0, NC, size=(B, S, S, S)).astype(np.uint8)
    lab[rng.uniform(size=lab.shape) < 0.6] = 0
    out = {}
    # 1. soft: a generic softmax output -- no probability reaches 0.5 (the Glorot regime: f1 = wr = 0)
    z = rng.normal(size=(B, S, S, S, NC))
    p = np.exp(z); p /= p.sum(-1, keepdims=True)
    out["generic"] = (lab, p)
    # 2. confident: logit of the true class raised on 70 % of voxels, of a wrong class on 10 %
    z = rng.normal(size=(B, S, S, S, NC))
    u = rng.uniform(size=lab.shape)
    wrong = (lab.astype(np.int64) + 1 + rng.integers(0, NC - 1, size=lab.shape)) % NC
    idx = np.where(u < 0.7, lab, wrong)
    boost = np.where(u < 0.8, 9.0, 0.0)
    np.put_along_axis(z, idx[..., None].astype(np.int64), np.take_along_axis(z, idx[..., None].astype(np.int64), -1)
                      + boost[..., None], -1)
    p = np.exp(z - z.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    out["confident"] = (lab, p)
    # 3. saturated / clipped: exact one-hot rows (true class prob 1 -> clipped at 1 - 1e-7; others 0 -> clipped at 1e-7),
    #    rows with the true class at exactly 0, and un-normalised rows (sum != 1: the renormalisation matters)
    p = np.zeros((B, S, S, S, NC))
    np.put_along_axis(p, lab[..., None].astype(np.int64), 1.0, -1)
    miss = rng.uniform(size=lab.shape) < 0.25
    pm = np.zeros_like(p)
    np.put_along_axis(pm, wrong[..., None], 1.0, -1)
    p = np.where(miss[..., None], pm, p)
    p[0] *= 3.0                                                # sums to 3
    out["saturated"] = (lab, p)
    # 4. halves: true-class probability exactly 0.5 (rounds to 0: half-to-even), 0.5 + 2^-20 (rounds to 1), 1.5 after
    #    a scale of 3 (y*p clipped to 1 -> rounds to 1) -- the K.round / K.clip corners of r_m / p_m
    p = np.full((B, S, S, S, NC), 0.5 / (NC - 1))
    np.put_along_axis(p, lab[..., None].astype(np.int64), 0.5, -1)
    half = p.copy()
    up = rng.uniform(size=lab.shape) < 0.5
    bump = np.zeros_like(p)
    np.put_along_axis(bump, lab[..., None].astype(np.int64), 2.0 ** -20, -1)
    half = np.where(up[..., None], half + bump, half)
    half[2] *= 3.0
    out["halves"] = (lab, half)
    return out


def main():
    fx = {}
    NC = 95
    for dt in (np.float64, np.float32):
        K = NumpyBackend(dt)
        U = unet_namespace(K)
        tag = "f64" if dt is np.float64 else "f32"
        for name, (lab, p) in unet_cases(np.random.default_rng(23)).items():
            y = (lab[..., None] == np.arange(NC)).astype(dt)
            pp = p.astype(dt)
            if dt is np.float64:
                fx["unet/%s/labels" % name] = lab
                fx["unet/%s/p" % name] = p
            for w, wn in ((95, "w95"), (np.linspace(0.5, 2.0, NC), "wvec")):
                loss = U["weighted_categorical_crossentropy"](w)(y, pp.copy())
                fx["unet/%s/%s/wcce_%s" % (name, tag, wn)] = np.asarray(loss, np.float64)
            for fn in ("r_m", "p_m", "f1_m", "wr_m"):
                fx["unet/%s/%s/%s" % (name, tag, fn)] = np.float64(U[fn](y, pp))
            # the integer counts behind the ratios (the reference never returns them; same K calls)
            fx["unet/%s/%s/counts" % (name, tag)] = np.array([
                K.sum(K.round(K.clip(y * pp, 0, 1))), K.sum(K.round(K.clip(y, 0, 1))),
                K.sum(K.round(K.clip(pp, 0, 1)))], np.float64)

        # --- VAE side (one seeded input set, cast per dtype)
        B, L = 4, 256
        vr = np.random.default_rng(29)
        zm = vr.normal(size=(B, L)).astype(dt)
        zlv = (0.5 * vr.normal(size=(B, L))).astype(dt)
        eps = vr.normal(size=(B, L)).astype(dt)
        K = NumpyBackend(dt, eps_source=eps)
        V = vae_namespace(K)
        z = V["sampling"]([zm, zlv])
        x = vr.uniform(size=(B, 4, 4, 4, 1)).astype(dt)
        rec = (x + 0.1 * vr.normal(size=x.shape)).astype(dt)
        names = ["re_lu_2", "re_lu_4", "re_lu_6", "re_lu_8"]
        maps = {n: vr.normal(size=(64, 8 * (i + 1))).astype(dt) for i, n in enumerate(names)}
        weights = [1.0, 0.5, 2.0, 0.25]
        vae = V["RefVae"]()
        vae.z_mean, vae.z_log_var = zm, zlv
        vae.pm, vae.pm_layers, vae.pm_layer_weights = _Pm(maps), names, weights
        mse_v = vae.mse_loss(x, rec)
        kld_v = vae.kld_loss()
        pm_v = vae.perceptual_loss(x, rec)
        total = vae._vae_dfc_loss(0.5, 3e-4)(x, rec)
        if dt is np.float64:
            fx.update({"vae/zm": zm, "vae/zlv": zlv, "vae/eps": eps, "vae/x": x, "vae/rec": rec,
                       "vae/pm_weights": np.array(weights)})
            for n in names:
                fx["vae/map/" + n] = maps[n]
        fx["vae/%s/z" % tag] = np.asarray(z, np.float64)
        fx["vae/%s/mse" % tag] = np.float64(mse_v)
        fx["vae/%s/kld" % tag] = np.asarray(kld_v, np.float64)
        fx["vae/%s/pm" % tag] = np.asarray(pm_v, np.float64)
        fx["vae/%s/loss" % tag] = np.float64(total)
        assert np.ndim(mse_v) == 0 and np.shape(kld_v) == (B,) and np.shape(pm_v) == (B,) and np.ndim(total) == 0

    path = os.path.join(HERE, "loss_golden.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, len(fx), "arrays,", os.path.getsize(path), "bytes")
    for k in sorted(fx):
        if k.endswith(("f1_m", "wr_m", "counts", "loss", "mse")):
            print(" ", k, fx[k])


if __name__ == "__main__":
    main()
