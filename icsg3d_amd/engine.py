"""Thin Python handles over the C ABI (include/icsg3d.h): numpy in / numpy out.

These are what the reference-shaped classes (icsg3d_amd/unet/unet.py, icsg3d_amd/vae/lattice_vae.py)
call where the reference calls Keras.  All compute happens in libicsg3d_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class _Net:
    """Common handle plumbing: named tensors, gradients, profiling, data-parallel comm."""

    def __init__(self):
        self._h = L._H(None)
        self._lib = L.load()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ics_net_destroy(self._h)
            self._h = L._H(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tensors
    def tensor_infos(self):
        n = C.c_int(0)
        L.check(self._lib.ics_net_num_tensors(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.c_char_p()
            nd = C.c_int(0)
            dims = (C.c_int64 * 5)()
            tr = C.c_int(0)
            L.check(self._lib.ics_net_tensor_info(self._h, i, C.byref(name), C.byref(nd), dims, C.byref(tr)))
            out.append((name.value.decode(), tuple(int(dims[k]) for k in range(nd.value)), bool(tr.value)))
        return out

    def get_tensor(self, name, shape=None):
        if shape is None:
            shape = dict((n, s) for n, s, _ in self.tensor_infos())[name]
        a = np.empty(shape, np.float32)
        L.check(self._lib.ics_net_get_tensor(self._h, name.encode(), L.fptr(a), a.size))
        return a

    def set_tensor(self, name, value):
        a = _f32(value)
        L.check(self._lib.ics_net_set_tensor(self._h, name.encode(), L.fptr(a), a.size))

    def get_grad(self, name, shape):
        a = np.empty(shape, np.float32)
        L.check(self._lib.ics_net_get_grad(self._h, name.encode(), L.fptr(a), a.size))
        return a

    def get_activation(self, layer, shape):
        a = np.empty(shape, np.float32)
        L.check(self._lib.ics_net_get_activation(self._h, layer.encode(), L.fptr(a), a.size))
        return a

    def get_bn_affine(self, layer, channels):
        sc = np.empty(channels, np.float32)
        sh = np.empty(channels, np.float32)
        L.check(self._lib.ics_net_get_bn_affine(self._h, layer.encode(), L.fptr(sc), L.fptr(sh), channels))
        return sc, sh

    def get_weights(self):
        return {n: self.get_tensor(n, s) for n, s, _ in self.tensor_infos()}

    def set_weights(self, weights):
        known = dict((n, s) for n, s, _ in self.tensor_infos())
        for k, v in weights.items():
            if k not in known:
                raise KeyError("unknown tensor %r" % k)
            if tuple(np.shape(v)) != known[k]:
                raise ValueError("shape mismatch for %s: %s vs %s" % (k, np.shape(v), known[k]))
            self.set_tensor(k, v)

    def check_canaries(self):
        """(dirty count, description): diagnosis aid, needs ICSG3D_DEBUG_CANARY=1 when the engine was created."""
        n = C.c_int(0)
        L.check(self._lib.ics_net_check_canaries(self._h, C.byref(n)))
        return n.value, (self._lib.ics_last_error().decode() if n.value else "")

    def set_lr(self, lr):
        L.check(self._lib.ics_net_set_lr(self._h, float(lr)))

    def reset_optimizer(self):
        L.check(self._lib.ics_net_reset_optimizer(self._h))

    def num_params(self):
        n = C.c_size_t(0)
        L.check(self._lib.ics_net_num_params(self._h, C.byref(n)))
        return int(n.value)

    def get_optimizer_state(self):
        """(m, v, t): Adam moments over the flat parameter buffer and the step count."""
        n = self.num_params()
        m, v, t = np.empty(n, np.float32), np.empty(n, np.float32), C.c_int(0)
        L.check(self._lib.ics_net_get_optimizer_state(self._h, L.fptr(m), L.fptr(v), n, C.byref(t)))
        return m, v, int(t.value)

    def set_optimizer_state(self, m, v, t):
        m, v = _f32(m), _f32(v)
        L.check(self._lib.ics_net_set_optimizer_state(self._h, L.fptr(m), L.fptr(v), m.size, int(t)))

    def sync(self):
        L.check(self._lib.ics_net_sync(self._h))

    def share_stream(self, other):
        """enqueue on `other`'s stream from now on (joint training of two engines on one GPU: steps alternate in order)"""
        L.check(self._lib.ics_net_share_stream(self._h, other._h))
        self._stream_owner = other          # keep it alive

    def timer_start(self):
        L.check(self._lib.ics_net_timer_start(self._h))

    def timer_stop(self):
        """milliseconds of device time on the engine's stream since timer_start (waits for the work in between)"""
        ms = C.c_double(0)
        L.check(self._lib.ics_net_timer_stop(self._h, C.byref(ms)))
        return ms.value

    # ---- profiling (HIP events on the engine's stream)
    def graph_probe(self, iters=20):
        """Measurement aid: the resident train step eagerly and as a replayed hipGraph -> (eager ms, graph ms, graph nodes)."""
        a, b, n = C.c_double(0), C.c_double(0), C.c_int(0)
        L.check(self._lib.ics_net_graph_probe(self._h, int(iters), C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def profile_enable(self, on=True):
        L.check(self._lib.ics_net_profile_enable(self._h, 1 if on else 0))

    def profile_filter(self, prefix=""):
        """events only around launch sites whose label starts with `prefix` ("" = all)"""
        L.check(self._lib.ics_net_profile_filter(self._h, prefix.encode() if prefix else None))

    def profile_rows(self):
        n = C.c_int(0)
        L.check(self._lib.ics_net_profile_count(self._h, C.byref(n)))
        rows = []
        for i in range(n.value):
            label = C.c_char_p()
            cnt = C.c_int64(0)
            ms, fl, by = C.c_double(0), C.c_double(0), C.c_double(0)
            L.check(self._lib.ics_net_profile_row(self._h, i, C.byref(label), C.byref(cnt), C.byref(ms),
                                                  C.byref(fl), C.byref(by)))
            rows.append({"label": label.value.decode(), "launches": cnt.value, "ms": ms.value,
                         "flop": fl.value, "bytes": by.value})
        return rows

    # ---- data parallel
    def comm_init(self, rank, nranks, uid):
        L.check(self._lib.ics_net_comm_init(self._h, int(rank), int(nranks), uid))

    def broadcast_state(self, root=0):
        """rank `root`'s parameters, BN moving statistics and Adam state overwrite every replica's."""
        L.check(self._lib.ics_net_comm_broadcast_state(self._h, int(root)))

    def set_sync_bn(self, on=True):
        L.check(self._lib.ics_net_set_sync_bn(self._h, 1 if on else 0))

    def comm_info(self):
        r, n, b = C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(self._lib.ics_net_comm_info(self._h, C.byref(r), C.byref(n), C.byref(b)))
        return {"rank": r.value, "nranks": n.value, "buckets_last_step": b.value}

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        L.check(self._lib.ics_net_comm_allreduce_max(self._h, C.byref(v)))
        return v.value


def comm_unique_id():
    buf = C.create_string_buffer(128)
    L.check(L.load().ics_comm_unique_id(buf))
    return buf.raw


class UnetEngine(_Net):
    def __init__(self, in_channels=1, num_classes=95, d=32, max_batch=32, lr=1e-6, loss_weight=0.0,
                 pool_ties="tf_cpu", bn_unbias=True, device=None, bce_from_logits=False):
        super().__init__()
        if device is not None:
            L.check(self._lib.ics_set_device(int(device)))
        self.in_channels, self.num_classes, self.d, self.max_batch = in_channels, num_classes, d, max_batch
        cfg = L.UnetConfig(in_channels, num_classes, d, max_batch, lr, loss_weight,
                           1 if pool_ties == "tf_cpu" else 0, 1 if bn_unbias else 0, 1 if bce_from_logits else 0)
        L.check(self._lib.ics_unet_create(C.byref(cfg), C.byref(self._h)))

    def _check_x(self, x):
        x = _f32(x)
        d = self.d
        if x.ndim != 5 or x.shape[1:] != (d, d, d, self.in_channels):
            raise ValueError("expected (B,%d,%d,%d,%d), got %s" % (d, d, d, self.in_channels, x.shape))
        return x

    def predict(self, x):
        x = self._check_x(x)
        B, d = x.shape[0], self.d
        soft = np.empty((B, d, d, d, self.num_classes), np.float32)
        sig = np.empty((B, d, d, d, 1), np.float32)
        for i in range(0, B, self.max_batch):
            xs = x[i:i + self.max_batch]
            L.check(self._lib.ics_unet_predict(self._h, L.fptr(xs), xs.shape[0], L.fptr(soft[i:i + self.max_batch]),
                                               L.fptr(sig[i:i + self.max_batch])))
        return soft, sig

    def predict_labels(self, x, thresh=0.8):
        x = self._check_x(x)
        B, d = x.shape[0], self.d
        sp = np.empty((B, d, d, d), np.uint8)
        mk = np.empty((B, d, d, d), np.uint8)
        for i in range(0, B, self.max_batch):
            xs = x[i:i + self.max_batch]
            L.check(self._lib.ics_unet_predict_labels(self._h, L.fptr(xs), xs.shape[0], float(thresh),
                                                      L.u8ptr(sp[i:i + self.max_batch]), L.u8ptr(mk[i:i + self.max_batch])))
        return sp, mk

    def _labels(self, labels, B):
        lab = np.ascontiguousarray(labels, dtype=np.uint8)
        if lab.shape != (B, self.d, self.d, self.d):
            raise ValueError("labels must be (B,d,d,d) uint8 class ids, got %s" % (lab.shape,))
        return lab

    def train_step(self, x, labels):
        x = self._check_x(x)
        lab = self._labels(labels, x.shape[0])
        m = np.zeros(5, np.float32)
        L.check(self._lib.ics_unet_train_step(self._h, L.fptr(x), L.u8ptr(lab), x.shape[0], L.fptr(m)))
        return m

    def test_step(self, x, labels):
        x = self._check_x(x)
        lab = self._labels(labels, x.shape[0])
        m = np.zeros(5, np.float32)
        L.check(self._lib.ics_unet_test_step(self._h, L.fptr(x), L.u8ptr(lab), x.shape[0], L.fptr(m)))
        return m

    def metric_sums(self):
        """The K.sum terms of the last step's metrics (unet/unet.py:159-193): dict(sum_lsoft, sum_lsig, tp, predicted,
        wr_tp, wr_possible, voxels)."""
        s = (C.c_double * 7)()
        L.check(self._lib.ics_unet_metric_sums(self._h, s))
        return dict(zip(("sum_lsoft", "sum_lsig", "tp", "predicted", "wr_tp", "wr_possible", "voxels"), [float(v) for v in s]))

    def upload_batch(self, x, labels):
        x = self._check_x(x)
        lab = self._labels(labels, x.shape[0])
        L.check(self._lib.ics_unet_upload_batch(self._h, L.fptr(x), L.u8ptr(lab), x.shape[0]))

    def predict_resident(self, labels_only=False, thresh=0.8):
        """model.predict on the uploaded batch, enqueued only; the outputs stay in HBM (benchmark path)."""
        L.check(self._lib.ics_unet_predict_resident(self._h, 1 if labels_only else 0, float(thresh)))

    def train_step_resident(self, want_metrics=False):
        if want_metrics:
            m = np.zeros(5, np.float32)
            L.check(self._lib.ics_unet_train_step_resident(self._h, L.fptr(m)))
            return m
        L.check(self._lib.ics_unet_train_step_resident(self._h, None))
        return None


class VaeEngine(_Net):
    def __init__(self, unet: UnetEngine, in_channels=1, cond_shape=10, latent_dim=256,
                 filters=(16, 32, 64, 128), d=32, max_batch=32, lr=5e-4, alpha=0.5, beta=3e-4,
                 pm_layer_weights=(1.0, 1.0, 1.0, 1.0), bn_unbias=True):
        super().__init__()
        self.unet = unet   # keep the perceptual engine alive
        self.in_channels, self.cond_shape, self.latent_dim, self.d, self.max_batch = (
            in_channels, cond_shape, latent_dim, d, max_batch)
        cfg = L.VaeConfig(in_channels, cond_shape, latent_dim, (C.c_int * 4)(*filters), d, max_batch, lr,
                          alpha, beta, (C.c_float * 4)(*pm_layer_weights), 1 if bn_unbias else 0)
        L.check(self._lib.ics_vae_create(C.byref(cfg), unet._h if unet is not None else None, C.byref(self._h)))

    def _args(self, x, cond, eps):
        x, cond = _f32(x), _f32(cond)
        B = x.shape[0]
        d = self.d
        if x.shape != (B, d, d, d, self.in_channels):
            raise ValueError("bad input shape %s" % (x.shape,))
        if cond.shape != (B, self.cond_shape):
            raise ValueError("bad cond shape %s" % (cond.shape,))
        eps = _f32(eps)
        if eps.shape != (B, self.latent_dim):
            raise ValueError("bad eps shape %s" % (eps.shape,))
        return x, cond, eps, B

    def encode(self, x, cond, eps):
        x, cond, eps, B = self._args(x, cond, eps)
        zm = np.empty((B, self.latent_dim), np.float32)
        zlv = np.empty_like(zm)
        z = np.empty_like(zm)
        for i in range(0, B, self.max_batch):
            s = slice(i, i + self.max_batch)
            L.check(self._lib.ics_vae_encode(self._h, L.fptr(x[s]), L.fptr(cond[s]), L.fptr(eps[s]), x[s].shape[0],
                                             L.fptr(zm[s]), L.fptr(zlv[s]), L.fptr(z[s])))
        return zm, zlv, z

    def decode(self, z, cond):
        z, cond = _f32(z), _f32(cond)
        B, d = z.shape[0], self.d
        out = np.empty((B, d, d, d, self.in_channels), np.float32)
        for i in range(0, B, self.max_batch):
            s = slice(i, i + self.max_batch)
            L.check(self._lib.ics_vae_decode(self._h, L.fptr(z[s]), L.fptr(cond[s]), z[s].shape[0], L.fptr(out[s])))
        return out

    def decode_to_labels(self, unet, z, cond, thresh=0.8, want_density=True):
        """decoder.predict -> unet.model.predict -> argmax / threshold without leaving the device
        (generate.py:204-225).  Returns dict(species u8, mask u8, density f32 | None, coord_minmax (B,3,2))."""
        z, cond = _f32(z), _f32(cond)
        B, d = z.shape[0], self.d
        mb = min(self.max_batch, unet.max_batch)
        sp = np.empty((B, d, d, d), np.uint8)
        mk = np.empty((B, d, d, d), np.uint8)
        dens = np.empty((B, d, d, d), np.float32) if want_density else None
        mm = np.zeros((B, 3, 2), np.float32)
        for i in range(0, B, mb):
            s = slice(i, i + mb)
            L.check(self._lib.ics_vae_decode_to_unet_labels(
                self._h, unet._h, L.fptr(z[s]), L.fptr(cond[s]), z[s].shape[0], float(thresh), L.u8ptr(sp[s]),
                L.u8ptr(mk[s]), L.fptr(dens[s]) if want_density else None, L.fptr(mm[s])))
        return {"species": sp, "mask": mk, "density": dens, "coord_minmax": mm}

    def decode_to_atoms(self, unet, z, cond, thresh=0.8, min_voxels=3, max_atoms=512, want_density=True,
                        want_regions=False):
        """decode_to_labels continued on the device through connected components + region statistics
        (generate.py:204-236, watershed.py:52-56,153-187): dict(species, mask, density, coord_minmax, regions | None,
        n_components (B,), n_atoms (B,), stats int32 (B,max_atoms,11), atoms [(species list, mean list)], failed (B,)).
        Every kept component is taken as convex here; icsg3d_amd.watershed.refine_atoms continues with the convexity
        test and the recursive split (needs want_regions=True)."""
        from .watershed import STAT_FIELDS, _atoms_from_stats
        z, cond = _f32(z), _f32(cond)
        B, d = z.shape[0], self.d
        mb = min(self.max_batch, unet.max_batch)
        sp = np.empty((B, d, d, d), np.uint8)
        mk = np.empty((B, d, d, d), np.uint8)
        dens = np.empty((B, d, d, d), np.float32) if want_density else None
        reg = np.empty((B, d, d, d), np.int32) if want_regions else None
        mm = np.zeros((B, 3, 2), np.float32)
        counts = np.zeros((B, 2), np.int32)
        stats = np.zeros((B, max_atoms, len(STAT_FIELDS)), np.int32)
        bounds = np.zeros((B, max_atoms, 8), np.int64) if want_regions else None    # refine_atoms' convexity shortcut
        for i in range(0, B, mb):
            s = slice(i, i + mb)
            L.check(self._lib.ics_vae_decode_to_unet_atoms(
                self._h, unet._h, L.fptr(z[s]), L.fptr(cond[s]), z[s].shape[0], float(thresh), int(min_voxels),
                int(max_atoms), L.u8ptr(sp[s]), L.u8ptr(mk[s]), L.fptr(dens[s]) if want_density else None,
                L.fptr(mm[s]), L.i32ptr(reg[s]) if want_regions else None, L.i32ptr(counts[s]), L.i32ptr(stats[s]),
                L.i64ptr(bounds[s]) if bounds is not None else None))
        failed = counts[:, 1] > max_atoms                 # more kept components than rows: sample skipped, not the batch
        counts[failed, 1] = 0
        return {"species": sp, "mask": mk, "density": dens, "coord_minmax": mm, "regions": reg,
                "n_components": counts[:, 0].copy(), "n_atoms": counts[:, 1].copy(), "stats": stats, "bounds": bounds,
                "atoms": _atoms_from_stats(counts, stats, d ** 3), "failed": failed}

    def train_step(self, x, cond, eps):
        x, cond, eps, B = self._args(x, cond, eps)
        m = np.zeros(4, np.float32)
        L.check(self._lib.ics_vae_train_step(self._h, L.fptr(x), L.fptr(cond), L.fptr(eps), B, L.fptr(m)))
        return m

    def test_step(self, x, cond, eps):
        x, cond, eps, B = self._args(x, cond, eps)
        m = np.zeros(4, np.float32)
        L.check(self._lib.ics_vae_test_step(self._h, L.fptr(x), L.fptr(cond), L.fptr(eps), B, L.fptr(m)))
        return m

    def upload_batch(self, x, cond, eps):
        x, cond, eps, B = self._args(x, cond, eps)
        L.check(self._lib.ics_vae_upload_batch(self._h, L.fptr(x), L.fptr(cond), L.fptr(eps), B))

    def train_step_resident(self, want_metrics=False):
        if want_metrics:
            m = np.zeros(4, np.float32)
            L.check(self._lib.ics_vae_train_step_resident(self._h, L.fptr(m)))
            return m
        L.check(self._lib.ics_vae_train_step_resident(self._h, None))
        return None


def conv3d_forward(x, w, bias=None, pre_act=0):
    """Single-op entry (kernel parity tests): x (B,S,S,S,Cin), w (k,k,k,Cin,Cout)."""
    lib = L.load()
    x, w = _f32(x), _f32(w)
    B, S, Cin = x.shape[0], x.shape[1], x.shape[4]
    taps, Cout = w.shape[0] ** 3, w.shape[4]
    y = np.empty((B, S, S, S, Cout), np.float32)
    b = _f32(bias) if bias is not None else None
    L.check(lib.ics_op_conv3d_forward(L.fptr(x), L.fptr(w), L.fptr(b), B, S, Cin, Cout, taps, pre_act, L.fptr(y)))
    return y


def unet_head(x, wsoft, bsoft, wsig, bsig, labels, mode=1, fused=True, loss_weight=0.0, bce_from_logits=False):
    """Single-op entry (kernel parity tests): the heads + losses + metrics on a given trunk output x (M,128).
    Returns (out (M, ncls+1) | None, metrics (5,), sums dict) -- see ics_op_unet_head."""
    lib = L.load()
    x, wsoft, bsoft, wsig, bsig = _f32(x), _f32(wsoft), _f32(bsoft), _f32(wsig), _f32(bsig)
    M, ncls = x.shape[0], wsoft.shape[1]
    lab = np.ascontiguousarray(labels, dtype=np.uint8).reshape(M)
    out = np.empty((M, ncls + 1), np.float32) if mode != 1 else None
    m = np.zeros(5, np.float32)
    s = (C.c_double * 7)()
    L.check(lib.ics_op_unet_head(L.fptr(x), L.fptr(wsoft), L.fptr(bsoft), L.fptr(wsig), L.fptr(bsig), L.u8ptr(lab), M, ncls,
                                 float(loss_weight), int(mode) | (4 if bce_from_logits else 0), 1 if fused else 0, L.fptr(out),
                                 L.fptr(m), s))
    keys = ("sum_lsoft", "sum_lsig", "tp", "predicted", "wr_tp", "wr_possible", "voxels")
    return out, m, dict(zip(keys, [float(v) for v in s]))


def conv3d_backward(x, w, dy):
    lib = L.load()
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    B, S, Cin = x.shape[0], x.shape[1], x.shape[4]
    taps, Cout = w.shape[0] ** 3, w.shape[4]
    dx = np.empty_like(x)
    dw = np.empty_like(w)
    L.check(lib.ics_op_conv3d_backward(L.fptr(x), L.fptr(w), L.fptr(dy), B, S, Cin, Cout, taps, L.fptr(dx), L.fptr(dw)))
    return dx, dw
