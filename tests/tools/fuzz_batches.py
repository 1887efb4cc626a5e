"""Every batch size 1..max_batch on ONE handle (guard bytes on) against a handle created for exactly that batch: the same
metrics and the same gradients bit for bit -- the plans are functions of the batch, not of max_batch, so nothing may depend on
what the handle was sized for -- and no guard byte written.  No oracle: engine against engine, which makes the sweep cheap
enough to cover max_batch = 32 at 32^3, 8 at 64^3 and 64 at 16^3 (tests/tools/fuzz_steps.py checks values against the oracle at
smaller max_batch).  Found by this class of check in round 6: DESIGN 11.10.

    python tests/tools/fuzz_batches.py [d=32] [max_batch=32] [C=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ICSG3D_DEBUG_CANARY"] = "1"


def main():
    from icsg3d_amd.engine import UnetEngine, VaeEngine
    from icsg3d_amd.synthetic import glorot_params, synthetic_batch, unet_param_shapes, vae_param_shapes
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    maxB = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    X, lab, cond = synthetic_batch(maxB, d, C, seed=0, noise=1e-3)
    eps = np.random.default_rng(2).standard_normal((maxB, 256)).astype(np.float32)
    Pu, Pv = glorot_params(unet_param_shapes(C, 95), 1), glorot_params(vae_param_shapes(C, d=d), 3)

    def handles(mb):
        ue = UnetEngine(in_channels=C, d=d, max_batch=mb, lr=1e-4); ue.set_weights(Pu)
        ve = VaeEngine(ue, in_channels=C, d=d, max_batch=mb, lr=5e-4); ve.set_weights(Pv)
        return ue, ve

    def step(ue, ve, b):
        ue.set_weights(Pu); ve.set_weights(Pv); ue.reset_optimizer(); ve.reset_optimizer()
        for e in (ue, ve):          # the moving statistics too: test_on_batch / predict read them
            for n, s, tr in e.tensor_infos():
                if n.endswith("/moving_mean"):
                    e.set_tensor(n, np.zeros(s, np.float32))
                elif n.endswith("/moving_var"):
                    e.set_tensor(n, np.ones(s, np.float32))
        mv = ve.train_step(X[:b], cond[:b], eps[:b])
        gv = {n: ve.get_grad(n, s) for n, s, tr in ve.tensor_infos() if tr}
        mu = ue.train_step(X[:b], lab[:b])
        gu = {n: ue.get_grad(n, s) for n, s, tr in ue.tensor_infos() if tr}
        tu = ue.test_step(X[:b], lab[:b]); tv = ve.test_step(X[:b], cond[:b], eps[:b])
        sp, mk = ue.predict_labels(X[:b])
        return mu, mv, tu, tv, gu, gv, sp, mk

    big = handles(maxB)
    nbad, t0 = 0, time.time()
    order = list(range(1, maxB + 1))
    np.random.default_rng(7).shuffle(order)            # a small batch after a large one and the reverse
    for b in order:
        got = step(*big, b)
        ue, ve = handles(b)
        ref = step(ue, ve, b)
        dirty = ue.check_canaries()[0] + ve.check_canaries()[0]
        ve.close(); ue.close()
        same_m = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(got[:4], ref[:4]))
        diff_g = [n for gg, rr in ((got[4], ref[4]), (got[5], ref[5])) for n in gg if not np.array_equal(gg[n], rr[n])]
        same_l = np.array_equal(got[6], ref[6]) and np.array_equal(got[7], ref[7])
        bad = (not same_m) or bool(diff_g) or (not same_l) or dirty
        nbad += bool(bad)
        print("  %s d=%d C=%d B=%2d on max_batch %d: metrics %s, gradients %s, labels %s%s" % (
            "FAIL" if bad else "ok  ", d, C, b, maxB, "equal" if same_m else "DIFFER", "equal" if not diff_g else "DIFFER %s" % diff_g[:6],
            "equal" if same_l else "DIFFER", ", %d guards of the exact-size handles written" % dirty if dirty else ""), flush=True)
    dirty = big[0].check_canaries(), big[1].check_canaries()
    if dirty[0][0] or dirty[1][0]:
        nbad += 1
        print("  FAIL guards of the max_batch handles: %s %s" % (dirty[0][1][:300], dirty[1][1][:300]))
    big[1].close(); big[0].close()
    print("fuzz_batches: d=%d C=%d max_batch=%d: %d batch sizes, %d bad (%.0f s)" % (d, C, maxB, maxB, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
