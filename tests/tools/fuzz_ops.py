"""Randomised sweep of the single-op entry points (ics_op_conv3d_forward / _backward) over shapes the fixed list of
tests/test_gpu_conv.py does not hold: any batch 1..9, S in {1, 2, 4, 8, 16, 32}, channel counts from the networks' own
(1, 4, 11, 16, 32, 44, 64, 95, 96, 128, 192, 256, 384, 512) and odd ones, k in {1, 3}.  Checker: torch conv3d in fp64
(oracle/torch_ref.py's layouts).  Tensor-relative tolerance 1e-5 forward / backward-data, 2e-5 backward-weight at
>= 100 k rows (fp32 accumulation over M rows).  Exit code 1 on any miss.

    python tests/tools/fuzz_ops.py [cases=60] [seed=0]
Reference: Keras Conv3D "same", /root/reference/unet/unet.py:272-336, vae/lattice_vae.py:160-230."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import torch_ref as T          # noqa: E402

CH = [1, 2, 3, 4, 8, 11, 16, 24, 32, 44, 48, 64, 95, 96, 128, 160, 192, 256, 384, 512]


def main():
    import torch
    import torch.nn.functional as F
    from icsg3d_amd.engine import conv3d_backward, conv3d_forward
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.set_num_threads(min(32, os.cpu_count() or 32))
    nbad, t0 = 0, time.time()
    for i in range(n):
        while True:
            S = int(rng.choice([1, 2, 4, 4, 8, 8, 16, 16, 32]))
            B = int(rng.integers(1, 10))
            Cin, Cout = int(rng.choice(CH)), int(rng.choice(CH))
            k = int(rng.choice([3, 3, 3, 1]))
            work = 2.0 * B * S ** 3 * k ** 3 * Cin * Cout
            if work <= 3e11 and B * S ** 3 * max(Cin, Cout) <= 1 << 27:
                break
        x = rng.standard_normal((B, S, S, S, Cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, k, Cin, Cout)) / np.sqrt(k ** 3 * Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        dy = rng.standard_normal((B, S, S, S, Cout)).astype(np.float32)
        xt = T.to_t(x, torch.float64).requires_grad_(True)
        wt = T.kernel_t(w, torch.float64).requires_grad_(True)
        yt = F.conv3d(xt, wt, torch.as_tensor(b, dtype=torch.float64), padding=k // 2)
        yt.backward(T.to_t(dy, torch.float64))
        y_ref, dx_ref, dw_ref = T.to_n(yt.detach()), T.to_n(xt.grad), T.kernel_grad_n(wt.grad)
        tag = "B%d S%d %d->%d k%d" % (B, S, Cin, Cout, k)
        try:
            y = conv3d_forward(x, w, b)
            dx, dw = conv3d_backward(x, w, dy)
        except Exception as e:            # an entry point that refuses a shape says so; anything else is a miss
            print("  FAIL %-26s %s" % (tag, str(e)[:200]), flush=True)
            nbad += 1
            continue
        e = [float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-30)) for a, r in ((y, y_ref), (dx, dx_ref), (dw, dw_ref))]
        tol_w = 2e-5 if B * S ** 3 >= 100000 else 1e-5
        bad = e[0] > 1e-5 or e[1] > 1e-5 or e[2] > tol_w
        nbad += bad
        print("  %s %-26s fwd %.1e  dgrad %.1e  wgrad %.1e" % ("FAIL" if bad else "ok  ", tag, *e), flush=True)
    print("fuzz_ops: %d shapes, %d outside the bounds (%.0f s)" % (n, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
