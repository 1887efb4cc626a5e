"""ics_op_watershed_split on host threads (round 6; /root/reference/watershed.py:95-110): no device work is involved, so the
entry point is held to oracle/watershed_ref.split_component on CPU as well -- both tie rules, label 1 (the shell opens, the
flood runs) and label 5 (eroded cores), ragged random blobs with holes, many boxes per call (threads), a box the size of a
whole 32^3 sample.  The GPU suite (tests/test_gpu_segment.py) runs the same statement against the kernel form."""
import numpy as np
import pytest

from oracle import watershed_ref as W


def _blob(rng, lo=5, hi=22):
    D, H, Wd = (int(v) for v in rng.integers(lo, hi, 3))
    zz, yy, xx = np.mgrid[:D, :H, :Wd]
    m = np.zeros((D, H, Wd), bool)
    for _ in range(int(rng.integers(2, 5))):
        c = rng.uniform(0, 1, 3) * [D, H, Wd]
        r = rng.uniform(2, 5)
        m |= ((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= r * r
    m &= rng.uniform(size=m.shape) > 0.03          # single-voxel cavities: pockets the background flood cannot reach
    return m


@pytest.mark.parametrize("tie", ["heap", "fifo"])
def test_host_split_equals_the_oracle(tie, monkeypatch):
    from icsg3d_amd.watershed import watershed_split
    monkeypatch.setenv("ICSG3D_WS_DEVICE", "0")
    rng = np.random.default_rng(3)
    boxes, cls = [], []
    for k in range(40):
        cl = 1 if k % 2 == 0 else 5
        boxes.append(np.where(_blob(rng), cl, 0).astype(np.int32)); cls.append(cl)
    boxes.append(np.zeros((3, 4, 5), np.int32)); cls.append(1)                       # empty box
    boxes.append(np.ones((4, 4, 4), np.int32)); cls.append(1)                        # full box: no background marker at all
    whole = (rng.uniform(size=(32, 32, 32)) < 0.8).astype(np.int32)                  # a whole-sample blob
    boxes.append(whole); cls.append(1)
    got = watershed_split(boxes, cls, tie=tie)
    for b, cl, g in zip(boxes, cls, got):
        assert np.array_equal(g, W.split_component(b, cl, tie=tie)), (b.shape, cl)
    monkeypatch.setenv("ICSG3D_HOST_THREADS", "1")                                   # one thread: same bits
    for a, b in zip(got, watershed_split(boxes, cls, tie=tie)):
        assert np.array_equal(a, b)


def test_host_pool_survives_concurrent_callers_and_a_fork(monkeypatch):
    """The library's host thread pool (segment.hip HostPool): created on first use, one call at a time (callers from several
    Python threads queue), re-created in a forked child (which has none of the parent's threads)."""
    import os
    import threading
    from icsg3d_amd.watershed import watershed_split
    monkeypatch.setenv("ICSG3D_WS_DEVICE", "0")
    rng = np.random.default_rng(9)
    boxes = [np.where(_blob(rng, 12, 28), 1, 0).astype(np.int32) for _ in range(24)]
    cls = [1] * len(boxes)
    ref = watershed_split(boxes, cls)                       # (creates the pool)
    out = [None] * 4

    def call(i):
        out[i] = watershed_split(boxes, cls)
    ts = [threading.Thread(target=call, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for o in out:
        assert all(np.array_equal(a, b) for a, b in zip(o, ref))
    pid = os.fork()
    if pid == 0:                                            # child: the pool's threads do not exist here
        try:
            got = watershed_split(boxes, cls)
            ok = all(np.array_equal(a, b) for a, b in zip(got, ref))
            os._exit(0 if ok else 1)
        except BaseException:
            os._exit(2)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
