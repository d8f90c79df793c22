"""Pin the restatement against the LIVE compiled reference (build container only: needs
/root/reference, which does not exist on the GPU box -> skipped there)."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/ttcr"), reason="reference sources absent")


@pytest.fixture(scope="module")
def O():
    from oracle import oracle as O

    O.build(with_ref=True)
    return O


SMALL = [c for c in cases.cases3d() + cases.cases2d() if np.prod(np.array(c["ncells"]) + 1) < 20000]


@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("weno", [False, True], ids=["first-order", "weno3"])
def test_restatement_bit_exact_vs_reference(O, c, dt, weno):
    if weno and not cases.weno_ok(c):
        pytest.skip("grid too thin for the WENO stencil")
    if c["dim"] == 3:
        kw = dict(dtype=dt, ncells=c["ncells"], dx=c["dx"], origin=c["origin"], slowness=c["slowness"], weno=weno,
                  src=c["src"], t0=c["t0"], cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"])
        a, b = O.solve3d(**kw), O.ref_solve3d(**kw)
    else:
        kw = dict(dtype=dt, ncells=c["ncells"], dx=c["dx"], dz=c["dz"], origin=c["origin"], slowness=c["slowness"],
                  src=c["src"], t0=c["t0"], cell_slowness=c["cell_slowness"], rcv=c["rcv"], weno=weno)
        a, b = O.solve2d(**kw), O.ref_solve2d(**kw)
    assert a["niter"] == b["niter"]
    assert a["niterw"] == b["niterw"]
    np.testing.assert_array_equal(a["tt"], b["tt"])
    np.testing.assert_array_equal(a["tt_rcv"], b["tt_rcv"])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_rotated_template_bit_exact_vs_reference(O, dt):
    """sweep45 on fresh random media, several sizes and source positions (on-node, off-node, corner, multi-point)"""
    rng = np.random.default_rng(3)
    for ncx, ncz in ((20, 30), (33, 17), (1, 9)):
        nn = (ncx + 1) * (ncz + 1)
        s = rng.uniform(0.25, 1.0, nn)
        for src in ([[0.15, 2.05]], [[0.5, 3.0]], [[0.0, 0.0]], [[0.2, 0.7], [0.4, 3.0]]):
            kw = dict(dtype=dt, ncells=(ncx, ncz), dx=0.5, dz=0.5, origin=(0, 0), slowness=s, src=src, rotated=True)
            a, b = O.solve2d(**kw), O.ref_solve2d(**kw)
            assert a["niter"] == b["niter"]
            np.testing.assert_array_equal(a["tt"], b["tt"])


def test_reference_rejects_outside_point(O):
    with pytest.raises(RuntimeError, match="outside grid"):
        O.ref_solve3d(np.float64, (4, 4, 4), 1.0, (0, 0, 0), np.ones(125), [[5.0, 1.0, 1.0]])
    with pytest.raises(RuntimeError, match="outside grid"):
        O.solve3d(np.float64, (4, 4, 4), 1.0, (0, 0, 0), np.ones(125), [[5.0, 1.0, 1.0]])
