"""comp_tri_far (mgf_amd/csrc/dev_geom.h): the cheap conservative reject the r06 front end runs ahead of the body-triangle tests.  It may only
drop a (body, face) candidate for which the reference's tests (collision.rs:610-1086) report nothing - including where those tests' own f32
arithmetic reports contacts a radius and a bit away from a 200 m edge.  Millions of random problems at scales from centimetres to hundreds of
metres through mgf_tri_reject_batch: never a contact among the dropped ones; and the reject must drop a good part of what is plainly apart."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def _problems(rng, n, tri_size, near):
    """triangles of about `tri_size` somewhere within 50 x their size of the origin; bodies (half spheres, half capsules, r and |d| up to a few
    metres) placed relative to a random point OF the triangle at a distance of about `near` x (r + |d|) - from touching to clearly apart"""
    centre = rng.uniform(-50.0, 50.0, (n, 1, 3)) * tri_size
    tris = (centre + rng.normal(0.0, 1.0, (n, 3, 3)) * tri_size).astype(np.float32)
    # a share of long thin and of almost degenerate triangles
    thin = rng.random(n) < 0.2
    tris[thin, 2] = (tris[thin, 0] + (tris[thin, 1] - tris[thin, 0]) * rng.uniform(0.0, 1.0, (thin.sum(), 1)) + rng.normal(0.0, 1e-3, (thin.sum(), 3)) * tri_size).astype(np.float32)
    tag = (rng.random(n) < 0.5).astype(np.int32)
    r = rng.uniform(0.05, 2.0, n).astype(np.float32)
    d = (rng.normal(0.0, 1.0, (n, 3)) * rng.uniform(0.0, 3.0, (n, 1))).astype(np.float32)
    d[tag == 0] = 0.0
    w = rng.dirichlet((1.0, 1.0, 1.0), n)
    on_tri = np.einsum("nk,nkj->nj", w, tris.astype(np.float64))
    reach = (r + np.linalg.norm(d, axis=1)).astype(np.float64)
    off = rng.normal(0.0, 1.0, (n, 3))
    off /= np.linalg.norm(off, axis=1, keepdims=True)
    p = (on_tri + off * (reach * rng.uniform(0.0, near, n))[:, None] - 0.5 * d * rng.uniform(0.0, 2.0, (n, 1))).astype(np.float32)
    delta = (rng.normal(0.0, 1.0, (n, 3)) * rng.choice([0.0, 1e-3, 0.05, 0.5, 4.0], (n, 1))).astype(np.float32)
    return tag, p, d, r, delta, tris


def _grid_problems(rng, n, cell):
    """the faces of a heightfield (right triangles on a grid of pitch `cell`, heights within a fifth of it) and bodies lying on and beside them
    the way a pile does: axes exactly along x or z (parallel to the faces' edges - the reference's exact-parallel branch, collision.rs:915),
    exactly level, or anywhere; at rest, creeping, or falling"""
    ij = rng.integers(-20, 20, (n, 2)).astype(np.float64)
    h = rng.uniform(-0.2, 0.2, (n, 4)) * cell * (rng.random((n, 1)) < 0.7)      # (three in ten: a level floor)
    x0, z0 = ij[:, 0] * cell, ij[:, 1] * cell
    corners = np.stack([np.stack([x0, h[:, 0], z0], 1), np.stack([x0 + cell, h[:, 1], z0], 1),
                        np.stack([x0, h[:, 2], z0 + cell], 1), np.stack([x0 + cell, h[:, 3], z0 + cell], 1)], 1)
    upper = rng.random(n) < 0.5
    tris = np.where(upper[:, None, None], corners[:, [1, 3, 2]], corners[:, [0, 1, 2]]).astype(np.float32)
    tag = (rng.random(n) < 0.7).astype(np.int32)
    r = rng.choice([0.1, 0.25, 0.3, 0.5], n).astype(np.float32)
    length = rng.choice([0.2, 0.6, 1.0, 1.4, 3.0], n)
    kind = rng.integers(0, 4, n)
    axis = rng.normal(0.0, 1.0, (n, 3))
    axis[kind == 0] = (1.0, 0.0, 0.0); axis[kind == 1] = (0.0, 0.0, 1.0)
    axis[kind == 2, 1] = 0.0
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    d = (axis * length[:, None]).astype(np.float32)
    d[tag == 0] = 0.0
    w = rng.dirichlet((1.0, 1.0, 1.0), n)
    on_tri = np.einsum("nk,nkj->nj", w, tris.astype(np.float64))
    side = rng.normal(0.0, 1.0, (n, 3)) * np.array([1.0, 0.0, 1.0]) * rng.choice([0.0, 0.3, 1.0, 2.5], (n, 1)) * cell   # on the face ... two cells away
    height = r * rng.choice([0.5, 0.98, 1.0, 1.02, 1.5, 3.0], n)
    p = (on_tri + side + np.array([0.0, 1.0, 0.0]) * height[:, None] - d * rng.uniform(0.0, 1.0, (n, 1))).astype(np.float32)
    delta = (rng.normal(0.0, 1.0, (n, 3)) * rng.choice([1e-5, 1e-3, 0.02, 0.2], (n, 1))).astype(np.float32)
    fall = rng.random(n) < 0.5
    delta[fall, 0] = 0.0; delta[fall, 2] = 0.0
    return tag, p, d, r, delta, tris


@pytest.mark.parametrize("cell", [0.25, 1.0, 4.0, 50.0])
def test_the_reject_never_drops_a_contact_on_a_heightfield(ctx, cell):
    import mgf_amd
    rng = np.random.default_rng(int(cell * 100) + 11)
    dropped = contacts = total = 0
    for _ in range(5):
        tag, p, d, r, delta, tris = _grid_problems(rng, 200_000, cell)
        far, cnt = mgf_amd._capi.tri_reject_batch(ctx, tag, p, d, r, delta, tris)
        bad = np.nonzero((far == 1) & (cnt > 0))[0]
        assert len(bad) == 0, (cell, len(bad), [(int(tag[i]), p[i].tolist(), d[i].tolist(), float(r[i]), delta[i].tolist(), tris[i].tolist(), int(cnt[i])) for i in bad[:2]])
        dropped += int(far.sum()); contacts += int((cnt > 0).sum()); total += len(far)
    assert contacts > 0.05 * total and dropped > 0.1 * total, (contacts, dropped, total)


@pytest.mark.parametrize("tri_size", [0.05, 0.5, 3.0, 40.0, 400.0])
def test_the_reject_never_drops_a_contact(ctx, tri_size):
    import mgf_amd
    rng = np.random.default_rng(int(tri_size * 1000) + 6)
    dropped = contacts = dropped_apart = apart = 0
    for near in (0.6, 1.0, 1.1, 1.6, 4.0):
        tag, p, d, r, delta, tris = _problems(rng, 200_000, tri_size, near)
        far, cnt = mgf_amd._capi.tri_reject_batch(ctx, tag, p, d, r, delta, tris)
        bad = np.nonzero((far == 1) & (cnt > 0))[0]
        assert len(bad) == 0, (tri_size, near, len(bad), [(int(tag[i]), p[i].tolist(), d[i].tolist(), float(r[i]), delta[i].tolist(), tris[i].tolist(), int(cnt[i])) for i in bad[:2]])
        dropped += int(far.sum()); contacts += int((cnt > 0).sum())
        if near == 4.0:
            dropped_apart += int(far.sum()); apart += len(far)
    assert contacts > 50_000, contacts                      # (the problems are not all misses)
    if tri_size <= 3.0:
        assert dropped_apart > 0.3 * apart, (dropped_apart, apart)  # (... and what is plainly apart is dropped - but for capsules that do not move, short ones pointing
                                                                    # at the face and slivers, which the reference's own quirks keep; beside huge faces the reach grows on purpose)


def test_the_floor_diagonal_of_a_wide_box(ctx):
    """the case BASELINE config 5 found (EXPERIMENTS.md, round 6): a capsule of r = 0.3 whose axis ends 0.31 from the 205 m diagonal of a floor -
    the reference's edge test reports a contact there (f32: |m|^2 |D|^2 - (m.D)^2 with m from the edge's far end); the reject must not drop it"""
    import mgf_amd
    tag = np.array([1], np.int32)
    p = np.array([[-62.7942886, 0.668068588, 62.7116165]], np.float32)
    d = np.array([[0.531787515, -0.429781318, -0.729719162]], np.float32)
    r = np.array([0.300000012], np.float32)
    delta = np.array([[0.000573722355, 0.000415271294, 0.00104012177]], np.float32)
    tris = np.array([[[-72.4000015, 0, 72.4000015], [72.4000015, 0, 72.4000015], [72.4000015, 0, -72.4000015]]], np.float32)
    far, cnt = mgf_amd._capi.tri_reject_batch(ctx, tag, p, d, r, delta, tris)
    assert cnt[0] == 1 and far[0] == 0
