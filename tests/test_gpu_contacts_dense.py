"""k_contacts_spheres where it is NOT in its usual state (a pile at rest: four contacts per body as `a`, ~900 per block of 256 bodies):
spheres pressed into each other - thirteen and more partner contacts per body (the row's long form), several windows of the block's
1024-entry contact list.  A collapsing million-sphere pile reaches that state around tick 150 (tools/soak_tiles.py found it: the second
window's listing faulted); here a small lattice at a pitch of 0.62 diameters starts in it."""
import numpy as np
import pytest

from tests.util import compare_constraints, oracle_world, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mgf_amd
    c = mgf_amd.Context(0)
    yield c
    c.close()


def dense_scene(nx, pitch, seed=3):
    from mgf_amd import scenes
    rng = np.random.default_rng(seed)
    i, j, k = np.meshgrid(np.arange(nx), np.arange(nx), np.arange(nx), indexing="ij")
    c = (np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1) * pitch).astype(np.float32)
    c += rng.uniform(-0.03, 0.03, c.shape).astype(np.float32)
    c[:, 0] -= np.float32(0.5 * nx * pitch); c[:, 2] -= np.float32(0.5 * nx * pitch); c[:, 1] += np.float32(0.5)
    c = c[rng.permutation(len(c))]
    v0 = rng.uniform(-0.5, 0.5, c.shape).astype(np.float32)
    terrain = scenes.box_terrain(nx * pitch + 4.0, nx * pitch + 6.0, (0.0, 0.0, 0.0))
    return scenes._scene(f"dense_{nx}", scenes._spheres(c, 0.5), terrain, v0=v0, iters=4)


@pytest.mark.parametrize("nx,pitch", [(16, 0.62), (20, 0.7)])
def test_pressed_spheres_contact_lists_match_the_oracle_and_the_list_path(ctx, nx, pitch):
    import mgf_amd
    sc = dense_scene(nx, pitch)
    dt, it = float(sc["dt"]), sc["iters"]
    gw, gl, ow = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc), oracle_world(sc)
    gl.set_option("fused_contacts", 0)  # (the candidate lists' path: k_lists_spheres, k_setup_pairs)
    for tick in range(3):
        sg, sl = gw.step(dt, it), gl.step(dt, it)
        so = ow.step(dt, it)
        assert sg.n_constraints == so.n_constraints == sl.n_constraints, tick
        cg, co = gw.constraints(), ow.constraints()
        if tick == 0:  # what the test is for: long rows and blocks of several list windows
            per_a = np.bincount(co["a"][co["b"] >= 0], minlength=len(gw))
            assert per_a.max() > 12 and len(co) > 6 * len(gw), (per_a.max(), len(co))  # (256 bodies x 6: past the list's first 1024 entries)
        compare_constraints(cg, co)
        compare_constraints(gl.constraints(), co)
        g, o = gw.state(), ow.state()
        for k in ("x", "q", "v", "omega"):
            assert values_equal(g[k], o[k]), (tick, k)
