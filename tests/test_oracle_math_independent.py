"""The oracle's restatements of cgmath 0.17 (quaternion from_arc, rotate_vector, Matrix3::from(Quaternion), Matrix3::invert) are
written from the crate's published algorithms - the crate itself is not in /root/reference ("restated from memory").  They
cannot be pinned bit for bit without its source; here they are at least held against INDEPENDENT implementations of the same
mathematics (scipy's Rotation, numpy's inverse) to float accuracy, which catches what a restatement can get wrong: a transposed
matrix, a swapped multiplication order, a conjugate, the wrong quaternion component order."""
import ctypes as C

import numpy as np
from scipy.spatial.transform import Rotation

from mgf_amd import scenes
from oracle import oracle as O
from tests.util import oracle_world


def _from_arc(src, dst):
    q = O.Quat()
    O.lib().mgfo_quat_from_arc(C.byref(O.vec3(src)), C.byref(O.vec3(dst)), C.byref(q))
    return np.array([q.s, q.x, q.y, q.z], np.float64)


def _rotate(q, v):
    out = O.Vec3()
    qq = O.Quat(); qq.s, qq.x, qq.y, qq.z = [float(t) for t in q]
    O.lib().mgfo_rotate_vector(C.byref(qq), C.byref(O.vec3(v)), C.byref(out))
    return np.array([out.x, out.y, out.z], np.float64)


def test_from_arc_and_rotate_vector_against_scipy():
    rng = np.random.default_rng(7)
    for _ in range(300):
        a, b = rng.normal(size=3), rng.normal(size=3)
        a /= np.linalg.norm(a); b /= np.linalg.norm(b)
        q = _from_arc(a, b)                      # Quaternion::from_arc(src, dst, None): the shortest rotation of src onto dst
        assert abs(np.linalg.norm(q) - 1.0) < 1e-5
        R = Rotation.from_quat([q[1], q[2], q[3], q[0]])  # scipy: (x, y, z, w)
        assert np.allclose(R.apply(a), b, atol=2e-6), "from_arc does not carry src onto dst"
        assert abs(np.dot(q[1:], a)) < 2e-6 and abs(np.dot(q[1:], b)) < 2e-6, "the axis of the shortest arc is normal to both"
        v = rng.normal(size=3)
        assert np.allclose(_rotate(q, v), R.apply(v), atol=5e-6), "rotate_vector differs from scipy's rotation by the same quaternion"


def test_world_inverse_inertia_is_R_Iinv_Rt_of_the_integrated_orientation():
    """physics.rs:226-232: q <- normalize(q + (0, w dt) * 0.5 * q); I_world^-1 = R(q) * I_body^-1 * R(q)^T with R = Matrix3::from(q).
    Capsules (anisotropic inertia) spinning freely: what ConstrainedSet::get reports must be that product for scipy's R(q) and
    numpy's inverse of the body tensor - and q itself must follow the first-order update."""
    scene = scenes.capsule_field(3, 2, 3, pitch=4.0, y0=40.0)  # far apart and high: no contact within the test
    ow = oracle_world(scene)
    n = len(scene["comps"])
    rng = np.random.default_rng(3)
    omega = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    ow.set_state(omega=omega)
    inv0 = [ow.get(i)["inv_moment"].reshape(3, 3).astype(np.float64) for i in range(n)]   # q = identity: the body tensor's inverse
    for i in range(n):
        assert np.allclose(inv0[i], inv0[i].T, atol=1e-7) and np.linalg.eigvalsh(inv0[i]).min() > 0
        assert np.allclose(np.linalg.inv(np.linalg.inv(inv0[i])), inv0[i], rtol=1e-9)
    dt = float(scene["dt"])
    q_prev = ow.state()["q"].astype(np.float64)
    for tick in range(25):
        w_before = ow.state()["omega"].astype(np.float64)
        ow.step(dt, 0)
        st = ow.state()
        q = st["q"].astype(np.float64)
        for i in range(n):
            # the update (Hamilton product, scalar first)
            s0, v0 = q_prev[i, 0], q_prev[i, 1:]
            wv = w_before[i] * dt * 0.5
            dq = np.concatenate([[-np.dot(wv, v0)], s0 * wv + np.cross(wv, v0)])
            want = q_prev[i] + dq
            want /= np.linalg.norm(want)
            assert np.allclose(q[i], want, atol=3e-6), f"tick {tick} body {i}: orientation update"
            R = Rotation.from_quat([q[i, 1], q[i, 2], q[i, 3], q[i, 0]]).as_matrix()
            got = ow.get(i)["inv_moment"].reshape(3, 3).astype(np.float64)
            # the 9 words are the matrix's columns one after the other (cgmath is column-major); the product is symmetric either way
            assert np.allclose(got, R @ inv0[i] @ R.T, rtol=2e-4, atol=2e-6 * np.abs(inv0[i]).max()), f"tick {tick} body {i}: R I^-1 R^T"
        q_prev = q
