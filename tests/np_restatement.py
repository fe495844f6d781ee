"""TEST INFRASTRUCTURE ONLY.  A second, independently written restatement (numpy float32 scalars) of the
parts of the mgf hot path that the reference's own tests do NOT pin:

    RigidBodyVec::complete_motion / integrate      physics.rs:222-269
    ConstrainedSet::get for Dynamic / Static        physics.rs:272-304
    compute_basis                                   geom.rs:1138-1145
    ContactConstraint::new (one contact)            solver.rs:101-191
    ContactConstraint::solve (one contact)          solver.rs:203-252
    Solver::solve                                   solver.rs:72-78

It is deliberately written in a different style from oracle/*.hpp (plain tuples of np.float32, one
expression per reference expression) so that a transcription slip in either restatement shows up as a
bit difference in tests/test_oracle_cross_check.py.  cgmath 0.17 conventions used here:
Matrix3 is column-major, M*v = c0*v.x + c1*v.y + c2*v.z; q.normalize() = q * (1/|q|);
Quaternion * Quaternion is the Hamilton product written out per component.
Pure-Python loops: small scenes only.
"""
import numpy as np

F = np.float32
ZERO, ONE, HALF = F(0.0), F(1.0), F(0.5)


# ---- vectors (tuples of np.float32) -------------------------------------------------------------------
def vec(a):
    return (F(a[0]), F(a[1]), F(a[2]))


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def scale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def dot(a, b):  # cgmath Vector3::dot = sum of products, left to right
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def normalize(a):
    inv = ONE / np.sqrt(dot(a, a))
    return scale(a, inv)


# ---- matrices: tuple of three column vectors ----------------------------------------------------------
def mat_vec(m, v):
    return add(add(scale(m[0], v[0]), scale(m[1], v[1])), scale(m[2], v[2]))


def mat_mat(a, b):
    return (mat_vec(a, b[0]), mat_vec(a, b[1]), mat_vec(a, b[2]))


def transpose(m):
    return ((m[0][0], m[1][0], m[2][0]), (m[0][1], m[1][1], m[2][1]), (m[0][2], m[1][2], m[2][2]))


def mat_from_flat(f9):  # the oracle hands matrices out column by column
    return (vec(f9[0:3]), vec(f9[3:6]), vec(f9[6:9]))


def mat_flat(m):
    return np.array([m[0][0], m[0][1], m[0][2], m[1][0], m[1][1], m[1][2], m[2][0], m[2][1], m[2][2]], np.float32)


ZERO_MAT = ((ZERO, ZERO, ZERO),) * 3


# ---- quaternions (s, x, y, z) --------------------------------------------------------------------------
def quat_mul(p, q):
    return (p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3],
            p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2],
            p[0] * q[2] + p[2] * q[0] + p[3] * q[1] - p[1] * q[3],
            p[0] * q[3] + p[3] * q[0] + p[1] * q[2] - p[2] * q[1])


def quat_to_mat(q):
    s, x, y, z = q
    x2, y2, z2 = x + x, y + y, z + z
    xx2, xy2, xz2 = x2 * x, x2 * y, x2 * z
    yy2, yz2, zz2 = y2 * y, y2 * z, z2 * z
    sy2, sz2, sx2 = y2 * s, z2 * s, x2 * s
    return ((ONE - yy2 - zz2, xy2 + sz2, xz2 - sy2),
            (xy2 - sz2, ONE - xx2 - zz2, yz2 + sx2),
            (xz2 + sy2, yz2 - sx2, ONE - xx2 - yy2))


# ---- RigidBodyVec ------------------------------------------------------------------------------------
class Bodies:
    """Sphere bodies only (the collider rebuild is then `Sphere{c: x, r}` swept by v*dt)."""

    def __init__(self, x, q, v, omega, delta, force, inv_mass, inv_moment_body, restitution, friction):
        n = len(x)
        self.x = [vec(x[i]) for i in range(n)]
        self.q = [(F(q[i][0]), F(q[i][1]), F(q[i][2]), F(q[i][3])) for i in range(n)]
        self.v = [vec(v[i]) for i in range(n)]
        self.omega = [vec(omega[i]) for i in range(n)]
        self.delta = [vec(delta[i]) for i in range(n)]
        self.force = [vec(force[i]) for i in range(n)]
        self.torque = [(ZERO, ZERO, ZERO)] * n
        self.inv_mass = [F(m) for m in inv_mass]
        self.inv_moment_body = [mat_from_flat(inv_moment_body[i]) for i in range(n)]
        self.inv_moment = list(self.inv_moment_body)
        self.restitution = [F(e) for e in restitution]
        self.friction = [F(f) for f in friction]

    def complete_motion(self):  # physics.rs:262-269
        for i in range(len(self.x)):
            self.x[i] = add(self.x[i], self.delta[i])

    def integrate(self, dt):  # physics.rs:222-253
        dt = F(dt)
        for i in range(len(self.x)):
            q = self.q[i]
            w = scale(self.omega[i], dt)
            spin = (ZERO, w[0], w[1], w[2])
            half = tuple(c * HALF for c in spin)
            dq = quat_mul(half, q)
            qn = tuple(q[k] + dq[k] for k in range(4))
            mag = np.sqrt(qn[0] * qn[0] + (qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]))
            inv = ONE / mag
            self.q[i] = tuple(c * inv for c in qn)
        for i in range(len(self.x)):
            r = quat_to_mat(self.q[i])
            self.inv_moment[i] = mat_mat(mat_mat(r, self.inv_moment_body[i]), transpose(r))
        for i in range(len(self.x)):
            self.v[i] = add(self.v[i], scale(scale(self.force[i], self.inv_mass[i]), dt))
        for i in range(len(self.x)):
            self.omega[i] = add(self.omega[i], scale(mat_vec(self.inv_moment[i], self.torque[i]), dt))
        for i in range(len(self.x)):
            self.delta[i] = scale(self.v[i], dt)

    def get(self, ref, static_center):  # physics.rs:272-304; ref < 0 = Static{center, friction: 0}
        if ref >= 0:
            return (self.v[ref], self.omega[ref], add(self.x[ref], self.delta[ref]), self.restitution[ref], self.friction[ref],
                    self.inv_mass[ref], self.inv_moment[ref])
        z = (ZERO, ZERO, ZERO)
        return (z, z, vec(static_center), ZERO, ZERO, ZERO, ZERO_MAT)

    def set(self, ref, v, omega):  # physics.rs:306-314
        if ref >= 0:
            self.v[ref], self.omega[ref] = v, omega


def compute_basis(n):  # geom.rs:1138-1145
    if abs(n[0]) >= F(0.57735):
        b = normalize((n[1], -n[0], ZERO))
    else:
        b = normalize((ZERO, n[2], -n[1]))
    return b, cross(n, b)


BAUMGARTE, SLOP = F(0.2), F(0.05)  # DefaultContactConstraintParams solver.rs:40-50


class Constraint:
    """ContactConstraint with a single contact (every manifold on this path has one)."""

    def __init__(self, bodies, a, b, normal, local_a, local_b, dt, static_center):  # solver.rs:101-191
        va, oa, xa, rest_a, fric_a, im_a, I_a = bodies.get(a, static_center)
        vb, ob, xb, rest_b, fric_b, im_b, I_b = bodies.get(b, static_center)
        self.a, self.b, self.static_center = a, b, static_center
        self.n = vec(normal)
        self.t = compute_basis(self.n)  # Manifold::from manifold.rs:125,144
        restitution = max(rest_a, rest_b)
        self.friction = np.sqrt(fric_a * fric_b)
        ra, rb = vec(local_a), vec(local_b)
        self.ra, self.rb = ra, rb
        ca, cb = add(ra, xa), add(rb, xb)
        ra_cn, rb_cn = cross(ra, self.n), cross(rb, self.n)
        pen = dot(sub(cb, ca), self.n)
        dv = sub(sub(add(vb, cross(ob, rb)), va), cross(oa, ra))
        rel_v = dot(dv, self.n)
        self.bias = -BAUMGARTE / F(dt) * (ZERO if pen > ZERO else pen + SLOP) + (-restitution * rel_v if rel_v < F(-1.0) else ZERO)
        self.normal_mass = ONE / (im_a + dot(ra_cn, mat_vec(I_a, ra_cn)) + im_b + dot(rb_cn, mat_vec(I_b, rb_cn)))
        tm = []
        for k in range(2):
            ra_ct, rb_ct = cross(ra, self.t[k]), cross(rb, self.t[k])
            tm.append(ONE / (im_a + dot(ra_ct, mat_vec(I_a, ra_ct)) + im_b + dot(rb_ct, mat_vec(I_b, rb_ct))))
        self.tangent_mass = tm
        self.normal_impulse = ZERO

    def solve(self, bodies):  # solver.rs:203-252
        va, oa, _, _, _, im_a, I_a = bodies.get(self.a, self.static_center)
        vb, ob, _, _, _, im_b, I_b = bodies.get(self.b, self.static_center)
        ra, rb = self.ra, self.rb
        dv = sub(sub(add(vb, cross(ob, rb)), va), cross(oa, ra))
        for k in range(2):
            lam = -dot(dv, self.t[k]) * self.tangent_mass[k]
            # the clamped accumulator (:222-226) is dead state: the applied impulse uses the raw lambda
            impulse = scale(self.t[k], lam)
            va = sub(va, scale(impulse, im_a))
            oa = sub(oa, mat_vec(I_a, cross(ra, impulse)))
            vb = add(vb, scale(impulse, im_b))
            ob = add(ob, mat_vec(I_b, cross(rb, impulse)))
        dv = sub(sub(add(vb, cross(ob, rb)), va), cross(oa, ra))
        vn = dot(dv, self.n)
        lam = self.normal_mass * (-vn + self.bias)
        prev = self.normal_impulse
        self.normal_impulse = max(prev + lam, ZERO)
        lam = self.normal_impulse - prev
        impulse = scale(self.n, lam)
        va = sub(va, scale(impulse, im_a))
        oa = sub(oa, mat_vec(I_a, cross(ra, impulse)))
        vb = add(vb, scale(impulse, im_b))
        ob = add(ob, mat_vec(I_b, cross(rb, impulse)))
        bodies.set(self.a, va, oa)
        bodies.set(self.b, vb, ob)


def solver_solve(constraints, bodies, iters):  # solver.rs:72-78
    for _ in range(iters):
        for c in constraints:
            c.solve(bodies)
