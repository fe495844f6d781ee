"""Seeded synthetic scenes for the mgf hot path (BASELINE.json configs).

Pure numpy, no torch, no oracle: a scene is plain arrays (components, masses, initial
velocities, terrain mesh) that are fed unchanged to the HIP world (mgf_amd.World) and,
in tests/bench only, to the CPU oracle, so both sides step identical inputs.

Component records use the C-ABI layout `mgf_component` (include/mgf_hip.h):
tag 0 = Sphere{c = p, r}, tag 1 = Capsule{a = p, d, r}.
"""
import numpy as np

COMPONENT_DTYPE = np.dtype([("tag", "<i4"), ("p", "<f4", 3), ("d", "<f4", 3), ("r", "<f4")])

SEED = 0x6D6766  # "mgf"

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, stream=0):
    """n outputs of SplitMix64 started at `seed` (+ an independent stream offset)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1342543DE82EF95) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n, stream=0):
    """f32 uniforms in [0,1): top 24 bits of SplitMix64."""
    return ((splitmix64(seed, n, stream) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def uniform(seed, n, lo, hi, stream=0):
    return (np.float32(lo) + uniform01(seed, n, stream) * np.float32(hi - lo)).astype(np.float32)


def fisher_yates(seed, n, stream=0):
    """Body index order of SURVEY.md 8d: the seeded Fisher-Yates shuffle of 0..n-1 (Durstenfeld's form: for i = n-1 down to 1,
    swap i with j = draw mod (i + 1)), draws = consecutive SplitMix64 outputs of the stream."""
    if n <= 1:
        return np.arange(n, dtype=np.int64)
    draws = splitmix64(seed, n - 1, stream)
    js = (draws % np.arange(n, 1, -1, dtype=np.uint64)).astype(np.int64)  # draw k serves i = n - 1 - k: modulus i + 1 = n - k
    perm = list(range(n))
    for k, j in enumerate(js.tolist()):
        i = n - 1 - k
        perm[i], perm[j] = perm[j], perm[i]
    return np.asarray(perm, dtype=np.int64)


def argsort_permutation(seed, n, stream=0):
    """Rounds 1-2's body order, kept as a named variant (order="argsort"): stable argsort of SplitMix64 keys."""
    return np.argsort(splitmix64(seed, n, stream), kind="stable")


def seeded_permutation(seed, n, stream=0, order="fisher_yates"):
    """Body index order (it bounds the depth of the solver's dependency graph): SURVEY 8d's Fisher-Yates shuffle by
    default; "argsort" = the variant of rounds 1-2."""
    if order == "fisher_yates":
        return fisher_yates(seed, n, stream)
    if order == "argsort":
        return argsort_permutation(seed, n, stream)
    raise ValueError(f"unknown body order {order!r}")


# mgf_demo/world.rs:118-150 — 8 vertices, 10 faces, open-top box; same winding.
_BOX_FACES = np.array([(0, 1, 3), (1, 2, 3), (0, 5, 1), (0, 4, 5), (0, 3, 7), (0, 7, 4),
                       (2, 6, 3), (3, 6, 7), (1, 5, 2), (2, 5, 6)], dtype=np.uint32)


def box_terrain(half, height, pos, half_z=None):
    h, H = np.float32(half), np.float32(height)
    g = h if half_z is None else np.float32(half_z)
    verts = np.array([(-h, 0, -g), (-h, 0, g), (h, 0, g), (h, 0, -g),
                      (-h, H, -g), (-h, H, g), (h, H, g), (h, H, -g)], dtype=np.float32)
    return dict(verts=verts, faces=_BOX_FACES.copy(), pos=np.asarray(pos, np.float32))


def _spheres(centres, r):
    comps = np.zeros(len(centres), dtype=COMPONENT_DTYPE)
    comps["tag"] = 0
    comps["p"] = centres
    comps["r"] = r
    return comps


def _scene(name, comps, terrain, v0=None, dt=1.0 / 60.0, iters=10, mass=1.0, rest=0.3, fric=0.6,
           gravity=(0.0, -9.8, 0.0)):
    n = len(comps)
    return dict(name=name, comps=comps, terrain=terrain, dt=np.float32(dt), iters=int(iters),
                mass=np.full(n, mass, np.float32), restitution=np.full(n, rest, np.float32),
                friction=np.full(n, fric, np.float32),
                force=np.tile(np.asarray(gravity, np.float32), (n, 1)),
                v0=None if v0 is None else np.ascontiguousarray(v0, np.float32))


def balls_demo(num=8, extra_ball=False, iters=10):
    """mgf_demo/balls.rs:67-96 with `num` spheres per axis (BASELINE config 1: num=8 -> 512).
    The unmodified demo is num=11 (1500^(1/3) as usize), extra_ball=True, iters=20."""
    f = np.float32
    rad = f(0.5)
    shift = f(2.5) * rad
    centerx = shift * f(num) / f(2.0)
    centery = shift * f(num) / f(2.0)
    i, j, k = np.meshgrid(np.arange(num), np.arange(num), np.arange(num), indexing="ij")  # i outer, k inner
    x = i.astype(np.float32).ravel() * f(2.5) * rad - centerx
    y = f(10.0) + j.astype(np.float32).ravel() * f(2.5) * rad + centery * f(2.0)
    z = k.astype(np.float32).ravel() * f(2.5) * rad - centerx
    c = np.stack([x, y, z], axis=1).astype(np.float32)
    if extra_ball:
        c = np.concatenate([c, np.array([[0.0, 130.0, 0.0]], np.float32)])
    return _scene(f"balls_demo_{len(c)}", _spheres(c, 0.5), box_terrain(10.0, 10.0, (0.0, -10.0, 0.0)), iters=iters)


def sphere_pile(nx, ny, nz, seed=SEED, iters=10, shuffle=True, x_offset=0.0, order="fisher_yates"):
    """BASELINE config 2 family: nx*ny*nz spheres r=0.5 on a pitch-1.0 lattice with jitter
    U(-0.05,0.05)^3 and v0 ~ U(-1,1)^3, resting on the floor of an open box; body index order
    is a seeded permutation.  sphere_pile(64,64,64) is the 262 144-sphere headline config."""
    n = nx * ny * nz
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([i.ravel() - (nx - 1) / 2.0, j.ravel() + 0.5, k.ravel() - (nz - 1) / 2.0], axis=1).astype(np.float32)
    jit = np.stack([uniform(seed, n, -0.05, 0.05, stream=s) for s in (1, 2, 3)], axis=1)
    v0 = np.stack([uniform(seed, n, -1.0, 1.0, stream=s) for s in (4, 5, 6)], axis=1)
    c = (base + jit).astype(np.float32)
    c[:, 0] += np.float32(x_offset)
    if shuffle:
        perm = seeded_permutation(seed, n, stream=7, order=order)
        c, v0 = c[perm], v0[perm]
    half = max(nx, nz) / 2.0 + 1.0
    terrain = box_terrain(half, ny + 2.0, (x_offset, 0.0, 0.0))
    return _scene(f"sphere_pile_{nx}x{ny}x{nz}", _spheres(c, 0.5), terrain, v0=v0, iters=iters)


def sphere_pile_tile(nx, ny, nz, rank, world_size, seed=SEED, iters=10, drift=None):
    """x-slab tile `rank` of a (world_size*nx) x ny x nz pile in ONE open box (BASELINE config 4 family;
    world_size == 1 is sphere_pile).  Tile r owns lattice columns [r*nx, (r+1)*nx); its bodies, jitter and
    velocities come from per-tile SplitMix64 streams so every rank can build its own tile independently.
    `drift` = a velocity added to every body (tests: makes bodies cross slab faces); scene["tags"] = global body ids."""
    if world_size == 1:
        sc = sphere_pile(nx, ny, nz, seed=seed, iters=iters)
        sc["x_range"] = (-np.inf, np.inf)
        if drift is not None:
            sc["v0"] = (sc["v0"] + np.asarray(drift, np.float32)).astype(np.float32)
        sc["tags"] = np.arange(nx * ny * nz, dtype=np.uint32)
        return sc
    n = nx * ny * nz
    gx = world_size * nx
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([i.ravel() + rank * nx - (gx - 1) / 2.0, j.ravel() + 0.5, k.ravel() - (nz - 1) / 2.0], axis=1).astype(np.float32)
    st = 16 * (rank + 1)
    jit = np.stack([uniform(seed, n, -0.05, 0.05, stream=st + s) for s in (1, 2, 3)], axis=1)
    v0 = np.stack([uniform(seed, n, -1.0, 1.0, stream=st + s) for s in (4, 5, 6)], axis=1)
    c = (base + jit).astype(np.float32)
    perm = seeded_permutation(seed, n, stream=st + 7)  # (SURVEY 8d's Fisher-Yates shuffle, per tile)
    c, v0 = c[perm], v0[perm]
    if drift is not None:
        v0 = (v0 + np.asarray(drift, np.float32)).astype(np.float32)
    terrain = box_terrain(gx / 2.0 + 1.0, ny + 2.0, (0.0, 0.0, 0.0), half_z=nz / 2.0 + 1.0)
    sc = _scene(f"sphere_pile_tile{rank}of{world_size}_{nx}x{ny}x{nz}", _spheres(c, 0.5), terrain, v0=v0, iters=iters)
    sc["x_range"] = (rank * nx - gx / 2.0, (rank + 1) * nx - gx / 2.0)
    sc["tags"] = (rank * n + np.arange(n)).astype(np.uint32)
    return sc


def heightfield_terrain(quads_x, quads_z, size_x, size_z, amplitude, seed=SEED, pos=(0.0, 0.0, 0.0)):
    """Static triangle-soup terrain: (quads_x x quads_z) quads = 2*quads_x*quads_z triangles over
    [-size_x/2, size_x/2] x [-size_z/2, size_z/2], vertex heights U(-amplitude, amplitude), normals up."""
    vx, vz = quads_x + 1, quads_z + 1
    xs = np.linspace(-size_x / 2.0, size_x / 2.0, vx, dtype=np.float32)
    zs = np.linspace(-size_z / 2.0, size_z / 2.0, vz, dtype=np.float32)
    h = uniform(seed, vx * vz, -amplitude, amplitude, stream=21).reshape(vx, vz)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    verts = np.stack([X.ravel(), h.ravel(), Z.ravel()], axis=1).astype(np.float32)
    i, k = np.meshgrid(np.arange(quads_x), np.arange(quads_z), indexing="ij")
    v00 = (i * vz + k).ravel()
    v01, v10, v11 = v00 + 1, v00 + vz, v00 + vz + 1
    faces = np.concatenate([np.stack([v00, v01, v10], axis=1), np.stack([v01, v11, v10], axis=1)]).astype(np.uint32)
    # interleave the two triangles of each quad (insertion order matters for the mesh BVH)
    faces = faces.reshape(2, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
    return dict(verts=verts, faces=np.ascontiguousarray(faces), pos=np.asarray(pos, np.float32))


def _capsules(centres, dirs, half_len, r):
    comps = np.zeros(len(centres), dtype=COMPONENT_DTYPE)
    comps["tag"] = 1
    d = (dirs * np.float32(2.0 * half_len)).astype(np.float32)
    comps["p"] = (centres - d * np.float32(0.5)).astype(np.float32)
    comps["d"] = d
    comps["r"] = r
    return comps


def _unit_vectors(seed, n, stream):
    z = uniform(seed, n, -1.0, 1.0, stream=stream)
    phi = uniform(seed, n, 0.0, 2.0 * np.pi, stream=stream + 1)
    s = np.sqrt(np.maximum(0.0, 1.0 - z.astype(np.float64) ** 2))
    return np.stack([s * np.cos(phi), z, s * np.sin(phi)], axis=1).astype(np.float32)


def capsule_field(nx, ny, nz, quads=None, seed=SEED, iters=10, pitch=2.6, y0=1.2, sphere_fraction=0.0, order="fisher_yates"):
    """BASELINE config 3 family: nx*ny*nz capsules (r = 0.5, |d| = 0.5: the demo capsule of capsules.rs:67-75
    at half scale) with seeded random orientations on a lattice of pitch 2.6 above a heightfield of
    2*quads^2 triangles (Capsule-Triangle narrowphase).  capsule_field(128, 32, 32, quads=158) is the 131 072-capsule /
    49 928-triangle configuration.  sphere_fraction > 0 mixes in spheres (r = 0.5).  capsule_field_dense = the same at
    pitch 1.6 (rounds 1-2's default: neighbours touch from the first ticks on - what the small parity scenes want)."""
    n = nx * ny * nz
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([(i.ravel() - (nx - 1) / 2.0) * pitch, y0 + j.ravel() * pitch, (k.ravel() - (nz - 1) / 2.0) * pitch], axis=1)
    jit = np.stack([uniform(seed, n, -0.05, 0.05, stream=s) for s in (31, 32, 33)], axis=1)
    c = (base + jit).astype(np.float32)
    dirs = _unit_vectors(seed, n, 34)
    comps = _capsules(c, dirs, 0.25, 0.5)
    if sphere_fraction > 0.0:
        is_sphere = uniform01(seed, n, stream=36) < sphere_fraction
        comps["tag"][is_sphere] = 0
        comps["p"][is_sphere] = c[is_sphere]
        comps["d"][is_sphere] = 0.0
    v0 = np.stack([uniform(seed, n, -0.5, 0.5, stream=s) for s in (37, 38, 39)], axis=1)
    perm = seeded_permutation(seed, n, stream=40, order=order)
    comps, v0 = comps[perm], v0[perm]
    if quads is None:
        quads = max(4, int(round(max(nx, nz) * pitch / 1.3)))
    size_x, size_z = nx * pitch + 4.0, nz * pitch + 4.0
    terrain = heightfield_terrain(quads, quads, size_x, size_z, 0.2, seed=seed)
    return _scene(f"capsule_field_{nx}x{ny}x{nz}_q{quads}", comps, terrain, v0=v0, iters=iters)


def capsule_field_dense(nx, ny, nz, **kw):
    """capsule_field at pitch 1.6: a contact-rich field (the default of rounds 1-2, kept as a named variant)."""
    kw.setdefault("pitch", 1.6)
    return capsule_field(nx, ny, nz, **kw)


def dumbbell_field(nx, ny, nz, n_plain=0, seed=SEED, iters=10, pitch=2.2, y0=1.5):
    """BASELINE config 5 family: nx*ny*nz bodies of two components each - a sphere (r = 0.5) and a capsule (|d| = 1,
    r = 0.3) side by side, randomly oriented about the vertical - on a lattice above the floor of an open box, plus
    `n_plain` ordinary spheres dropped on top.  Ordinary bodies come first in the body order, the two-part bodies after."""
    n = nx * ny * nz
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([(i.ravel() - (nx - 1) / 2.0) * pitch, y0 + j.ravel() * pitch, (k.ravel() - (nz - 1) / 2.0) * pitch], axis=1)
    jit = np.stack([uniform(seed, n, -0.1, 0.1, stream=31 + s) for s in range(3)], axis=1)
    c = (base + jit).astype(np.float32)
    c = c[seeded_permutation(seed, n, stream=38)]  # body order independent of position (bounds the dependency depth)
    phi = uniform(seed, n, 0.0, 2.0 * np.pi, stream=34).astype(np.float64)
    ax = np.stack([np.cos(phi), np.zeros(n), np.sin(phi)], axis=1).astype(np.float32)   # body axis in the horizontal plane
    comps = np.zeros(2 * n, dtype=COMPONENT_DTYPE)
    # part 0: the sphere, 0.45 along -axis; part 1: the capsule standing next to it, 0.45 along +axis
    comps["tag"][0::2] = 0
    comps["p"][0::2] = c - np.float32(0.45) * ax
    comps["r"][0::2] = 0.5
    comps["tag"][1::2] = 1
    comps["p"][1::2] = c + np.float32(0.45) * ax - np.float32([0.0, 0.5, 0.0])
    comps["d"][1::2] = np.float32([0.0, 1.0, 0.0])
    comps["r"][1::2] = 0.3
    half = max(nx, nz) * pitch / 2.0 + 2.0
    terrain = box_terrain(half, ny * pitch + 6.0, (0.0, 0.0, 0.0))
    pc = np.zeros((n_plain, 3), np.float32)
    if n_plain:
        pc[:, 0] = uniform(seed, n_plain, -half + 1.5, half - 1.5, stream=35)
        pc[:, 2] = uniform(seed, n_plain, -half + 1.5, half - 1.5, stream=36)
        pc[:, 1] = y0 + ny * pitch + 1.0 + 1.2 * np.arange(n_plain)
    sc = _scene(f"dumbbell_field_{nx}x{ny}x{nz}+{n_plain}", _spheres(pc, 0.5), terrain, iters=iters)
    v0c = np.stack([uniform(seed, n, -0.5, 0.5, stream=37 + s) for s in range(3)], axis=1)
    sc["compound"] = dict(comps=comps, comp_mass=np.tile(np.float32([1.0, 0.8]), n), offsets=np.arange(0, 2 * n + 1, 2, dtype=np.int64),
                          restitution=np.full(n, 0.3, np.float32), friction=np.full(n, 0.6, np.float32),
                          force=np.tile(np.float32([0.0, -9.8, 0.0]), (n, 1)))
    sc["v0"] = np.concatenate([np.zeros((n_plain, 3), np.float32), v0c.astype(np.float32)])
    return sc


def jack_field(nx, ny, nz, seed=SEED, iters=10, pitch=2.6, y0=1.6):
    """Bodies of FOUR components each (round 3: VERDICT r2 item 7): a "jack" - a sphere (r = 0.45) at the hub and three capsules
    (|d| = 1.4, r = 0.22) through it along three mutually orthogonal axes, the whole randomly turned about the vertical and tilted -
    nx*ny*nz of them on a lattice above the floor of an open box.  Like dumbbell_field this is the build's own definition of a body
    of several components (mgf_world_add_compound_bodies); a pair of such bodies yields up to 16 part pairs."""
    n = nx * ny * nz
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([(i.ravel() - (nx - 1) / 2.0) * pitch, y0 + j.ravel() * pitch, (k.ravel() - (nz - 1) / 2.0) * pitch], axis=1)
    jit = np.stack([uniform(seed, n, -0.1, 0.1, stream=51 + s) for s in range(3)], axis=1)
    c = (base + jit).astype(np.float32)
    c = c[seeded_permutation(seed, n, stream=58)]
    # an orthonormal frame per body: a random unit vector u, a second one made orthogonal to it, their cross product
    u = _unit_vectors(seed, n, 54).astype(np.float64)
    t = _unit_vectors(seed, n, 56).astype(np.float64)
    v = t - u * np.sum(t * u, axis=1, keepdims=True)
    v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6)
    w = np.cross(u, v)
    comps = np.zeros(4 * n, dtype=COMPONENT_DTYPE)
    comps["tag"][0::4] = 0
    comps["p"][0::4] = c
    comps["r"][0::4] = 0.45
    for a, ax in enumerate((u, v, w)):
        d = (ax * 1.4).astype(np.float32)
        comps["tag"][1 + a::4] = 1
        comps["p"][1 + a::4] = c - d * np.float32(0.5)
        comps["d"][1 + a::4] = d
        comps["r"][1 + a::4] = 0.22
    half = max(nx, nz) * pitch / 2.0 + 2.0
    terrain = box_terrain(half, ny * pitch + 6.0, (0.0, 0.0, 0.0))
    sc = _scene(f"jack_field_{nx}x{ny}x{nz}", _spheres(np.zeros((0, 3), np.float32), 0.5), terrain, iters=iters)
    v0c = np.stack([uniform(seed, n, -0.5, 0.5, stream=61 + s) for s in range(3)], axis=1)
    sc["compound"] = dict(comps=comps, comp_mass=np.tile(np.float32([1.0, 0.4, 0.4, 0.4]), n), offsets=np.arange(0, 4 * n + 1, 4, dtype=np.int64),
                          restitution=np.full(n, 0.3, np.float32), friction=np.full(n, 0.6, np.float32),
                          force=np.tile(np.float32([0.0, -9.8, 0.0]), (n, 1)))
    sc["v0"] = v0c.astype(np.float32)
    return sc


def caterpillar_field(nx, ny, nz, n_plain=0, small_every=0, seed=SEED, iters=10, pitch=(5.8, 2.4, 5.8), y0=2.0):
    """Bodies of SIXTEEN components each (round 6, SURVEY 8f-1: more parts than a body's four slots hold - the pool of Bodies::xl0): a
    "caterpillar" - a zigzag spine of ten spheres (r = 0.32, 0.5 apart) with six capsule legs (|d| = 0.7, r = 0.12) pointing down and outwards
    from three of them - randomly turned about the vertical and tilted a little, nx*ny*nz of them on a lattice above the floor of an open box,
    `n_plain` ordinary spheres dropped on top.  small_every = k > 0: every k-th body keeps only its first three components (a body of the
    four-slot kind among the pooled ones).  Like dumbbell_field and jack_field this is the build's own definition of a body of several
    components (mgf_world_add_compound_bodies; oracle: RigidBodyVec::add_compound_body)."""
    n = nx * ny * nz
    px, py, pz = pitch
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([(i.ravel() - (nx - 1) / 2.0) * px, y0 + j.ravel() * py, (k.ravel() - (nz - 1) / 2.0) * pz], axis=1)
    jit = np.stack([uniform(seed, n, -0.1, 0.1, stream=71 + s) for s in range(3)], axis=1)
    c = (base + jit).astype(np.float64)
    c = c[seeded_permutation(seed, n, stream=78)]
    phi = uniform(seed, n, 0.0, 2.0 * np.pi, stream=74).astype(np.float64)
    tilt = uniform(seed, n, -0.25, 0.25, stream=75).astype(np.float64)
    # the body frame: e0 along the spine (turned by phi about y, tilted out of the horizontal), e1 sideways, e2 up
    e0 = np.stack([np.cos(phi) * np.cos(tilt), np.sin(tilt), np.sin(phi) * np.cos(tilt)], axis=1)
    e1 = np.stack([-np.sin(phi), np.zeros(n), np.cos(phi)], axis=1)
    e2 = np.cross(e0, e1)
    local = []  # (tag, p, d, r, mass) in the body frame
    for s_ in range(10):
        local.append((0, ((s_ - 4.5) * 0.5, 0.0, 0.15 if s_ % 2 else -0.15), (0.0, 0.0, 0.0), 0.32, 0.5))
    for s_ in (1, 4, 7):
        for side in (-1.0, 1.0):
            root = ((s_ - 4.5) * 0.5, side * 0.2, -0.1)
            d = np.array([0.0, side * 0.45, -0.55]); d *= 0.7 / np.linalg.norm(d)
            local.append((1, root, tuple(d), 0.12, 0.2))
    P = len(local)
    counts = np.full(n, P, np.int64)
    if small_every:
        counts[::small_every] = 3
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    comps = np.zeros(int(offsets[-1]), dtype=COMPONENT_DTYPE)
    mass = np.zeros(int(offsets[-1]), np.float32)
    frame = lambda v: e0 * v[0] + e1 * v[1] + e2 * v[2]  # noqa: E731
    for b in range(n):
        for a in range(int(counts[b])):
            tag, p_, d_, r_, m_ = local[a]
            at = int(offsets[b]) + a
            comps["tag"][at] = tag
            comps["p"][at] = (c[b] + frame(p_)[b]).astype(np.float32)
            comps["d"][at] = frame(d_)[b].astype(np.float32) if tag else 0.0
            comps["r"][at] = r_
            mass[at] = m_
    half = max(nx * px, nz * pz) / 2.0 + 2.5
    terrain = box_terrain(half, ny * py + 8.0, (0.0, 0.0, 0.0))
    pc = np.zeros((n_plain, 3), np.float32)
    if n_plain:
        pc[:, 0] = uniform(seed, n_plain, -half + 1.5, half - 1.5, stream=76)
        pc[:, 2] = uniform(seed, n_plain, -half + 1.5, half - 1.5, stream=77)
        pc[:, 1] = y0 + ny * py + 1.0 + 1.2 * np.arange(n_plain)
    sc = _scene(f"caterpillar_field_{nx}x{ny}x{nz}+{n_plain}", _spheres(pc, 0.5), terrain, iters=iters)
    v0c = np.stack([uniform(seed, n, -0.5, 0.5, stream=81 + s) for s in range(3)], axis=1)
    sc["compound"] = dict(comps=comps, comp_mass=mass, offsets=offsets, restitution=np.full(n, 0.3, np.float32), friction=np.full(n, 0.6, np.float32),
                          force=np.tile(np.float32([0.0, -9.8, 0.0]), (n, 1)))
    sc["v0"] = np.concatenate([np.zeros((n_plain, 3), np.float32), v0c.astype(np.float32)])
    return sc


def split_by_slabs(scene, world_size, half_x):
    """x-slab tiles of a scene built by dumbbell_field: tile r gets the bodies whose centre lies in its slab of
    [-half_x, half_x), in their original order; `tags` are the bodies' indices in the undivided scene."""
    cb = scene["compound"]
    n_plain = len(scene["comps"])
    off = cb["offsets"]
    nb = len(off) - 1
    m = np.asarray(cb["comp_mass"], np.float64)
    ctr = cb["comps"]["p"].astype(np.float64) + 0.5 * cb["comps"]["d"].astype(np.float64) * (cb["comps"]["tag"][:, None] == 1)
    com_x = np.array([np.sum(ctr[off[b]:off[b + 1], 0] * m[off[b]:off[b + 1]]) / np.sum(m[off[b]:off[b + 1]]) for b in range(nb)])
    plain_x = scene["comps"]["p"][:, 0].astype(np.float64)
    edges = np.linspace(-half_x, half_x, world_size + 1)
    tiles = []
    for r in range(world_size):
        lo = -np.inf if r == 0 else edges[r]
        hi = np.inf if r + 1 == world_size else edges[r + 1]
        pi = np.nonzero((plain_x >= lo) & (plain_x < hi))[0]
        bi = np.nonzero((com_x >= lo) & (com_x < hi))[0]
        sc = dict(scene)
        for key in ("comps", "mass", "restitution", "friction", "force"):
            sc[key] = scene[key][pi]
        parts = np.concatenate([np.arange(off[b], off[b + 1]) for b in bi]) if len(bi) else np.zeros(0, np.int64)
        sizes = np.array([off[b + 1] - off[b] for b in bi], np.int64)
        sc["compound"] = dict(comps=cb["comps"][parts], comp_mass=np.asarray(cb["comp_mass"], np.float32)[parts],
                              offsets=np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64),
                              restitution=cb["restitution"][bi], friction=cb["friction"][bi], force=cb["force"][bi])
        sc["v0"] = np.concatenate([scene["v0"][pi], scene["v0"][n_plain + bi]])
        sc["tags"] = np.concatenate([pi, n_plain + bi]).astype(np.uint32)
        sc["x_range"] = (float(edges[r]), float(edges[r + 1]))
        sc["name"] = f"{scene['name']}_tile{r}of{world_size}"
        tiles.append(sc)
    return tiles


def config(idx):
    """BASELINE.json configs by index."""
    if idx == 0:
        return balls_demo(8)
    if idx == 1:
        return sphere_pile(64, 64, 64)
    if idx == 2:
        return capsule_field(128, 32, 32, quads=158)
    raise NotImplementedError(f"config {idx}")
