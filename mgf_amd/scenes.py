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


def seeded_permutation(seed, n, stream=0):
    """Body index order: stable argsort of SplitMix64 keys (bounds the solver DAG depth)."""
    return np.argsort(splitmix64(seed, n, stream), kind="stable")


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


def sphere_pile(nx, ny, nz, seed=SEED, iters=10, shuffle=True, x_offset=0.0):
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
        perm = seeded_permutation(seed, n, stream=7)
        c, v0 = c[perm], v0[perm]
    half = max(nx, nz) / 2.0 + 1.0
    terrain = box_terrain(half, ny + 2.0, (x_offset, 0.0, 0.0))
    return _scene(f"sphere_pile_{nx}x{ny}x{nz}", _spheres(c, 0.5), terrain, v0=v0, iters=iters)


def sphere_pile_tile(nx, ny, nz, rank, world_size, seed=SEED, iters=10):
    """x-slab tile `rank` of a (world_size*nx) x ny x nz pile in ONE open box (BASELINE config 4 family;
    world_size == 1 is sphere_pile).  Tile r owns lattice columns [r*nx, (r+1)*nx); its bodies, jitter and
    velocities come from per-tile SplitMix64 streams so every rank can build its own tile independently."""
    if world_size == 1:
        sc = sphere_pile(nx, ny, nz, seed=seed, iters=iters)
        sc["x_range"] = (-np.inf, np.inf)
        return sc
    n = nx * ny * nz
    gx = world_size * nx
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([i.ravel() + rank * nx - (gx - 1) / 2.0, j.ravel() + 0.5, k.ravel() - (nz - 1) / 2.0], axis=1).astype(np.float32)
    st = 16 * (rank + 1)
    jit = np.stack([uniform(seed, n, -0.05, 0.05, stream=st + s) for s in (1, 2, 3)], axis=1)
    v0 = np.stack([uniform(seed, n, -1.0, 1.0, stream=st + s) for s in (4, 5, 6)], axis=1)
    c = (base + jit).astype(np.float32)
    perm = seeded_permutation(seed, n, stream=st + 7)
    c, v0 = c[perm], v0[perm]
    terrain = box_terrain(gx / 2.0 + 1.0, ny + 2.0, (0.0, 0.0, 0.0), half_z=nz / 2.0 + 1.0)
    sc = _scene(f"sphere_pile_tile{rank}of{world_size}_{nx}x{ny}x{nz}", _spheres(c, 0.5), terrain, v0=v0, iters=iters)
    sc["x_range"] = (rank * nx - gx / 2.0, (rank + 1) * nx - gx / 2.0)
    return sc


def config(idx):
    """BASELINE.json configs by index."""
    if idx == 0:
        return balls_demo(8)
    if idx == 1:
        return sphere_pile(64, 64, 64)
    raise NotImplementedError(f"config {idx}")
