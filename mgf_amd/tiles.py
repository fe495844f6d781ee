"""Spatial tiling of one scene across the GPUs of a node: x-slab tiles, one process
per GPU, ghost bodies exchanged with the two slab neighbours (point-to-point, no collective on the
data path).  The tick on every tile:

    begin_tick                      complete_motion + integrate the owned bodies
    select_boundary / export        owned bodies whose fat AABB reaches into the halo of a slab face
    <-> neighbours                  body records (36 floats each)
    import_ghosts, collide          broadphase / narrowphase / ContactConstraint::new on owned + ghost
    iters x { solve(1); <-> neighbours: velocities of the exported bodies (8 floats each) }

Semantics (what the oracle's tile mode reproduces exactly): Gauss-Seidel inside a tile, ghost
velocities refreshed from their owner after every solver iteration (block-Jacobi across tiles);
a constraint between bodies of two tiles exists on both tiles, each tile keeping the result for the
body it owns.  Ownership is by initial slab; a body drifting past the halo raises (migration is
future work).

The driver is transport- and engine-agnostic: `HipEngine` drives mgf_amd.World through the C-ABI
with torch CUDA tensors as exchange buffers (RCCL via torch.distributed); tests run the same driver
with a CPU engine over gloo.
"""
import numpy as np

from . import scenes

GHOST_FLOATS = 36
VEL_FLOATS = 8


class HipEngine:
    """One tile on one GPU.  Exchange buffers are torch CUDA tensors; the C-ABI gets raw pointers."""

    def __init__(self, ctx, scene, device):
        import torch
        from ._capi import World
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.world = World.from_scene(ctx, scene)
        n = len(self.world)
        self.ids = [torch.zeros(max(n, 1), dtype=torch.int32, device=self.device) for _ in range(2)]
        self.counts = [0, 0]

    def begin_tick(self, dt):
        self.world.begin_tick(dt)

    def select_boundary(self, x_left, x_right):
        n = len(self.world)
        self.counts = list(self.world.select_boundary(x_left, x_right, self.ids[0].data_ptr(), self.ids[1].data_ptr(), max(n, 1)))
        return tuple(self.counts)

    def export_bodies(self, side):
        m = self.counts[side]
        out = self.torch.empty((m, GHOST_FLOATS), dtype=self.torch.float32, device=self.device)
        self.world.export_bodies(self.ids[side].data_ptr(), m, out.data_ptr())
        return out

    def _sync_torch(self):
        # torch (cat, RCCL recv) runs on torch's stream, the C-ABI on its own: make the buffers final first
        self.torch.cuda.current_stream(self.device).synchronize()

    def import_ghosts(self, recs):
        recs = recs.contiguous()
        self._sync_torch()
        self.world.import_ghosts(recs.data_ptr(), recs.shape[0])

    def collide(self, dt):
        return self.world.collide(dt).as_dict()

    def solve(self, iters):
        return self.world.solve(iters).as_dict()

    def export_velocities(self, side):
        m = self.counts[side]
        out = self.torch.empty((m, VEL_FLOATS), dtype=self.torch.float32, device=self.device)
        self.world.export_velocities(self.ids[side].data_ptr(), m, out.data_ptr())
        return out

    def import_ghost_velocities(self, vel):
        vel = vel.contiguous()
        self._sync_torch()
        self.world.import_ghost_velocities(vel.data_ptr(), vel.shape[0])

    def empty(self, width):
        return self.torch.empty((0, width), dtype=self.torch.float32, device=self.device)

    def cat(self, parts):
        return self.torch.cat(parts, dim=0)

    def state(self):
        return self.world.state()


class DistTransport:
    """Neighbour exchange over torch.distributed point-to-point ops (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, dist, rank, world_size, torch_device, host_staging=False):
        import torch
        self.dist, self.rank, self.world_size = dist, rank, world_size
        self.torch = torch
        self.out_dev = torch_device
        # host_staging: move payloads through CPU tensors (gloo has no CUDA point-to-point); used to
        # validate the multi-rank flow on a single GPU.  RCCL exchanges device buffers directly.
        self.dev = torch.device("cpu") if host_staging else torch_device

    def exchange(self, send_left, send_right, width, recv_counts=None):
        """Send row blocks to the left/right neighbour, receive theirs.  Returns (from_left, from_right).
        recv_counts = (n_from_left, n_from_right) when already known (velocity refresh: one row per ghost),
        which saves the count round-trip."""
        torch, dist = self.torch, self.dist
        send_left, send_right = send_left.to(self.dev), send_right.to(self.dev)
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank + 1 < self.world_size else None
        if recv_counts is None:
            # 1. row counts
            cnt_send = torch.tensor([send_left.shape[0], send_right.shape[0]], dtype=torch.int64, device=self.dev)
            cnt_recv = torch.zeros(2, dtype=torch.int64, device=self.dev)
            ops = []
            if left is not None:
                ops += [dist.P2POp(dist.isend, cnt_send[0:1], left), dist.P2POp(dist.irecv, cnt_recv[0:1], left)]
            if right is not None:
                ops += [dist.P2POp(dist.isend, cnt_send[1:2], right), dist.P2POp(dist.irecv, cnt_recv[1:2], right)]
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
            nl, nr = (int(v) for v in cnt_recv.tolist())
        else:
            nl, nr = recv_counts
        # 2. payloads
        from_left = torch.empty((nl, width), dtype=torch.float32, device=self.dev)
        from_right = torch.empty((nr, width), dtype=torch.float32, device=self.dev)
        ops = []
        if left is not None:
            if send_left.shape[0]:
                ops.append(dist.P2POp(dist.isend, send_left.contiguous(), left))
            if nl:
                ops.append(dist.P2POp(dist.irecv, from_left, left))
        if right is not None:
            if send_right.shape[0]:
                ops.append(dist.P2POp(dist.isend, send_right.contiguous(), right))
            if nr:
                ops.append(dist.P2POp(dist.irecv, from_right, right))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return from_left.to(self.out_dev), from_right.to(self.out_dev)


class NullTransport:
    def exchange(self, send_left, send_right, width, recv_counts=None):
        return send_left[:0], send_right[:0]


class Tile:
    """One tile's tick, split into phases so several tiles can also be stepped in one process."""

    def __init__(self, engine, x_range, rank, world_size, dt, iters, halo=1.0):
        self.e, self.rank, self.world_size = engine, rank, world_size
        self.x_lo, self.x_hi = x_range
        self.dt, self.iters, self.halo = float(dt), int(iters), float(halo)
        self.has_left, self.has_right = rank > 0, rank + 1 < world_size

    def phase_begin(self):
        e = self.e
        e.begin_tick(self.dt)
        x_left = self.x_lo + self.halo if self.has_left else -np.inf
        x_right = self.x_hi - self.halo if self.has_right else np.inf
        e.select_boundary(np.float32(max(x_left, -3.0e38)), np.float32(min(x_right, 3.0e38)))
        return (e.export_bodies(0) if self.has_left else e.empty(GHOST_FLOATS),
                e.export_bodies(1) if self.has_right else e.empty(GHOST_FLOATS))

    def phase_collide(self, from_left, from_right):
        self.e.import_ghosts(self.e.cat([from_left, from_right]))
        return self.e.collide(self.dt)

    def phase_solve_one(self):
        st = self.e.solve(1)
        return st, (self.e.export_velocities(0) if self.has_left else self.e.empty(VEL_FLOATS),
                    self.e.export_velocities(1) if self.has_right else self.e.empty(VEL_FLOATS))

    def phase_refresh(self, from_left, from_right):
        self.e.import_ghost_velocities(self.e.cat([from_left, from_right]))


def step_tile(tile, transport):
    """One tick of one tile with a real transport (one process per tile)."""
    sl, sr = tile.phase_begin()
    fl, fr = transport.exchange(sl, sr, GHOST_FLOATS)
    ghosts = (fl.shape[0], fr.shape[0])
    stats = tile.phase_collide(fl, fr)
    launches, ms_solve, ms_kern = 0, 0.0, 0.0
    for it in range(tile.iters):
        st, (vl, vr) = tile.phase_solve_one()
        launches += st["solver_kernel_launches"]
        ms_solve += st["ms_solve"]
        ms_kern += st.get("ms_solver_kernels", 0.0)
        if it + 1 < tile.iters and tile.world_size > 1:
            fl, fr = transport.exchange(vl, vr, VEL_FLOATS, recv_counts=ghosts)
            tile.phase_refresh(fl, fr)
    stats = dict(stats)
    stats.update(solver_kernel_launches=launches, ms_solve=ms_solve, ms_solver_kernels=ms_kern, n_levels=launches)
    return stats


def step_tiles_inprocess(tiles):
    """All tiles of a scene in ONE process (tests / emulation): same phases, exchange by hand."""
    P = len(tiles)
    sends = [t.phase_begin() for t in tiles]
    stats = []
    for r, t in enumerate(tiles):
        fl = sends[r - 1][1] if r > 0 else t.e.empty(GHOST_FLOATS)
        fr = sends[r + 1][0] if r + 1 < P else t.e.empty(GHOST_FLOATS)
        stats.append(t.phase_collide(fl, fr))
    for it in range(tiles[0].iters):
        vels = [t.phase_solve_one()[1] for t in tiles]
        if it + 1 < tiles[0].iters and P > 1:
            for r, t in enumerate(tiles):
                fl = vels[r - 1][1] if r > 0 else t.e.empty(VEL_FLOATS)
                fr = vels[r + 1][0] if r + 1 < P else t.e.empty(VEL_FLOATS)
                t.phase_refresh(fl, fr)
    return stats


class TiledWorld:
    """bench.py's view: this rank's tile of the BASELINE sphere-pile workload."""

    def __init__(self, ctx, rank, world_size, nx, ny, nz, iters=10, dist=None, device=0, seed=scenes.SEED, halo=1.0,
                 host_staging=False):
        self.rank, self.world_size, self.dist = rank, world_size, dist
        self.iters = iters
        self.scene = scenes.sphere_pile_tile(nx, ny, nz, rank, world_size, seed=seed, iters=iters)
        self.dt = float(self.scene["dt"])
        if world_size == 1:
            from ._capi import World
            self.world = World.from_scene(ctx, self.scene)
            self.tile = None
        else:
            import torch
            eng = HipEngine(ctx, self.scene, device)
            self.world = eng.world
            self.tile = Tile(eng, self.scene["x_range"], rank, world_size, self.dt, iters, halo=halo)
            self.transport = DistTransport(dist, rank, world_size, torch.device("cuda", device), host_staging=host_staging)

    def step(self):
        if self.tile is None:
            return self.world.step(self.dt, self.iters).as_dict()
        return step_tile(self.tile, self.transport)
