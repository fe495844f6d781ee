"""Spatial tiling of one scene across the GPUs of a node: x-slab tiles, one process
per GPU, ghost bodies exchanged with the two slab neighbours (point-to-point, no collective on the
data path).  The tick on every tile:

    begin_tick                      complete_motion + integrate the owned bodies
    select_boundary / export        owned bodies whose fat AABB reaches into the halo of a slab face
    <-> neighbours                  body records (72 floats each)
    import_ghosts, collide          broadphase / narrowphase / ContactConstraint::new on owned + ghost
    iters/R x { solve(R); <-> neighbours: velocities of the exported bodies (8 floats each) }
    finish                          the one place the host waits for the solver (status, timings)
    migrate (only when needed)      owned bodies whose centre left the slab: full records (148 floats) to the
                                    neighbour, removed here, appended there - selected at the start of the tick
                                    (the counts ride on the ghost count message), moved at its end

Semantics (what the oracle's tile mode reproduces exactly): Gauss-Seidel inside a tile, ghost
velocities refreshed from their owner after every R solver iterations (block-Jacobi across tiles);
a constraint between bodies of two tiles exists on both tiles, each tile keeping the result for the
body it owns.  A body belongs to the tile whose slab [x_lo, x_hi) holds its centre: it is handed over
at the end of the first tick that starts with the centre outside (so it spends at most that one tick on
the wrong side, well inside the halo).  Arrivals are appended in the order left neighbour's, then right
neighbour's; the remaining bodies keep their relative order.  Body identity across tiles is the 32-bit
tag (`scene["tags"]`, default: the body's index in the tile's scene).

Process set-up note: `import torch` must happen before the first mgf_amd.Context (torch ships its own
libamdhip64 with the same soname as /opt/rocm's; the first one loaded serves the whole process).

The driver is transport- and engine-agnostic: `HipEngine` drives mgf_amd.World through the C-ABI
with torch CUDA tensors as exchange buffers (RCCL via torch.distributed); tests run the same driver
with a CPU engine over gloo.
"""
import numpy as np

from . import scenes

GHOST_FLOATS = 72
VEL_FLOATS = 8
MIGRANT_FLOATS = 148
# Solver iterations between two ghost velocity refreshes (1 = refresh after every iteration).  Measured on a
# two-tile 6x6x6 pile after 60 ticks (tests/test_tiles_cpu.py::test_seam_quality_vs_refresh_interval): mean
# resting penetration of the sphere pairs straddling the slab face 0.043 (R=1), 0.046 (R=2), 0.055 (R=10)
# against 0.044 for pairs inside a tile - R=2 halves the exchanges and solver launches and keeps the seam at
# the interior's level.
# (r06) 4: measured at full size (BASELINE config 4 as 8 tiles, the pile collapsing and at rest - ticks 260, 380, 460 - and config 5's two-part
# bodies, EXPERIMENTS.md round 6): the depth of the pairs across tile faces at R = 4 is within 6 % of R = 2's (p99; mean within 3 %) and below
# the interior's; R = 5 is 30-60 % deeper, R = 3 no faster than 2.  Ten iterations are then three solver launches (4 + 4 + 2) and two
# velocity exchanges instead of five and four: the tile tick -5 % (falling) to -10 % (at rest).
DEFAULT_REFRESH_EVERY = 4


class HipEngine:
    """One tile on one GPU.  Exchange buffers are torch CUDA tensors; the C-ABI gets raw pointers.

    Everything - the C-ABI's kernels, torch's copies, and the RCCL transfers torch.distributed orders against
    the current stream - is issued on ONE torch stream (mgf_ctx_set_stream + option stream_ordered), so a tick
    needs no host synchronisation between its exchange steps: the host only waits where it needs a number
    (boundary counts, the collide phase's list sizes) and once at the end (finish)."""

    def __init__(self, ctx, scene, device):
        import torch
        from ._capi import World
        self.torch = torch
        self.device = torch.device("cuda", device)
        # one stream per context: tiles that share a context (in-process emulation) share its stream
        if getattr(ctx, "torch_stream", None) is None:
            ctx.torch_stream = torch.cuda.Stream(device=self.device)
            ctx.set_stream(ctx.torch_stream.cuda_stream)
        self.stream = ctx.torch_stream
        self.world = World.from_scene(ctx, scene)
        self.world.set_option("stream_ordered", 1)
        if scene.get("tags") is not None:
            self.world.set_tags(scene["tags"])
        self.n_cap = 0
        self._ensure_ids()
        self.counts = (0, 0)
        self.migrants = (0, 0)

    def _ensure_ids(self):
        """Id lists sized for the current body count: [0:cap] left face, [cap:2cap] right face, [2cap:3cap] migrants."""
        n = len(self.world)
        if n <= self.n_cap:
            return
        self.n_cap = n + n // 4 + 64
        with self.torch.cuda.stream(self.stream):
            self.ids = self.torch.zeros(3 * self.n_cap, dtype=self.torch.int32, device=self.device)

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def alloc(self, rows, width):
        return self.torch.empty((rows, width), dtype=self.torch.float32, device=self.device)

    def begin_tick(self, dt):
        self.world.begin_tick(dt)

    def select_tile(self, x_left, x_right, x_lo, x_hi):
        p = self.ids.data_ptr()
        c = self.world.select_tile(x_left, x_right, x_lo, x_hi, p, p + 4 * self.n_cap, p + 8 * self.n_cap, self.n_cap)
        self.counts, self.migrants = c[:2], c[2:]
        return c

    def export_migrants(self):
        m = sum(self.migrants)
        out = self.alloc(m, MIGRANT_FLOATS)
        if m:
            self.world.export_migrants(self.ids.data_ptr() + 8 * self.n_cap, m, out.data_ptr())
        return out

    def apply_migration(self, arrivals):
        """Drop the bodies selected by the last select_tile, append the neighbours' (left neighbour's first)."""
        m = sum(self.migrants)
        if m:
            self.world.remove_bodies(self.ids.data_ptr() + 8 * self.n_cap, m)
        if arrivals.shape[0]:
            self._keep_m = arrivals
            self.world.import_migrants(arrivals.data_ptr(), arrivals.shape[0])
        self.migrants = (0, 0)
        self._ensure_ids()

    def kinds(self):
        return self.world.counter("body_kinds")

    def add_kinds(self, mask):
        self.world.set_option("body_kinds", mask)

    def tags(self):
        return self.world.tags()

    def _export(self, fn, width):
        ml, mr = self.counts
        out = self.alloc(ml + mr, width)
        p = self.ids.data_ptr()
        if ml:
            fn(p, ml, out.data_ptr())
        if mr:
            fn(p + 4 * self.n_cap, mr, out.data_ptr() + 4 * width * ml)
        return out

    def export_bodies(self):
        return self._export(self.world.export_bodies, GHOST_FLOATS)

    def import_ghosts(self, recs):
        self._keep = recs  # the kernel reads it asynchronously (same stream: safe against reuse, keep it anyway)
        self.world.import_ghosts(recs.data_ptr(), recs.shape[0])

    def collide(self, dt):
        return self.world.collide(dt).as_dict()

    def max_fat_half_extent_x(self):
        return self.world.counter("max_fat_half_extent_x_milli") / 1000.0

    def solve_iterations(self, k):
        self.world.solve_enqueue(k)

    def export_velocities(self):
        return self._export(self.world.export_velocities, VEL_FLOATS)

    def import_ghost_velocities(self, vel):
        self._keep_v = vel
        self.world.import_ghost_velocities(vel.data_ptr(), vel.shape[0])

    def finish(self):
        return self.world.finish().as_dict()

    def state(self):
        return self.world.state()


class DistTransport:
    """Neighbour exchange over torch.distributed point-to-point ops (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, dist, rank, world_size, host_staging=False):
        import torch
        self.dist, self.rank, self.world_size = dist, rank, world_size
        self.torch = torch
        # host_staging: move payloads through CPU tensors (gloo has no CUDA point-to-point); used to
        # validate the multi-rank flow on a single GPU.  RCCL exchanges device buffers directly.
        self.host_staging = host_staging
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank + 1 < world_size else None

    def _batch(self, ops):
        if ops:
            for r in self.dist.batch_isend_irecv(ops):
                r.wait()  # RCCL: orders the current stream after the transfer, does not block the host

    def exchange(self, send, split, width, alloc, recv_counts=None, extra=None):
        """`send` holds split[0] rows for the left neighbour followed by split[1] rows for the right one.
        Returns the rows received: the left neighbour's first.  recv_counts = (n_from_left, n_from_right)
        when already known (velocity refresh: one row per ghost), which saves the count round-trip.
        extra = (ints for the left neighbour, ints for the right one) rides on the count message; the
        neighbours' are returned as a third value ((from left), (from right)), zeros where there is none."""
        torch, dist = self.torch, self.dist
        ml, mr = split
        if self.host_staging:
            out_alloc, alloc = alloc, (lambda rows, w: torch.empty((rows, w), dtype=torch.float32))
            send = send.cpu()
        got_extra = None
        if recv_counts is None:
            dev = send.device
            xl, xr = (list(extra[0]), list(extra[1])) if extra is not None else ([], [])
            k = 1 + len(xl)
            cnt_send = torch.tensor([ml] + xl + [mr] + xr, dtype=torch.int64, device=dev)
            cnt_recv = torch.zeros(2 * k, dtype=torch.int64, device=dev)
            ops = []
            if self.left is not None:
                ops += [dist.P2POp(dist.isend, cnt_send[0:k], self.left), dist.P2POp(dist.irecv, cnt_recv[0:k], self.left)]
            if self.right is not None:
                ops += [dist.P2POp(dist.isend, cnt_send[k:], self.right), dist.P2POp(dist.irecv, cnt_recv[k:], self.right)]
            self._batch(ops)
            got = [int(v) for v in cnt_recv.tolist()]
            nl, nr = got[0], got[k]
            got_extra = (tuple(got[1:k]), tuple(got[k + 1:]))
        else:
            nl, nr = recv_counts
        recv = alloc(nl + nr, width)
        ops = []
        if self.left is not None:
            if ml:
                ops.append(dist.P2POp(dist.isend, send[:ml], self.left))
            if nl:
                ops.append(dist.P2POp(dist.irecv, recv[:nl], self.left))
        if self.right is not None:
            if mr:
                ops.append(dist.P2POp(dist.isend, send[ml:], self.right))
            if nr:
                ops.append(dist.P2POp(dist.irecv, recv[nl:], self.right))
        self._batch(ops)
        if self.host_staging:
            dev_recv = out_alloc(nl + nr, width)
            dev_recv.copy_(recv)
            recv = dev_recv
        if extra is not None:
            return recv, (nl, nr), got_extra
        return recv, (nl, nr)


class Tile:
    """One tile's tick, split into phases so several tiles can also be stepped in one process."""

    def __init__(self, engine, x_range, rank, world_size, dt, iters, halo=1.0, refresh_every=DEFAULT_REFRESH_EVERY, migrate=True):
        self.e, self.rank, self.world_size = engine, rank, world_size
        self.x_lo, self.x_hi = x_range
        self.dt, self.iters, self.halo = float(dt), int(iters), float(halo)
        self.has_left, self.has_right = rank > 0, rank + 1 < world_size
        self.refresh_every = max(1, int(refresh_every))
        self.migrate = bool(migrate)
        self.mig_split = (0, 0)  # bodies leaving to the left / right at the end of this tick
        self.n_migrated_out = self.n_migrated_in = 0

    def chunks(self):
        """Solver iterations between two ghost velocity refreshes: [R, R, ..., rest]."""
        left, out = self.iters, []
        while left > 0:
            out.append(min(self.refresh_every, left))
            left -= out[-1]
        return out

    def phase_begin(self):
        """-> (rows for [left | right], (n_left, n_right))"""
        e = self.e
        e.begin_tick(self.dt)
        big = 3.0e38
        x_left = self.x_lo + self.halo if self.has_left else -big
        x_right = self.x_hi - self.halo if self.has_right else big
        x_lo = self.x_lo if (self.has_left and self.migrate) else -big
        x_hi = self.x_hi if (self.has_right and self.migrate) else big
        c = e.select_tile(np.float32(x_left), np.float32(x_right), np.float32(x_lo), np.float32(x_hi))
        self.mig_split = tuple(c[2:])
        return e.export_bodies(), tuple(c[:2])

    def extra(self):
        """What rides on the ghost count message: (migrants going that way, kinds of this tile's bodies)."""
        k = self.e.kinds()
        return [self.mig_split[0], k], [self.mig_split[1], k]

    def phase_migrate_out(self):
        return self.e.export_migrants()

    def phase_migrate_in(self, arrivals):
        self.n_migrated_out += sum(self.mig_split)
        self.n_migrated_in += int(arrivals.shape[0])
        self.e.apply_migration(arrivals)
        self.mig_split = (0, 0)

    def phase_collide(self, ghosts):
        self.e.import_ghosts(ghosts)
        st = self.e.collide(self.dt)
        # A body is sent to the neighbour when its fat box comes within `halo` of the slab face; it can touch a body the
        # neighbour owns only through that body's fat half extent, which must not exceed the halo (else a contact across the
        # face is dropped silently).  Engines that know the extent report it.
        rmax = getattr(self.e, "max_fat_half_extent_x", lambda: 0.0)()
        if self.world_size > 1 and rmax > self.halo:
            raise ValueError(f"tiles: a body's fat half extent along x ({rmax:.3f}) exceeds the halo ({self.halo:.3f}): "
                             "contacts across tile faces would be missed - use a larger halo")
        return st

    def phase_solve(self, k):
        self.e.solve_iterations(k)
        return self.e.export_velocities()

    def phase_refresh(self, vel):
        self.e.import_ghost_velocities(vel)

    def phase_end(self):
        return self.e.finish()


def step_tile(tile, transport):
    """One tick of one tile with a real transport (one process per tile)."""
    e = tile.e
    with e.stream_ctx():
        send, split = tile.phase_begin()
        ghosts, counts, (from_l, from_r) = transport.exchange(send, split, GHOST_FLOATS, e.alloc, extra=tile.extra())
        arriving = (from_l[0] if from_l else 0, from_r[0] if from_r else 0)
        e.add_kinds((from_l[1] if from_l else 0) | (from_r[1] if from_r else 0))
        stats = dict(tile.phase_collide(ghosts))
        chunks = tile.chunks()
        for ci, k in enumerate(chunks):
            vel = tile.phase_solve(k)
            if ci + 1 < len(chunks) and tile.world_size > 1:
                got, _ = transport.exchange(vel, split, VEL_FLOATS, e.alloc, recv_counts=counts)
                tile.phase_refresh(got)
        fin = tile.phase_end()
        if sum(tile.mig_split) or sum(arriving):  # both ends of a hand-over know it from the count message
            out = tile.phase_migrate_out()
            got, _ = transport.exchange(out, tile.mig_split, MIGRANT_FLOATS, e.alloc, recv_counts=arriving)
            tile.phase_migrate_in(got)
    for k in ("solver_kernel_launches", "n_levels", "iters"):
        if k in fin:
            stats[k] = fin[k]
    for k in ("ms_solve", "ms_solver_kernels", "ms_total"):  # measured only with the options phase_timing / time_solver_kernels: 0 = not measured
        if fin.get(k):
            stats[k] = fin[k]
    return stats


def _gather_rows(tiles, sends, splits, r, cat, empty, width):
    """What tile r receives: the right-face rows of tile r-1, then the left-face rows of tile r+1."""
    parts = []
    if r > 0:
        parts.append(sends[r - 1][splits[r - 1][0]:])
    if r + 1 < len(tiles):
        parts.append(sends[r + 1][:splits[r + 1][0]])
    return cat(parts) if parts else empty(0, width)


def step_tiles_inprocess(tiles):
    """All tiles of a scene in ONE process (tests / emulation): same phases, exchange by hand."""
    import torch
    with tiles[0].e.stream_ctx():
        return _step_tiles_inprocess(tiles, torch)


def _step_tiles_inprocess(tiles, torch):
    P = len(tiles)
    begun = [t.phase_begin() for t in tiles]
    sends, splits = [b[0] for b in begun], [b[1] for b in begun]
    cat = lambda parts: torch.cat(parts, dim=0)  # noqa: E731
    kinds = [t.e.kinds() for t in tiles]
    for r, t in enumerate(tiles):
        t.e.add_kinds((kinds[r - 1] if r > 0 else 0) | (kinds[r + 1] if r + 1 < P else 0))
    stats = []
    for r, t in enumerate(tiles):
        stats.append(dict(t.phase_collide(_gather_rows(tiles, sends, splits, r, cat, t.e.alloc, GHOST_FLOATS))))
    chunks = tiles[0].chunks()
    for ci, k in enumerate(chunks):
        vels = [t.phase_solve(k) for t in tiles]
        if ci + 1 < len(chunks) and P > 1:
            for r, t in enumerate(tiles):
                t.phase_refresh(_gather_rows(tiles, vels, splits, r, cat, t.e.alloc, VEL_FLOATS))
    for r, t in enumerate(tiles):
        stats[r].update(t.phase_end())
    msplits = [t.mig_split for t in tiles]
    if any(sum(m) for m in msplits):
        outs = [t.phase_migrate_out() for t in tiles]
        for r, t in enumerate(tiles):
            t.phase_migrate_in(_gather_rows(tiles, outs, msplits, r, cat, t.e.alloc, MIGRANT_FLOATS))
    return stats


class TiledWorld:
    """bench.py's view: this rank's tile of the BASELINE sphere-pile workload."""

    def __init__(self, ctx, rank, world_size, nx, ny, nz, iters=10, dist=None, device=0, seed=scenes.SEED, halo=1.0,
                 host_staging=False, refresh_every=DEFAULT_REFRESH_EVERY, migrate=True):
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
            self.tile = Tile(eng, self.scene["x_range"], rank, world_size, self.dt, iters, halo=halo, refresh_every=refresh_every, migrate=migrate)
            self.transport = DistTransport(dist, rank, world_size, host_staging=host_staging)

    def step(self):
        if self.tile is None:
            return self.world.step(self.dt, self.iters)  # StepStats: indexable by field name
        return step_tile(self.tile, self.transport)
