"""CPU engine for mgf_amd.tiles (tests only): the oracle's tile mode behind the same interface as
mgf_amd.tiles.HipEngine, with torch CPU tensors as exchange buffers."""
import contextlib

import numpy as np
import torch

from oracle import oracle as O
from tests.util import oracle_world


class OracleEngine:
    def __init__(self, scene):
        self.w = oracle_world(scene, O.ORDER_CANONICAL)
        if scene.get("tags") is not None:
            self.w.set_tags(scene["tags"])
        self.ids = [np.zeros(0, np.uint32), np.zeros(0, np.uint32)]
        self.mig = [np.zeros(0, np.uint32), np.zeros(0, np.uint32)]

    def stream_ctx(self):
        return contextlib.nullcontext()

    def alloc(self, rows, width):
        return torch.empty((rows, width), dtype=torch.float32)

    def begin_tick(self, dt):
        self.w.begin_tick(dt)

    def select_tile(self, x_left, x_right, x_lo, x_hi):
        l, r = self.w.select_boundary(float(x_left), float(x_right))
        self.ids = [l, r]
        self.mig = list(self.w.select_migrants(float(x_lo), float(x_hi)))
        return len(l), len(r), len(self.mig[0]), len(self.mig[1])

    def export_migrants(self):
        W = self.w.MIGRANT_FLOATS
        return torch.from_numpy(np.concatenate([self.w.export_migrants(self.mig[0]).reshape(-1, W), self.w.export_migrants(self.mig[1]).reshape(-1, W)]))

    def apply_migration(self, arrivals):
        gone = np.sort(np.concatenate(self.mig)).astype(np.uint32)
        if len(gone):
            self.w.remove_bodies(gone)
        if arrivals.shape[0]:
            self.w.import_migrants(np.ascontiguousarray(arrivals.numpy()))
        self.mig = [np.zeros(0, np.uint32), np.zeros(0, np.uint32)]

    def kinds(self):
        return 3

    def add_kinds(self, mask):
        pass

    def tags(self):
        return self.w.tags()

    def export_bodies(self):
        return torch.from_numpy(np.concatenate([self.w.export_bodies(self.ids[0]).reshape(-1, O.GHOST_FLOATS), self.w.export_bodies(self.ids[1]).reshape(-1, O.GHOST_FLOATS)]))

    def import_ghosts(self, recs):
        self.w.import_ghosts(np.ascontiguousarray(recs.numpy()))

    def collide(self, dt):
        st = self.w.collide(dt)
        return dict(n_constraints=st.n_constraints, n_terrain_constraints=st.n_terrain_constraints,
                    n_pair_candidates=st.n_pair_candidates, n_refits=st.n_refits)

    def solve_iterations(self, k):
        self.w.solve(k)

    def export_velocities(self):
        return torch.from_numpy(np.concatenate([self.w.export_velocities(self.ids[0]).reshape(-1, 8), self.w.export_velocities(self.ids[1]).reshape(-1, 8)]))

    def import_ghost_velocities(self, vel):
        self.w.import_ghost_velocities(np.ascontiguousarray(vel.numpy()))

    def finish(self):
        return dict(solver_kernel_launches=0, ms_solve=0.0, ms_solver_kernels=0.0)

    def state(self):
        return self.w.state()
