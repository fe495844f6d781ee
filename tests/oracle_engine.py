"""CPU engine for mgf_amd.tiles (tests only): the oracle's tile mode behind the same interface as
mgf_amd.tiles.HipEngine, with torch CPU tensors as exchange buffers."""
import numpy as np
import torch

from oracle import oracle as O
from tests.util import oracle_world


class OracleEngine:
    def __init__(self, scene):
        self.w = oracle_world(scene, O.ORDER_CANONICAL)
        self.ids = [np.zeros(0, np.uint32), np.zeros(0, np.uint32)]

    def begin_tick(self, dt):
        self.w.begin_tick(dt)

    def select_boundary(self, x_left, x_right):
        l, r = self.w.select_boundary(float(x_left), float(x_right))
        self.ids = [l, r]
        return len(l), len(r)

    def export_bodies(self, side):
        return torch.from_numpy(self.w.export_bodies(self.ids[side]))

    def import_ghosts(self, recs):
        self.w.import_ghosts(recs.numpy())

    def collide(self, dt):
        st = self.w.collide(dt)
        return dict(n_constraints=st.n_constraints, n_terrain_constraints=st.n_terrain_constraints,
                    n_pair_candidates=st.n_pair_candidates, n_refits=st.n_refits)

    def solve(self, iters):
        self.w.solve(iters)
        return dict(solver_kernel_launches=0, ms_solve=0.0, ms_solver_kernels=0.0)

    def export_velocities(self, side):
        return torch.from_numpy(self.w.export_velocities(self.ids[side]))

    def import_ghost_velocities(self, vel):
        self.w.import_ghost_velocities(vel.numpy())

    def empty(self, width):
        return torch.empty((0, width), dtype=torch.float32)

    def cat(self, parts):
        return torch.cat(parts, dim=0)

    def state(self):
        return self.w.state()
