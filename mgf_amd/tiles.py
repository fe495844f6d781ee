"""Spatial tiling of one scene across the GPUs of a node: x-slab tiles, one
process per GPU.  world_size == 1 is a plain World."""
import numpy as np

from . import scenes
from ._capi import World


class TiledWorld:
    def __init__(self, ctx, rank, world_size, nx, ny, nz, iters=10, dist=None, device=0, seed=scenes.SEED):
        self.rank, self.world_size, self.dist = rank, world_size, dist
        self.iters = iters
        if world_size != 1:
            raise NotImplementedError("multi-tile stepping lands with mgf_amd.halo")
        self.scene = scenes.sphere_pile(nx, ny, nz, seed=seed, iters=iters)
        self.dt = float(self.scene["dt"])
        self.world = World.from_scene(ctx, self.scene)

    def step(self):
        return self.world.step(self.dt, self.iters).as_dict()
