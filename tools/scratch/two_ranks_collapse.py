"""BASELINE config 4 through the pile's collapse (220 ticks, thousands of hand-overs per tick) as 2 ranks x 4 tiles on ONE device over the
stand-in transport (tests/fake_rccl) against the same 8 tiles in one process: the multi-rank path - counts, ghosts, velocity refreshes and
migrants across the RANK face - must give the single-process result bit for bit."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

def main():
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 220
    from tests.test_gpu_multi_device import _launch
    P, dims = 8, (16, 128, 64)
    res = _launch(2, P, dims, None, ticks, shared_device=True, timeout=900, world_opts={"flow_max_blocks": 128})
    got = {}
    for r in (0, 1):
        assert res[r]["failed_at"] is None, res[r]
        for t in res[r]["tiles"]:
            got[t["tile"]] = t
    print("two ranks:", [len(got[k]["tags"]) for k in range(P)], "bodies per tile; ticks retried", [res[r]["ticks_retried"] for r in (0, 1)],
          "bytes across the rank face", res[0]["bytes_out"], res[1]["bytes_out"], flush=True)
    import mgf_amd
    from mgf_amd import scenes
    ctx = mgf_amd.Context(0)
    scs = [scenes.sphere_pile_tile(*dims, r, P) for r in range(P)]
    worlds = []
    for sc in scs:
        w = mgf_amd.World.from_scene(ctx, sc); w.set_tags(sc["tags"]); worlds.append(w)
    T = mgf_amd.Tiles(ctx, worlds, [sc["x_range"] for sc in scs])
    dt, it = float(scs[0]["dt"]), scs[0]["iters"]
    for _ in range(ticks):
        T.step(dt, it)
    for k, w in enumerate(worlds):
        assert np.array_equal(w.tags(), got[k]["tags"]), f"tile {k}: bodies / order differ"
        st = w.state()
        for f in ("x", "q", "v", "omega"):
            assert np.array_equal(st[f].view(np.uint32), got[k][f].view(np.uint32)), f"tile {k}: {f} differs"
    print(f"one process, 8 tiles, {ticks} ticks: bit-identical to the two ranks'; hand-overs {sum(T.migrated(k) for k in range(P))}")

if __name__ == "__main__":
    main()
