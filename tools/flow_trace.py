"""Critical-path breakdown of the dataflow solver from per-node timestamps (development aid)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(n, n, n)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option('solver_mode', mode)
import os
for kv in [kv.split('=') for kv in os.environ.get('MGF_F6_OPTS', '').split(',') if kv]: w.set_option(kv[0], int(kv[1]))
for s in range(ticks): w.step(float(sc['dt']), 10)
w.set_option('flow_trace', 1)
w.step(float(sc['dt']), 10)
w.set_option('flow_trace', 0)
cons = w.constraints()
raw = np.fromfile('/tmp/mgf_flow_trace.bin', dtype=np.uint64)
C, iters, n_rank, nb_block = int(raw[0]), int(raw[1]), int(raw[2]), int(raw[3])
tr = raw[4:4 + 2 * iters * C].reshape(iters, C, 2).astype(np.int64)
rank = raw[4 + 2 * iters * C:].view(np.uint32)[:n_rank].astype(np.int64) if n_rank else None
cls = (tr[0, :, 0] & 3) if mode in (5, 6) else np.zeros(C, np.int64)   # mode 5 stamps the slot class into the low bits
if mode in (5, 6):
    tr[:, :, 0] &= ~3
    print("classes (0 all-LDS, 1 global counter, 2 shared body):", np.bincount(cls, minlength=3) / C)
t0 = tr[:, :, 0].min()
seen = (tr[:, :, 0] - t0) * 0.01   # us (100 MHz)
done = (tr[:, :, 1] - t0) * 0.01
print(f"C={C} iters={iters} span={done.max():.1f} us; node service (seen->released) mean {np.mean(done-seen):.2f} us p50 {np.median(done-seen):.2f} p90 {np.percentile(done-seen,90):.2f}")
A = cons['a'].astype(np.int64); B = cons['b'].astype(np.int64)
# predecessors in the unrolled graph
nb = int(max(A.max(), B.max())) + 1
lists = [[] for _ in range(nb)]
for c in range(C):
    lists[A[c]].append(c)
    if B[c] >= 0: lists[B[c]].append(c)
preds = [[] for _ in range(C)]   # (pred constraint, wrap)
for l in lists:
    for k, c in enumerate(l):
        if k > 0: preds[c].append((l[k-1], 0))
        else: preds[c].append((l[-1], 1))
# hand-off latency: seen(node) - max(done(pred))
lat = []
crit_pred = {}
for r in range(iters):
    for c in range(C):
        best = -1.0; bp = None
        for (p, wrap) in preds[c]:
            rr = r - wrap
            if rr < 0: continue
            if done[rr, p] > best: best = done[rr, p]; bp = (rr, p)
        if bp is not None:
            lat.append(seen[r, c] - best); crit_pred[(r, c)] = bp
lat = np.array(lat)
print(f"hand-off (last pred released -> seen) mean {lat.mean():.2f} us p10 {np.percentile(lat,10):.2f} p50 {np.median(lat):.2f} p90 {np.percentile(lat,90):.2f} p99 {np.percentile(lat,99):.2f}")
# walk the critical path back from the last finished node
r, c = np.unravel_index(np.argmax(done), done.shape)
hops = 0; svc = 0.0; ho = 0.0
path = []
by = {}
while (r, c) in crit_pred:
    pr, pc = crit_pred[(r, c)]
    svc += done[r, c] - seen[r, c]; ho += seen[r, c] - done[pr, pc]; hops += 1
    path.append((seen[r, c] - done[pr, pc], done[r, c] - seen[r, c]))
    k = (int(cls[pc]), int(cls[c]))
    if rank is not None:
        k = k + ("same block" if rank[A[pc]] // nb_block == rank[A[c]] // nb_block else "other block",)
    e = by.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += seen[r, c] - done[pr, pc]; e[2] += done[r, c] - seen[r, c]
    r, c = pr, pc
if mode in (5, 6):
    for k in sorted(by):
        n, h, sv = by[k]
        print(f"  critical hops class {k[0]} -> {k[1]} {k[2] if len(k) > 2 else '':11s}: {n:4d}  hand-off {h / n:6.2f} us  service {sv / n:6.2f} us  (total {h + sv:7.1f} us)")
    for k in range(3):
        m = cls == k
        if m.any():
            print(f"  class {k}: service mean {np.mean((done - seen)[:, m]):.2f} us p50 {np.median((done - seen)[:, m]):.2f} p90 {np.percentile((done - seen)[:, m], 90):.2f}")
print(f"critical path: {hops} hops, service {svc:.1f} us ({svc/hops:.2f}/hop), hand-off {ho:.1f} us ({ho/hops:.2f}/hop), first node seen at {seen[r,c]:.1f} us")
h = np.array(path)
print("hand-off on the critical path: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(h[:,0],[10,50,90,100])))
print("service  on the critical path: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(h[:,1],[10,50,90,100])))

if mode == 6 and os.path.exists('/tmp/mgf_flow6_poll.bin'):
    st = np.fromfile('/tmp/mgf_flow6_poll.bin', dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    span = done.max()
    print(f"polling: sweeps/block mean {st[:,0].mean():.0f} (period {span / max(st[:,0].mean(),1):.2f} us), messages/block {st[:,1].mean():.0f}, "
          f"message latency sent->consumed mean {st[:,2].sum() / max(st[:,1].sum(),1) * 0.01:.2f} us, max {st[:,3].max() * 0.01:.1f} us, "
          f"full windows/block {st[:,4].mean():.1f}, incoming channels mean {st[:,5].mean():.1f} max {st[:,5].max():.0f}; "
          f"time a sweep waits for its loads {st[:,6].sum() / max(st[:,0].sum(),1) * 0.01:.2f} us")

# per block: constraints, when its last node was released - is the launch the slowest block's, and is that the fullest one?
if rank is not None and nb_block:
    blk = rank[A] // nb_block
    nblk = int(blk.max()) + 1
    Nb = np.bincount(blk, minlength=nblk)
    fin = np.zeros(nblk)
    np.maximum.at(fin, blk, done.max(axis=0))
    first = np.full(nblk, 1e9)
    np.minimum.at(first, blk, seen.min(axis=0))
    span = done.max()
    print(f"blocks {nblk}: constraints per block mean {Nb.mean():.0f} p10 {np.percentile(Nb,10):.0f} p50 {np.median(Nb):.0f} p90 {np.percentile(Nb,90):.0f} max {Nb.max()}")
    print("block finish time (us): p10 %.0f p25 %.0f p50 %.0f p75 %.0f p90 %.0f max %.0f" % tuple(np.percentile(fin, [10, 25, 50, 75, 90, 100])))
    cc = np.corrcoef(Nb, fin)[0, 1]
    print(f"correlation(constraints of a block, its finish time) = {cc:.2f}; blocks still running at 50 % / 75 % / 90 % of the span: "
          f"{int((fin > 0.5 * span).sum())} / {int((fin > 0.75 * span).sum())} / {int((fin > 0.9 * span).sum())}")
    # node throughput over time, whole chip: nodes released per us in ten slices of the span
    hist, _ = np.histogram(done.ravel(), bins=10, range=(0, span))
    print("nodes released per us, by tenth of the span:", " ".join(f"{h / (span / 10):.0f}" for h in hist))
    # per-iteration completion: when the last node of every iteration was released
    print("iteration r complete at (us):", " ".join(f"{done[r].max():.0f}" for r in range(iters)))
