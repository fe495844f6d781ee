"""Critical-path model of the block-local solver on a dumped constraint list (tools/mirror_sim_dump.py): today's scheme (a crossing
constraint runs in its body a's block, body b's velocity travels there and back) against mirrored constraints (a copy in
each block, one concurrent message each way).  Infinite serving waves, fixed latencies: local hand-off L, message M, service S (us)."""
import sys
import numpy as np

path = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L, M, S = 0.77, 4.38, 1.40
d = np.load(path)
a, b, x = d['a'].astype(np.int64), d['b'].astype(np.int64), d['x'].astype(np.float64)
n, C = len(x), len(a)
# blocks: equal-count boxes (x slabs, y rows, z runs), 1024 bodies each - what partition_order cuts
nb = 1024
B = n // nb
ext = x.max(0) - x.min(0)
side = (ext.prod() / B) ** (1 / 3)
fx = max(1, int(round(ext[0] / side))); fy = max(1, int(round(ext[1] / side)))
while B % fx: fx -= 1
while (B // fx) % fy: fy -= 1
fz = B // fx // fy
blk = np.zeros(n, np.int64)
ox = np.argsort(x[:, 0], kind='stable')
for ix, slab in enumerate(np.array_split(ox, fx)):
    oy = slab[np.argsort(x[slab, 1], kind='stable')]
    for iy, row in enumerate(np.array_split(oy, fy)):
        oz = row[np.argsort(x[row, 2], kind='stable')]
        for iz, run in enumerate(np.array_split(oz, fz)):
            blk[run] = (ix * fy + iy) * fz + iz
ba = blk[a]
bb = np.where(b >= 0, blk[np.maximum(b, 0)], -1)
cross = (b >= 0) & (ba != bb)
print(f"{path}: {n} bodies, {C} constraints, {B} blocks ({fx} x {fy} x {fz}), crossing constraints {cross.mean() * 100:.1f} %")

# chains: per body its constraints in insertion order (the list IS in insertion order)
order = np.arange(C)
def run(mirror):
    # end time of the copy of constraint c that updates body `side` (0: a, 1: b); without mirroring both sides share one copy
    end_a = np.zeros(C); end_b = np.zeros(C)
    last = np.full(n, -1, np.int64)         # last constraint executed on the body (previous in its chain, across iterations)
    last_side = np.zeros(n, np.int64)       # which side of that constraint the body was
    span = 0.0
    A, Bv, BA, BB, CR = a.tolist(), b.tolist(), ba.tolist(), bb.tolist(), cross.tolist()
    la, ls = last.tolist(), last_side.tolist()
    ea, eb = end_a.tolist(), end_b.tolist()
    for it in range(iters):
        for c in range(C):
            i, j = A[c], Bv[c]
            # predecessor on a's chain, on b's chain: (end time, block it ran in)
            def pred(body):
                p = la[body]
                if p < 0: return 0.0, -2
                if ls[body] == 0: return ea[p], (BA[p])
                return eb[p], (BB[p] if mirror else BA[p])
            ta, ga = pred(i)
            if j >= 0: tb, gb = pred(j)
            if not mirror or j < 0 or not CR[c]:
                g = BA[c]                       # one copy, in a's block
                st = ta + (L if ga in (g, -2) else M)
                if j >= 0: st = max(st, tb + (L if gb in (g, -2) else M))
                e = st + S
                ea[c] = e; eb[c] = e
            else:
                gA, gB = BA[c], BB[c]
                # copy in a's block: a's predecessor local (a's chain lives there), b's velocity by message from wherever b's chain is (its own block)
                e1 = max(ta + (L if ga in (gA, -2) else M), tb + (M if gb != -2 else L)) + S
                e2 = max(tb + (L if gb in (gB, -2) else M), ta + (M if ga != -2 else L)) + S
                ea[c] = e1; eb[c] = e2
            la[i] = c; ls[i] = 0
            if j >= 0: la[j] = c; ls[j] = 1
        span = max(max(ea), max(eb))
        print(f"  {'mirrored' if mirror else 'today   '} iteration {it}: complete at {span:8.1f} us", flush=True)
    return span
t0 = run(False)
t1 = run(True)
print(f"model span: today {t0:.1f} us, mirrored {t1:.1f} us  ({(1 - t1 / t0) * 100:.1f} % shorter)")
