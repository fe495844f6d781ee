#!/usr/bin/env python3
"""A static guard against the miscompile behind round 5's one device fault (EXPERIMENTS.md, round 6: "the fault of k_contacts_spheres, root cause").

The AMDGPU backend keeps a wave-uniform boolean (`ext != nullptr`, a kernel argument) in a VGPR as 0 / 1 per lane when it runs short of
scalar registers, and gets the lane mask back with `v_cmp_ne_u32 s[a:b], 1, vN` where it needs one - a compare that only sets the bits of the
lanes ACTIVE at that instruction.  Re-derived INSIDE a divergent loop (lanes leave through `s_andn2_b64 exec, exec, <done>`), the mask of the
last trip holds only the bits of the lanes that ran longest; read BEHIND the loop by `s_and_b64 vcc, exec, s[a:b]` + `s_cbranch_vccnz` under
an exec that holds other lanes, the "uniform" branch goes the wrong way.  (k_contacts_spheres of round 5: a lane with 13 row entries took the
`ext` path with ext = nullptr - a load from 4 x (partner id) - after a lane with 14 had run the pair loop's last trip alone.)

The script disassembles every gfx950 kernel of libmgf_hip.so, builds each kernel's control-flow graph and its loop nest (strongly connected
components, recursively without their headers), and reports every 0/1-VGPR compare that sits in a loop which lanes leave one by one and whose
mask reaches a reader outside that loop.  Usage: python tools/check_lane_masks.py [library]; exit status 1 if anything is found."""
import glob, os, re, shutil, subprocess, sys, tempfile

sys.setrecursionlimit(1000000)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def kernels(lib):
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(tmp, "l.so"))
        subprocess.run([LLVM + "llvm-objdump", "--offloading", "l.so"], cwd=tmp, capture_output=True)
        for f in sorted(glob.glob(os.path.join(tmp, "l.so.*gfx950"))):
            txt = subprocess.run([LLVM + "llvm-objdump", "-d", f], capture_output=True, text=True).stdout
            cur, body = None, []
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    if cur and body:
                        yield cur, body
                    cur, body = m.group(1), []
                    continue
                m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]{12}):", line)
                if cur and m:
                    body.append((int(m.group(2), 16), m.group(1)))
            if cur and body:
                yield cur, body
    finally:
        shutil.rmtree(tmp)


def sregs(tok):
    tok = tok.strip()
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(ins):
    parts = ins.split(None, 1)
    return [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []


def sccs(nodes, succ):
    """Tarjan, iterative; `nodes` a set, edges restricted to it"""
    index, low, on, stack, out, counter = {}, {}, set(), [], [], [0]
    for root in sorted(nodes):
        if root in index:
            continue
        work = [(root, iter([s for s in succ[root] if s in nodes]))]
        index[root] = low[root] = counter[0]; counter[0] += 1
        stack.append(root); on.add(root)
        while work:
            v, it = work[-1]
            advanced = False
            for w in it:
                if w not in index:
                    index[w] = low[w] = counter[0]; counter[0] += 1
                    stack.append(w); on.add(w)
                    work.append((w, iter([s for s in succ[w] if s in nodes])))
                    advanced = True
                    break
                if w in on:
                    low[v] = min(low[v], index[w])
            if advanced:
                continue
            work.pop()
            if work:
                low[work[-1][0]] = min(low[work[-1][0]], low[v])
            if low[v] == index[v]:
                comp = set()
                while True:
                    w = stack.pop(); on.discard(w); comp.add(w)
                    if w == v:
                        break
                if len(comp) > 1 or v in succ[v]:
                    out.append(comp)
    return out


def loop_nest(nodes, succ, pred):
    loops = []
    for comp in sccs(nodes, succ):
        loops.append(comp)
        headers = {v for v in comp if any(p not in comp for p in pred[v])} or {min(comp)}
        loops += loop_nest(comp - headers, succ, pred)
    return loops


def check(body):
    n = len(body)
    at = {a: k for k, (a, _) in enumerate(body)}
    succ, pred = [[] for _ in range(n)], [[] for _ in range(n)]
    for k, (a, ins) in enumerate(body):
        m = re.match(r"s_(cbranch_\w+|branch) (\d+)", ins)
        tgt = None
        if m:
            off = int(m.group(2)); off = off - 65536 if off >= 32768 else off
            tgt = at.get(a + 4 + 4 * off)
        if tgt is not None:
            succ[k].append(tgt)
        if not ins.startswith(("s_branch", "s_endpgm", "s_setpc")) and k + 1 < n:
            succ[k].append(k + 1)
    for k in range(n):
        for s in succ[k]:
            pred[s].append(k)
    bool_vgpr = set()
    cands = []
    for k, (a, ins) in enumerate(body):
        m = re.match(r"v_cndmask_b32_e64 (v\d+), 0, 1, ", ins)
        if m:
            bool_vgpr.add(m.group(1))
            continue
        m = re.match(r"v_cmp_(?:ne|eq)_u32_e64 (s\[\d+:\d+\]), 1, (v\d+)$", ins)
        if m and m.group(2) in bool_vgpr:
            cands.append((k, m.group(1)))
    if not cands:
        return []
    loops = loop_nest(set(range(n)), succ, pred)
    found = []
    for k, mask in cands:
        regs = sregs(mask)
        for L in loops:
            if k not in L:
                continue
            # lanes leave this loop one by one: an exit edge of L taken by `s_cbranch_execz / execnz` right behind `s_andn2_b64 exec, exec, ..`
            shrinking = False
            for v in L:
                ins = body[v][1]
                if ins.startswith("s_cbranch_exec") and any(s not in L for s in succ[v]) or (ins.startswith("s_cbranch_exec") and any(s in L and s <= v for s in succ[v])):
                    if any(body[u][1].startswith("s_andn2_b64 exec, exec,") for u in range(max(0, v - 3), v)):
                        shrinking = True
                        break
            if not shrinking:
                continue
            # does the mask written at k reach a reader outside L before it is written again?
            seen, work = set(), [s for s in succ[k]]
            while work:
                v = work.pop()
                if v in seen:
                    continue
                seen.add(v)
                ins = body[v][1]
                ops = operands(ins)
                reads = any(sregs(t) & regs for t in ops[1:])  # (the first operand is the destination)
                if reads and v not in L and not ins.startswith(("s_or_b64 exec, exec", "v_writelane")):
                    found.append((body[k][0], body[k][1], body[v][0], ins))
                    work = []
                    break
                writes = bool(ops) and bool(sregs(ops[0]) & regs) and not ins.startswith(("s_cbranch", "s_cmp", "s_bitcmp")) and v != k
                if writes:
                    continue
                work += succ[v]
            if found and found[-1][0] == body[k][0]:
                break
    return found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mgf_amd", "libmgf_hip.so")
    bad = n = 0
    for name, body in kernels(lib):
        n += 1
        for a, ins, a2, ins2 in check(body):
            bad += 1
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{dem[:100]}\n   {a:#x}: {ins}   (in a loop that lanes leave one by one)\n   {a2:#x}: {ins2}   (reads the mask outside that loop)")
    print(f"{n} kernels, {bad} lane masks derived inside a divergent loop and read outside it")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
