"""Vectorised reading of a `flow_trace` dump of the block-local solver (development aid; used by tools/r04_*.py).

The library (option flow_trace = 1, a synchronous mgf_world_step / mgf_world_solve) writes /tmp/mgf_flow_trace.bin - per (iteration,
constraint) the clock at which a serving lane took the node from the ready queue and the clock at which it released its successors
(100 MHz) - and /tmp/mgf_flow6_poll.bin - 16 words per block: polling statistics and the block's phase clocks.
"""
import numpy as np

TRACE = "/tmp/mgf_flow_trace.bin"
POLL = "/tmp/mgf_flow6_poll.bin"
WORDS = 32  # kF6TraceWords


def load(cons):
    """cons: World.constraints() of the traced tick (insertion order).  Returns a dict of arrays."""
    raw = np.fromfile(TRACE, dtype=np.uint64)
    C, iters, n_rank, nb, n_canon = int(raw[0]), int(raw[1]), int(raw[2]), int(raw[3]), int(raw[4])
    H = 8
    tr = raw[H:H + 2 * iters * C].reshape(iters, C, 2).astype(np.int64)
    rest = raw[H + 2 * iters * C:].view(np.uint32)
    rank = rest[:n_rank].astype(np.int64) if n_rank else None
    if n_canon:  # the stamps are indexed by constraint id; the caller's list is in insertion order: canon[r] = id
        canon = rest[n_rank:n_rank + C].astype(np.int64)
        tr = tr[:, canon, :]
    tr[:, :, 0] &= ~3
    A = cons["a"].astype(np.int64)
    B = cons["b"].astype(np.int64)
    assert len(A) == C
    poll = np.fromfile(POLL, dtype=np.uint64).reshape(-1, WORDS).astype(np.int64)
    return {"C": C, "iters": iters, "nb": nb, "seen": tr[:, :, 0], "done": tr[:, :, 1], "rank": rank, "A": A, "B": B, "poll": poll}


def predecessors(A, B):
    """pred[side][c], wrap[side][c]: the constraint before c on its body a (side 0) / b (side 1) in insertion order; the first of a
    body's chain wraps to the chain's last (previous iteration).  -1 where there is no body (Static b)."""
    C = len(A)
    body = np.concatenate([A, B])
    cid = np.concatenate([np.arange(C), np.arange(C)])
    side = np.concatenate([np.zeros(C, np.int64), np.ones(C, np.int64)])
    ok = body >= 0
    body, cid, side = body[ok], cid[ok], side[ok]
    o = np.lexsort((cid, body))
    body, cid, side = body[o], cid[o], side[o]
    first = np.ones(len(body), bool)
    first[1:] = body[1:] != body[:-1]
    last = np.ones(len(body), bool)
    last[:-1] = body[1:] != body[:-1]
    prev = np.empty(len(body), np.int64)
    prev[1:] = cid[:-1]
    # the chain's last entry, for every entry: index of the group's end
    grp_end = np.flatnonzero(last)
    grp_id = np.cumsum(first) - 1
    prev[first] = cid[grp_end[grp_id[first]]]
    pred = np.full((2, C), -1, np.int64)
    wrap = np.zeros((2, C), np.int64)
    pred[side, cid] = prev
    wrap[side, cid] = first.astype(np.int64)
    return pred, wrap


def analyse(T):
    """Critical path of the traced launch: hop counts and times by kind (inside a block / across a block face)."""
    C, iters, seen, done, A, B = T["C"], T["iters"], T["seen"], T["done"], T["A"], T["B"]
    pred, wrap = predecessors(A, B)
    blk = (T["rank"][A] // T["nb"]) if T["rank"] is not None and T["nb"] else np.zeros(C, np.int64)
    poll = T["poll"]
    nblk = int(blk.max()) + 1
    t_entry = poll[:nblk, 8]
    t0 = int(t_entry[t_entry > 0].min()) if (t_entry > 0).any() else int(seen.min())
    us = lambda x: (x - t0) * 0.01  # noqa: E731
    # per (r, c): the predecessor released last
    crit = np.full((iters, C), -1, np.int64)       # flat index r' * C + c' of the critical predecessor
    crit_done = np.full((iters, C), -1, np.int64)
    early_done = np.full((iters, C), -1, np.int64)  # release of the OTHER predecessor (-1: the node has one predecessor)
    for r in range(iters):
        for s in range(2):
            p, w = pred[s], wrap[s]
            rr = r - w
            ok = (p >= 0) & (rr >= 0)
            d = np.where(ok, done[np.clip(rr, 0, iters - 1), np.clip(p, 0, C - 1)], -1)
            better = d > crit_done[r]
            early_done[r] = np.where(better, crit_done[r], np.maximum(early_done[r], d))
            crit_done[r] = np.where(better, d, crit_done[r])
            crit[r] = np.where(better, rr * C + p, crit[r])
    lead = np.where(early_done >= 0, (crit_done - early_done) * 0.01, np.nan)  # us between a node's first and last arrival
    # walk back from the node released last
    r, c = np.unravel_index(int(np.argmax(done)), done.shape)
    kinds = {"in_block": [0, 0.0, 0.0], "cross_block": [0, 0.0, 0.0]}
    hops_by_iter = np.zeros(iters, np.int64)
    path_lead = []
    while crit[r, c] >= 0:
        path_lead.append(lead[r, c])
        pr, pc = divmod(int(crit[r, c]), C)
        k = "in_block" if blk[pc] == blk[c] else "cross_block"
        e = kinds[k]
        e[0] += 1
        e[1] += (seen[r, c] - done[pr, pc]) * 0.01
        e[2] += (done[r, c] - seen[r, c]) * 0.01
        hops_by_iter[r] += 1
        r, c = pr, pc
    first_seen = us(seen[r, c])
    out = {"constraints": C, "iters": iters, "blocks": nblk, "bodies_per_block": T["nb"],
           "span_us": round(float(us(done.max())), 1),
           "first_node_taken_us": round(float(first_seen), 1),
           "hops": {k: {"n": v[0], "handoff_us": round(v[1] / max(v[0], 1), 2), "service_us": round(v[2] / max(v[0], 1), 2),
                        "total_us": round(v[1] + v[2], 1)} for k, v in kinds.items()},
           "hops_total": int(sum(v[0] for v in kinds.values())),
           "hops_by_iteration": hops_by_iter.tolist(),
           "iteration_complete_us": [round(float(us(done[q].max())), 1) for q in range(iters)],
           "node_service_us_mean": round(float(np.mean(done - seen)) * 0.01, 2)}
    # how long before it becomes ready a node's record could be fetched: the time between its first and its last arrival
    pl = np.array(path_lead)
    two = ~np.isnan(pl)
    out["lead_us_on_path"] = {"nodes_with_two_preds": int(two.sum()), "of": len(pl),
                              "p25": round(float(np.percentile(pl[two], 25)), 2) if two.any() else None,
                              "p50": round(float(np.percentile(pl[two], 50)), 2) if two.any() else None,
                              "ge_1us_frac": round(float((pl[two] >= 1.0).mean()), 3) if two.any() else None}
    al = lead[~np.isnan(lead)]
    out["lead_us_all_nodes"] = {"p25": round(float(np.percentile(al, 25)), 2), "p50": round(float(np.percentile(al, 50)), 2), "ge_1us_frac": round(float((al >= 1.0).mean()), 3)}
    # the longest chain of one body: a lower bound of the unrolled graph's depth is iters x that
    cnt = np.bincount(np.concatenate([A, B[B >= 0]]))
    out["longest_body_chain"] = int(cnt.max())
    out["depth_lower_bound_hops"] = int(cnt.max()) * iters
    # edges across a block face, share of all dependency edges of one iteration
    cross = 0
    tot = 0
    for s in range(2):
        ok = pred[s] >= 0
        cross += int((blk[pred[s][ok]] != blk[np.flatnonzero(ok)]).sum())
        tot += int(ok.sum())
    out["edges_cross_block_frac"] = round(cross / max(tot, 1), 4)
    # the blocks' phases (clocks 8..12: entry, bodies in LDS, tables in LDS, serving loop left, written back)
    if (t_entry > 0).any():
        ph = poll[:nblk, 8:13].astype(np.float64)
        out["block_phases_us"] = {
            "entry_spread": round(float((ph[:, 0].max() - ph[:, 0].min()) * 0.01), 2),
            "bodies_to_lds_mean": round(float(np.mean(ph[:, 1] - ph[:, 0]) * 0.01), 2),
            "tables_to_lds_mean": round(float(np.mean(ph[:, 2] - ph[:, 1]) * 0.01), 2),
            "prologue_max_end": round(float(us(ph[:, 2].max())), 2),
            "serving_mean": round(float(np.mean(ph[:, 3] - ph[:, 2]) * 0.01), 2),
            "serving_max_end": round(float(us(ph[:, 3].max())), 2),
            "write_back_mean": round(float(np.mean(ph[:, 4] - ph[:, 3]) * 0.01), 2),
            "last_block_done": round(float(us(ph[:, 4].max())), 2)}
    if poll.shape[1] >= 21 and poll[:nblk, 17].sum() > 0:  # MGF_F6_PROFILE build: shader clocks of wave 0's trips, by section
        trips = float(poll[:nblk, 17].sum())
        out["trip_profile_clocks"] = {"trips_wave0_per_block": round(trips / nblk, 1), "nodes_per_trip": round(float(poll[:nblk, 18].sum()) / trips, 2),
                                      "pop": round(float(poll[:nblk, 13].sum()) / trips, 1), "loads": round(float(poll[:nblk, 14].sum()) / trips, 1),
                                      "solve_and_store": round(float(poll[:nblk, 15].sum()) / trips, 1), "release": round(float(poll[:nblk, 16].sum()) / trips, 1),
                                      "idle_polls_per_block": round(float(poll[:nblk, 20].sum()) / nblk, 1),
                                      "idle_poll_clocks": round(float(poll[:nblk, 19].sum()) / max(float(poll[:nblk, 20].sum()), 1.0), 1)}
    if (poll[:nblk, 23] > 0).any():
        dt = (poll[:nblk, 11] - poll[:nblk, 8]).astype(np.float64)
        out["shader_clock_mhz"] = round(float(np.mean(poll[:nblk, 23] / np.maximum(dt, 1.0)) * 100.0), 1)
    Nb = np.bincount(blk, minlength=nblk)
    out["constraints_per_block"] = {"mean": round(float(Nb.mean()), 1), "max": int(Nb.max())}
    sweeps = poll[:nblk, 0].astype(np.float64)
    if sweeps.sum() > 0:
        serving = (poll[:nblk, 11] - poll[:nblk, 10]).astype(np.float64) * 0.01  # (a block's polling wave sweeps while the block serves)
        out["polling"] = {"sweep_period_us": round(float(np.mean(serving / np.maximum(sweeps, 1.0))), 2),
                          "messages_per_block": round(float(poll[:nblk, 1].mean()), 1),
                          "message_latency_us": round(float(poll[:nblk, 2].sum() / max(poll[:nblk, 1].sum(), 1) * 0.01), 2),
                          "incoming_channels_mean": round(float(poll[:nblk, 5].mean()), 1),
                          "latency_hist_lt_1_2_3_4_6_more_us": [round(float(poll[:nblk, 24 + k].sum() / max(poll[:nblk, 1].sum(), 1)), 3) for k in range(6)],
                          "seen_by_quiet_sweeps_frac": round(float(poll[:nblk, 30].sum() / max(poll[:nblk, 1].sum(), 1)), 3)}
    return out
