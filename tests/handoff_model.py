"""A small-scope model of k_scan_one's in-launch hand-off (pigo_amd/csrc/pigo_kernels.hip.inc: one_push, one_consume,
one_poison, the last workgroup's clean-up), explored EXHAUSTIVELY over every interleaving of its agent-scope memory operations.

The protocol, as the kernel runs it (one of the eight queues; the others are copies):

  shared words (all accessed with agent-scope atomics / relaxed atomic loads and stores, never cached):
    work    items handed out            done    items finished          exited  workgroups that have left
    alloc   queue slots given to producers        head    queue slots claimed by consumers
    slot[i] = two 8-byte granules {A: valid | payload, B: tag}; each granule is written by ONE store, an entry counts when both
              are there (the consumer reads A, then B)

  workgroup:   loop { it = work++; if it >= nitems: break;
                      for every window the item leaves alive: s = alloc++; slot[s].B = tag; slot[s].A = entry;
                      (s_waitcnt vmcnt(0): the stores have been performed)
                      if ++done == nitems: POISON }
               then its waves become consumers; when all of them have left: if ++exited == grid: CLEAN-UP
  POISON:      al = alloc; h = head; for i in [al, h): slot[i] = poison           (`head` is read AFTER the `done` increment)
  consumer:    loop { i = head++; (the claim is PERFORMED -- its result is back -- before:) d = done;
                      if d >= nitems: al = alloc; if i >= al: leave
                      wait for slot[i]: entry -> zero the slot, finish the window, loop; poison -> zero the slot, leave }
  CLEAN-UP:    for i in [alloc, head): slot[i] = 0 (poison nobody took); every counter = 0

Two launches run back to back on the same memory (the second starts when the first has ended: a kernel boundary), because one
of the protocol's two historical bugs only shows in the NEXT launch.  Checked on every execution:
  * no execution gets stuck (a consumer waiting on a slot that nobody will ever fill);
  * every entry pushed in a launch is taken exactly once, in that launch;
  * after a launch every slot and every counter is zero.

`bug` re-introduces the two bugs round 5 found with a stress script (profiles/r05_experiments.md section 1):
  "claim_after_done"  a consumer's read of `done` may be performed before its claim (both were in flight together, without
                      the s_waitcnt): a claim that overtakes neither the finisher's increment nor its read of `head` waits on
                      a slot nobody poisons -- a hang;
  "no_cleanup"        the last workgroup does not zero [alloc, head): a consumer that claimed between the finisher's `done`
                      increment and its read of `head` left by itself, the finisher poisoned its slot all the same, and the
                      poison sends the NEXT launch's claimer of that slot away while a real entry for it is on its way -- a
                      window lost without a word.
Reference: core/pigo.go:212-258 (RunCascade returns every window with q > 0: none may be lost).
"""
from collections import deque

EMPTY, ENTRY, POISON = 0, 1, 2


class Config:
    def __init__(self, grid=2, waves=1, items=(1, 1), qcap=4, launches=2, bug=None, items2=None):
        """items[k] = entries item k pushes; items2: the same for the second launch (a frame with more survivors than the first)."""
        self.grid, self.waves, self.qcap, self.launches, self.bug = grid, waves, qcap, launches, bug
        self.items_by_launch = [tuple(items), tuple(items2 if items2 is not None else items)]
        assert len(self.items_by_launch[0]) == len(self.items_by_launch[1])
        self.nitems = len(self.items_by_launch[0])

    def items(self, launch):
        return self.items_by_launch[min(launch, 1)]


# A state is a tuple: (launch, work, done, exited, alloc, head, slotsA, slotsB, taken, procs)
#   slotsA / slotsB: tuples over the queue (granule A: EMPTY / ENTRY / POISON, granule B: 0 / 1 "tag present")
#   taken: entries taken in this launch (a count; the model's entries are indistinguishable, double takes cannot happen by
#          construction -- a consumer zeroes what it takes in the step it takes it -- so "exactly once" == taken == pushed)
#   procs: per workgroup (pc, a, b, waves...) -- see step()
def initial(cfg):
    wg = ("P0", 0, 0) + tuple(("W", 0, 0) for _ in range(cfg.waves))
    return (0, 0, 0, 0, 0, 0, (EMPTY,) * cfg.qcap, (0,) * cfg.qcap, 0, (wg,) * cfg.grid)


def successors(cfg, st):
    """Every state one atomic memory operation of one agent away.  A poll that finds nothing is no transition."""
    launch, work, done, exited, alloc, head, sa, sb, taken, procs = st
    out = []

    def put(procs2, **kw):
        d = dict(launch=launch, work=work, done=done, exited=exited, alloc=alloc, head=head, sa=sa, sb=sb, taken=taken)
        d.update(kw)
        out.append((d["launch"], d["work"], d["done"], d["exited"], d["alloc"], d["head"], d["sa"], d["sb"], d["taken"], procs2))

    def setp(g, wgstate):
        return procs[:g] + (wgstate,) + procs[g + 1:]

    def store(t, i, v):
        return t[:i] + (v,) + t[i + 1:]

    for g, wg in enumerate(procs):
        pc, a, b = wg[0], wg[1], wg[2]
        waves = wg[3:]
        # ---- the workgroup's producer side (thread 0 + the pushing waves, sequential inside a workgroup)
        if pc == "P0":  # it = work++
            it = work
            if it >= cfg.nitems:
                put(setp(g, ("C", 0, 0) + tuple(("C0", 0, 0) for _ in waves)), work=work + 1)
            else:
                put(setp(g, ("P1", it, cfg.items(launch)[it]) + waves), work=work + 1)
        elif pc == "P1":  # a = item, b = entries left to push: s = alloc++
            if b == 0:
                put(setp(g, ("P3", a, 0) + waves))
            else:
                put(setp(g, ("P2B", alloc, b) + waves), alloc=alloc + 1)
        elif pc == "P2B":  # slot[a].B = tag
            if a < cfg.qcap:
                put(setp(g, ("P2A", a, b) + waves), sb=store(sb, a, 1))
            else:
                put(setp(g, ("P1", 0, b - 1) + waves))  # (queue overflow: flagged, entry dropped -- not modelled further)
        elif pc == "P2A":  # slot[a].A = entry
            put(setp(g, ("P1", 0, b - 1) + waves), sa=store(sa, a, ENTRY))
        elif pc == "P3":  # d = ++done
            if done + 1 == cfg.nitems:
                put(setp(g, ("P4", 0, 0) + waves), done=done + 1)
            else:
                put(setp(g, ("P0", 0, 0) + waves), done=done + 1)
        elif pc == "P4":  # al = alloc
            put(setp(g, ("P5", min(alloc, cfg.qcap), 0) + waves))
        elif pc == "P5":  # h = head
            put(setp(g, ("P6", a, min(head, cfg.qcap)) + waves))
        elif pc == "P6":  # poison [a, b): one slot per step (B, then A)
            if a >= b:
                put(setp(g, ("P0", 0, 0) + waves))
            else:
                put(setp(g, ("P6A", a, b) + waves), sb=store(sb, a, 1))
        elif pc == "P6A":
            put(setp(g, ("P6", a + 1, b) + waves), sa=store(sa, a, POISON))
        # ---- its consumer waves
        elif pc == "C":
            all_out = all(w[0] == "X" for w in waves)
            if all_out:  # e = ++exited
                if exited + 1 == cfg.grid:
                    put(setp(g, ("E1", 0, 0) + waves), exited=exited + 1)
                else:
                    put(setp(g, ("OUT", 0, 0) + waves), exited=exited + 1)
            for k, w in enumerate(waves):
                wpc, wi, wd = w

                def setw(nw, **kw):
                    put(setp(g, (pc, a, b) + waves[:k] + (nw,) + waves[k + 1:]), **kw)

                if wpc == "C0":
                    if cfg.bug == "claim_after_done":
                        # the two operations are in flight together: either may be performed first
                        setw(("C1", head, 0), head=head + 1)       # claim first (then read done)
                        setw(("C0b", 0, done))                      # done first (then claim)
                    else:
                        setw(("C1", head, 0), head=head + 1)
                elif wpc == "C0b":  # (bug) the claim after the stale read of done
                    setw(("C2", head, wd), head=head + 1)
                elif wpc == "C1":  # d = done
                    setw(("C2", wi, done))
                elif wpc == "C2":
                    if wd >= cfg.nitems:
                        al = min(alloc, cfg.qcap)  # (one load)
                        if wi >= al:
                            setw(("X", 0, 0))
                        else:
                            setw(("C3", wi, 0))
                    elif wi >= cfg.qcap:
                        setw(("X", 0, 0))  # (overflow path: leaves when the items are done; not modelled further)
                    else:
                        setw(("C3", wi, 0))
                elif wpc == "C3":  # poll: A, then B
                    if sa[wi] != EMPTY and sb[wi] == 1:
                        kind = sa[wi]
                        sa2, sb2 = store(sa, wi, EMPTY), store(sb, wi, 0)  # (zeroing the slot: two stores nobody races with)
                        if kind == POISON:
                            setw(("X", 0, 0), sa=sa2, sb=sb2)
                        else:
                            setw(("C0", 0, 0), sa=sa2, sb=sb2, taken=taken + 1)
        elif pc == "E1":  # the workgroup that leaves last: zero the poison nobody took, then the counters
            sa2, sb2 = sa, sb
            if cfg.bug != "no_cleanup":
                for i in range(min(alloc, cfg.qcap), min(head, cfg.qcap)):
                    sa2, sb2 = store(sa2, i, EMPTY), store(sb2, i, 0)
            put(setp(g, ("OUT", 0, 0) + waves), sa=sa2, sb=sb2, work=0, done=0, exited=0, alloc=0, head=0)
    return out


def launch_over(st):
    return all(wg[0] == "OUT" for wg in st[9])


def explore(cfg, max_states=4_000_000):
    """Breadth-first over every interleaving.  Returns {"states": n, "violations": [...]} (a violation is a short description;
    the search stops at the first few)."""
    start = initial(cfg)
    seen = {start}
    todo = deque([start])
    violations = []
    while todo and len(violations) < 3:
        st = todo.popleft()
        if launch_over(st):
            launch, work, done, exited, alloc, head, sa, sb, taken, procs = st
            pushed_per_launch = sum(cfg.items(launch))
            if taken != pushed_per_launch:
                violations.append(f"launch {launch}: {pushed_per_launch} entries pushed, {taken} taken (a window lost)")
                continue
            if cfg.bug != "no_cleanup" and (any(sa) or any(sb) or work or done or exited or alloc or head):
                violations.append(f"launch {launch}: memory not clean at the end: A={sa} B={sb}")
                continue
            if launch + 1 < cfg.launches:
                nxt = (launch + 1, work, done, exited, alloc, head, sa, sb, 0, initial(cfg)[9])
                if nxt not in seen:
                    seen.add(nxt)
                    todo.append(nxt)
            continue
        succ = successors(cfg, st)
        if not succ:
            waiting = [(g, k, w) for g, wg in enumerate(st[9]) for k, w in enumerate(wg[3:]) if w[0] == "C3"]
            violations.append(f"launch {st[0]}: stuck -- consumers wait on slots nobody fills: {waiting}, A={st[6]}, alloc={st[4]}, head={st[5]}, done={st[2]}")
            continue
        for s2 in succ:
            if s2 not in seen:
                if len(seen) >= max_states:
                    raise RuntimeError("state space larger than expected")
                seen.add(s2)
                todo.append(s2)
    return {"states": len(seen), "violations": violations}
