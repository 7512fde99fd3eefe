"""Host-side model of the synchronisation of k_mgm_bands (s2p_amd/csrc/mgm_bands.hpp): every WAVE of every band is a
process, its steps are events under a RANDOM scheduler; the shared state is what the kernel shares -- the LDS ring
chan[row][T & 7] of a band with the progress words, and the two-slot global row ring of tagged granules between
bands.  There is no barrier in the sweep and no flag between bands, so the model checks what must hold for ANY
interleaving:
  * a row reads, in step T, exactly what the row above wrote in step T - 1 (never an entry that was already rewritten
    eight steps later, never one not yet written) -- the data wait and the back-pressure wait of the kernel;
  * row 0 of band b consumes, for every point both rows have in the image, the granule band b - 1 stored for that very
    u -- never a granule of band b - 3 (same slot, other tag) or of band b + 1 (the slot's next user);
  * nobody waits forever.
The previous band's row is brought in by the band's FETCHER wave (a fifth process): groups of FP points are
requested ahead (the snapshot may be stale), whatever prefix of the oldest group carries the right tag is staged into
the LDS ring of row 0 (under back-pressure from wave 0) and published; an incomplete group is requested again.
Round 3 adds two things the model follows: (i) WORKERS and a READY QUEUE -- a band runs only once a worker has taken it from
the queue, band 0 of a lattice is there from the start and band b + 1 is published by band b when b's wave 0 is TRIG steps
into its sweep (or at its end), so with fewer workers than bands nobody may wait on a band that cannot start; (ii) the
THREE-predecessor mode (nq = 3): a row also reads what the row above wrote TWO steps ago, so a wave leads the next by one
step less, the fetcher starts one group earlier and waits one step longer before it rewrites an entry.  The geometry (lattices, row intervals, sweep ranges) comes from the kernel's own header
through a tiny C wrapper.  The model fails the expected way when the back-pressure wait is removed or when the last
row is stored outside its image interval (the hazard the tag protocol's overwrite argument excludes)."""
import ctypes
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = 4                            # points per fetcher load (G <= 16)


def ring_len(G):
    return 16 if G <= 8 else 8        # mgm_ring() of the kernel (K = 4); a wave may lead the next by ring - 2 steps


def waves(G):
    return 15 if G >= 64 else 8 if G >= 16 else 4        # mgm_waves() of the kernel (K = 4)


@pytest.fixture(scope="module")
def geom(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    so = str(tmp_path_factory.mktemp("geom") / "mgm_geom_capi.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "probes", "mgm_geom_capi.cpp")], check=True)
    return ctypes.CDLL(so)


def tag_of(band):
    return 1 + (band >> 1)                                     # a 16-bit count spread over the two free bytes of a dword: unique per slot within a launch (<= 4095 bands)


class Band:
    def __init__(self, b, rows, prev_last, U, NP, NW):
        self.b, self.U, self.NP, self.R = b, U, NP, NW * NP
        self.rows, self.prev_last = rows, prev_last        # [(lo, span)] of its rows; interval of the previous band's last row
        starts = [lo + j for j, (lo, sp) in enumerate(rows) if sp > 0]
        ends = [lo + sp + j for j, (lo, sp) in enumerate(rows) if sp > 0]
        self.s0, self.s1 = (min(starts), max(ends)) if starts else (0, 1)
        self.s0 &= ~(max(8, ring_len(64 // NP)) - 1)
        self.T = [self.s0] * NW                            # next step of each wave (= its progress word)
        self.chan = {}                                     # (row, entry) -> (writer row, step)
        self.fu = self.s0                                  # fetcher: points < fu are staged (its progress word)
        self.grp = self.s0 // FP                           # fetcher: group being served; snapshots of grp and grp + 1 are in flight
        self.snap = {}                                     # group -> snapshot {u: (tag, band, u)} (may be stale)

    def done(self):
        return all(t >= self.s1 for t in self.T)


TRIG = 16                         # S2P_MGM_TRIG of the kernel


def run_lattice(geom, q, w, h, G, seed, backpressure=True, store_outside=False, nq=2, workers=None, lead=None, trig=TRIG):
    out = (ctypes.c_int * 9)()
    geom.mgm_capi_lattice(q, w, h, out)
    U, V = out[1], out[2]
    if U <= 0 or V <= 0:
        return 0
    NP, NW = 64 // G, waves(G)
    RING = ring_len(G)
    LEAD, RM = (RING - nq if lead is None else lead), RING - 1
    R = NW * NP

    def interval(v):
        lo, sp = ctypes.c_int(), ctypes.c_int()
        geom.mgm_capi_row_interval(q, w, h, v, ctypes.byref(lo), ctypes.byref(sp))
        return lo.value, sp.value

    nb = (V + R - 1) // R
    bands = [Band(b, [interval(b * R + j) for j in range(R)], interval(b * R - 1), U, NP, NW) for b in range(nb)]
    ring = [dict(), dict()]                                # slot -> {u: (tag, band, u)}; "memset": empty
    rng = random.Random(seed)
    checked = idle = 0

    def snapshot(bd, g):
        slot = ring[(bd.b - 1) & 1]
        return {u: slot.get(u) for u in range(g * FP, (g + 1) * FP)}

    def ulim(bd):
        return min(U, bd.s1)

    def fetcher_event(bd):
        """One scheduling quantum of the fetcher of band bd: serve the oldest group as far as it goes."""
        if bd.b == 0 or bd.grp * FP >= ulim(bd):
            return False
        g = bd.grp
        if g not in bd.snap:
            bd.snap[g] = snapshot(bd, g)
            return False
        snap = bd.snap[g]
        plo, psp = bd.prev_last
        moved = False
        while bd.fu < (g + 1) * FP:
            u = bd.fu
            need = u < U and plo <= u < plo + psp
            if need and not (snap[u] is not None and snap[u][0] == tag_of(bd.b - 1)):
                break                                      # the prefix ends here
            if backpressure and bd.T[0] < u - (RING - 1) + (nq - 2):  # entry (u - 1) & 7 was last read in step u - 8 (u - 7 with three predecessors)
                return moved
            bd.chan[(0, (u - 1) & RM)] = ("in", snap[u][1], u) if need else ("in", None, u)
            bd.fu += 1
            moved = True
        if bd.fu >= (g + 1) * FP:                           # group done: its registers take the group after next
            del bd.snap[g]
            bd.grp += 1
            if (g + 2) * FP < ulim(bd):
                bd.snap[g + 2] = snapshot(bd, g + 2)
            if bd.grp not in bd.snap and bd.grp * FP < ulim(bd):
                bd.snap[bd.grp] = snapshot(bd, bd.grp)
        else:
            bd.snap[g] = snapshot(bd, g)                   # ask again
        return moved

    if nq == 3:                                            # the first step also reads the point before s0 of the previous band's row
        for bd in bands:
            if bd.s0 >= FP:
                bd.fu -= FP
                bd.grp -= 1
    # workers and the ready queue: band 0 is there from the start, band b + 1 is published by band b
    nworkers = nb if workers is None else workers
    queue, started, free, published = [0], [False] * nb, nworkers, [False] * nb
    published[0] = True
    if trig < 0:                                           # (model knob: every band in the queue from the start, the round-2 ticket order)
        for b in range(1, nb):
            published[b] = True
            queue.append(b)

    def publish_next(bd):
        if bd.b + 1 < nb and not published[bd.b + 1]:
            published[bd.b + 1] = True
            queue.append(bd.b + 1)

    for bd in bands:                                       # the requests in front of the sweep
        if bd.b > 0:
            for k in range(2):
                if (bd.grp + k) * FP < ulim(bd):
                    bd.snap[bd.grp + k] = snapshot(bd, bd.grp + k)

    finished = [False] * nb
    while not all(b.done() for b in bands):
        progressed = False
        while queue and free > 0 and rng.random() < 0.8:   # a free worker takes the next published band
            started[queue.pop(0)] = True
            free -= 1
            progressed = True
        order = [(i, wv) for i in range(nb) for wv in range(NW + 1)]
        rng.shuffle(order)
        for i, wv in order:
            bd = bands[i]
            if not started[i]:
                continue
            if bd.done():
                if not finished[i]:                        # the worker is free again (the band published its successor at the latest now)
                    publish_next(bd)
                    finished[i] = True
                    free += 1
                    progressed = True
                continue
            if wv == NW:
                if rng.random() < 0.7 and fetcher_event(bd):
                    progressed = True
                continue
            T = bd.T[wv]
            if T >= bd.s1 or rng.random() < 0.3:
                continue
            if wv > 0 and bd.T[wv - 1] < T:                # the wave above has not written step T - 1
                continue
            if backpressure and wv < NW - 1 and bd.T[wv + 1] < T - LEAD:
                continue
            if wv == 0 and bd.b > 0 and T < U and bd.fu < T + 1:     # the point this step reads is not staged yet
                continue
            for j in range(wv * NP, (wv + 1) * NP):        # the rows of the wave, in lock step
                u = T - j
                lo, sp = bd.rows[j]
                inside = lo <= u < lo + sp
                got = bd.chan.get((j, (T - 1) & RM))
                if j > 0:
                    if T > bd.s0:
                        assert got == (j - 1, T - 1), "band %d row %d step %d read %r" % (bd.b, j, T, got)
                    if nq == 3 and T > bd.s0 + 1:
                        got2 = bd.chan.get((j, (T - 2) & RM))
                        assert got2 == (j - 1, T - 2), "band %d row %d step %d read %r two steps back" % (bd.b, j, T, got2)
                elif bd.b > 0 and inside and bd.prev_last[0] <= u < sum(bd.prev_last):
                    assert got == ("in", bd.b - 1, u), "band %d step %d consumed %r" % (bd.b, T, got)
                    checked += 1
                if nq == 3 and j == 0 and bd.b > 0 and inside and bd.prev_last[0] <= u - 1 < sum(bd.prev_last):
                    got2 = bd.chan.get((0, (T - 2) & RM))   # the third predecessor (u - 1, v - 1): the previous band's point u - 1
                    assert got2 == ("in", bd.b - 1, u - 1), "band %d step %d: third predecessor %r" % (bd.b, T, got2)
                    checked += 1
                bd.chan[(j + 1, T & RM)] = (j, T)
                if j == R - 1 and 0 <= u < U and (inside or store_outside):
                    ring[bd.b & 1][u] = (tag_of(bd.b), bd.b, u)
            bd.T[wv] = T + 1
            if wv == 0 and T + 1 >= bd.s0 + trig:          # wave 0 is TRIG steps in: the successor enters the queue
                publish_next(bd)
            progressed = True
        idle = 0 if progressed else idle + 1
        assert idle < 200, "deadlock: nobody can pass its wait"
    return checked


SHAPES = [(131, 257, 16), (257, 131, 16), (67, 129, 8), (128, 256, 16), (40, 300, 64), (300, 37, 2), (64, 64, 4)]


@pytest.mark.parametrize("h,w,G", SHAPES)
def test_protocol_holds_under_random_schedules(geom, h, w, G):
    total = 0
    for q in range(12):
        for seed in range(3):
            total += run_lattice(geom, q, w, h, G, seed)
    if max(h, w) > 256 // G:
        assert total > 0                                   # multi-band lattices exist: hand-offs were actually checked


@pytest.mark.parametrize("h,w,G", [(131, 257, 16), (257, 131, 16), (67, 129, 8), (300, 37, 2)])
def test_three_predecessor_protocol_and_few_workers(geom, h, w, G):
    """nq = 3 (the entry of two steps ago is read as well) with as few as ONE worker per lattice: every read finds what it
    must, the third predecessor of a band's first row comes from the previous band's point u - 1, and nobody waits forever."""
    total = 0
    for q in range(12):
        for seed, workers in ((0, 1), (1, 2), (2, None)):
            total += run_lattice(geom, q, w, h, G, seed, nq=3, workers=workers)
            run_lattice(geom, q, w, h, G, seed + 7, nq=2, workers=workers if workers else 3)
    assert total > 0


def test_three_predecessors_need_the_shorter_lead(geom):
    """With the two-predecessor lead (ring - 2) a wave could rewrite the entry the wave below still needs as its
    two-steps-ago message."""
    failures = 0
    for q in range(4):
        for seed in range(6):
            try:
                run_lattice(geom, q, 129, 67, 16, seed, nq=3, lead=6)
            except AssertionError:
                failures += 1
    assert failures > 0


def test_model_needs_the_backpressure_wait(geom):
    """Without the wait on the wave below, some schedule rewrites a ring entry before it was read."""
    failures = 0
    for q in range(4):
        for seed in range(4):
            try:
                run_lattice(geom, q, 129, 67, 16, seed, backpressure=False)
            except AssertionError:
                failures += 1
    assert failures > 0


def test_model_needs_the_in_image_store_rule(geom):
    """If a band stored its last row outside the row's image interval as well, band b + 2 could overwrite a granule band
    b + 1 has not consumed (no data dependency orders the two there): the diamond lattices expose it as a wrong
    granule or as a chunk that never completes."""
    failures = 0
    for q in range(4, 12):
        for seed in range(6):
            try:
                run_lattice(geom, q, 257, 131, 16, seed, store_outside=True, trig=-10 ** 6)   # (every band published at once: the rule must not lean on the queue's timing)
            except AssertionError:
                failures += 1
    assert failures > 0
