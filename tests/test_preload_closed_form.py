"""The property k_l2_locate's pre-load rests on (mm_l2.hip, "the state after the pre-load, in closed form"): with inserts only,
SlideMapper's state (slidingMap.hpp:103-160) does not depend on the order of the inserts -- the cell counts add up, the pivot is the
largest cell p whose cells 1..p hold at most S hashes, pivRank is that number, sharedSketchElements / strand_votes are sums over the
active cells up to the pivot.  Checked here on the CPU with a literal restatement of insert_minmer against the closed form, on random
pre-loads (the records still open at rangeStart, computeMap.hpp:1323-1338); the device's cells themselves are compared with the oracle
by the -m gpu suite."""
import random


def literal_preload(q, qstrand, inserts):
    """insert_minmer (slidingMap.hpp:125-160) applied one by one; cells 1..S, cell 0 a dummy as in the reference's vector"""
    S = len(q)
    cnt = [0] + [1] * S
    act = [0] * (S + 1)
    vote = [0] * (S + 1)
    hv = [0] + list(q)
    pivot, piv_rank, shared, votes = S, S, 0, 0
    for h, strand in inserts:
        lo, hi = 1, S + 1                                 # lower_bound over cells 1..S
        while lo < hi:
            mid = (lo + hi) // 2
            if hv[mid] < h:
                lo = mid + 1
            else:
                hi = mid
        j = lo
        if j == S + 1:
            continue
        if hv[j] == h:
            act[j] = 1
            vote[j] += qstrand[j - 1] * strand
            if hv[j] <= hv[pivot]:
                shared += 1
                votes += vote[j]
        else:
            cnt[j] += 1
            if hv[j] <= hv[pivot]:
                piv_rank += 1
            if piv_rank > S:
                shared -= act[pivot]
                votes -= vote[pivot]
                piv_rank -= cnt[pivot]
                pivot -= 1
    return cnt, act, vote, pivot, piv_rank, shared, votes


def closed_form(q, qstrand, inserts):
    """what k_l2_locate builds: cells from the multiset of inserts, the four scalars from the cells"""
    S = len(q)
    cnt = [0] + [1] * S
    act = [0] * (S + 1)
    vote = [0] * (S + 1)
    pos = {h: i + 1 for i, h in enumerate(q)}
    import bisect
    for h, strand in inserts:
        if h in pos:
            j = pos[h]
            act[j] = 1
            vote[j] += qstrand[j - 1] * strand
        else:
            j = bisect.bisect_left(q, h) + 1
            if j <= S:
                cnt[j] += 1
    pivot, piv_rank, shared, votes, run = 0, 0, 0, 0, 0
    for p in range(1, S + 1):
        run += cnt[p]
        if run > S:
            break
        pivot, piv_rank = p, run
        shared += act[p]
        votes += vote[p] if act[p] else 0
    return cnt, act, vote, pivot, piv_rank, shared, votes


def one_case(rng, S, n_ref, p_match, spread):
    q = sorted(rng.sample(range(1, spread), S))
    qstrand = [rng.choice((-1, 1)) for _ in q]
    matched = set()
    inserts = []
    for _ in range(n_ref):
        if rng.random() < p_match:
            h = rng.choice(q)
            if h in matched:                              # a query hash open twice goes to k_l2_sweep_exact, not through the closed form
                continue
            matched.add(h)
        else:
            h = rng.randrange(1, int(spread * 1.2))       # some beyond the largest query hash: no effect (:136-139)
            if h in set(q):
                continue
        inserts.append((h, rng.choice((-1, 1))))
    return q, qstrand, inserts


def test_preload_state_is_order_free_and_equals_the_closed_form():
    rng = random.Random(20260927)
    for trial in range(400):
        S = rng.choice((1, 2, 3, 7, 20, 64, 130))
        n_ref = rng.choice((0, 1, S // 2, S, 2 * S, 5 * S))
        q, qstrand, inserts = one_case(rng, S, n_ref, rng.choice((0.0, 0.2, 0.6)), rng.choice((4 * S + 8, 1000 * S)))
        want = closed_form(q, qstrand, inserts)
        for _ in range(3):
            rng.shuffle(inserts)
            got = literal_preload(q, qstrand, inserts)
            assert got == want, (trial, S, n_ref)


def test_pivot_can_reach_the_dummy_cell():
    """a pre-load with S reference-only hashes below the smallest query hash pushes the pivot to cell 0"""
    q = [100, 200, 300]
    inserts = [(1, 1), (2, 1), (3, -1), (150, 1)]
    got = literal_preload(q, [1, -1, 1], inserts)
    assert got == closed_form(q, [1, -1, 1], inserts)
    assert got[3] == 0 and got[4] == 0 and got[5] == 0
