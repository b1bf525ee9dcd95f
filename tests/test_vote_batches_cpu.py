"""CPU: the argument behind search_wave_kernel (ngmlr_amd/csrc/cvx_search.hip) checked as an algorithm, without a device.

The kernel casts 64 consecutive votes of a read at once and claims to reproduce CS::AddLocationStd's sequential semantics
(reference src/CS.cpp:101-149: open addressing with linear probing and a probe budget, scores of +1.0f, a threshold that grows
with the best score, rList in order of qualification) by construction.  Here both are written down in plain Python -- the
reference's vote-by-vote loop, and the batch procedure with exactly the kernel's rules (probe the table as it stood before the
batch; same free slot for different bins = hazard -> vote by vote; a batch that could exhaust the budget -> vote by vote; ranks
per entry and orientation in lane order; prefix maximum for the threshold of the moment; first qualifying vote of an entry lists
it) -- and compared on random vote streams over tables small enough that collisions, hazards, duplicates and overflows are the
normal case.  The kernels themselves are compared with the reference's recorded calls and with oracle/cs_oracle.c on the GPU
(tests/test_gpu_search.py)."""
import numpy as np

MULT = 11400714819323199488
MASK64 = (1 << 64) - 1


def _hash(b, bits):
    return ((b * MULT) & MASK64) >> (64 - bits)


class Table:
    def __init__(self, bits, hpoc):
        self.bits, self.size = bits, 1 << bits
        self.key = [None] * self.size
        self.score = [[0.0, 0.0] for _ in range(self.size)]
        self.listed = [False] * self.size
        self.rlist = []
        self.max_hit = np.float32(0.0)
        self.hpoc = hpoc
        self.overflow_at = None

    def state(self):
        return (self.key, self.score, self.listed, self.rlist, float(self.max_hit), self.hpoc, self.overflow_at)


def vote_serial(t, b, rev, sens, index):
    """one vote as the reference casts it; False when the budget ran out at this vote"""
    e = _hash(b, t.bits)
    while t.key[e] is not None and t.key[e] != b:
        e = (e + 1) % t.size
        t.hpoc -= 1
        if t.hpoc == 0:
            t.overflow_at = index
            return False
    if t.key[e] is None:
        t.key[e] = b
        t.score[e] = [0.0, 0.0]
    t.score[e][rev] += 1.0
    s = np.float32(t.score[e][rev])
    if s > t.max_hit:
        t.max_hit = s
    if not t.listed[e] and s >= np.float32(t.max_hit * np.float32(sens)):
        t.listed[e] = True
        t.rlist.append(e)
    return True


def run_serial(votes, bits, hpoc, sens):
    t = Table(bits, hpoc)
    for i, (b, rev) in enumerate(votes):
        if not vote_serial(t, b, rev, sens, i):
            break
    return t


def run_batched(votes, bits, hpoc, sens, width=64):
    t = Table(bits, hpoc)
    stats = {"parallel": 0, "hazard": 0, "budget": 0}
    for v0 in range(0, len(votes), width):
        batch = votes[v0:v0 + width]
        # probe the table as it stands
        final, steps, free = [], [], []
        for b, rev in batch:
            e, n = _hash(b, bits), 0
            while t.key[e] is not None and t.key[e] != b:
                e = (e + 1) % t.size
                n += 1
                if n >= t.hpoc:
                    break
            final.append(e)
            steps.append(n)
            free.append(t.key[e] is None)
        serial = False
        if sum(steps) >= t.hpoc:
            serial = True
            stats["budget"] += 1
        else:
            claimed = {}
            for j, (b, rev) in enumerate(batch):
                if free[j]:
                    if claimed.setdefault(final[j], b) != b:
                        serial = True
            if serial:
                stats["hazard"] += 1
        if serial:
            for j, (b, rev) in enumerate(batch):
                if not vote_serial(t, b, rev, sens, v0 + j):
                    return t, stats
            continue
        stats["parallel"] += 1
        t.hpoc -= sum(steps)
        listed_before = [t.listed[e] for e in final]
        base = []
        for j, (b, rev) in enumerate(batch):
            if free[j] and t.key[final[j]] is None:
                t.key[final[j]] = b
                t.score[final[j]] = [0.0, 0.0]
        for j, (b, rev) in enumerate(batch):
            base.append(t.score[final[j]][rev])
        s = []
        for j, (b, rev) in enumerate(batch):
            rank = sum(1 for i in range(j) if final[i] == final[j] and batch[i][1] == rev)
            s.append(np.float32(base[j] + rank + 1))
        pm, run = [], np.float32(0.0)
        for x in s:
            run = x if x > run else run
            pm.append(run)
        qual = [s[j] >= np.float32((pm[j] if pm[j] > t.max_hit else t.max_hit) * np.float32(sens)) for j in range(len(batch))]
        for j in range(len(batch)):
            if qual[j] and not listed_before[j] and not any(qual[i] and final[i] == final[j] for i in range(j)):
                t.listed[final[j]] = True
                t.rlist.append(final[j])
        for j, (b, rev) in enumerate(batch):
            if not any(final[i] == final[j] and batch[i][1] == rev for i in range(j + 1, len(batch))):
                t.score[final[j]][rev] = float(s[j])
        if pm[-1] > t.max_hit:
            t.max_hit = pm[-1]
    return t, stats


def _stream(rng, n, n_bins, hot):
    """votes over n_bins bins, a share `hot` of them into a handful of bins (repeats: the same entry many times per batch)"""
    bins = rng.integers(0, 1 << 40, size=n_bins)
    favourites = bins[:max(1, n_bins // 16)]
    out = []
    for _ in range(n):
        b = int(favourites[rng.integers(0, len(favourites))]) if rng.random() < hot else int(bins[rng.integers(0, n_bins)])
        out.append((b, int(rng.integers(0, 2))))
    return out


def test_batches_of_64_votes_equal_the_sequential_vote():
    rng = np.random.default_rng(5)
    seen = {"parallel": 0, "hazard": 0, "budget": 0}
    overflows = 0
    for case in range(300):
        bits = int(rng.integers(4, 11))
        n_bins = int(rng.integers(2, (1 << bits) * 3 // 4 + 2))
        votes = _stream(rng, int(rng.integers(1, 700)), n_bins, float(rng.choice([0.0, 0.3, 0.9])))
        hpoc = int((1 << bits) * float(rng.choice([0.333, 0.777, 8.0])))
        sens = float(rng.choice([0.5, 0.8, 0.9, 1.0]))
        want = run_serial(votes, bits, hpoc, sens)
        got, stats = run_batched(votes, bits, hpoc, sens)
        assert got.state() == want.state(), (case, bits, n_bins, len(votes), hpoc, sens)
        for k in seen:
            seen[k] += stats[k]
        overflows += want.overflow_at is not None
    # every path of the procedure ran, many times
    assert seen["parallel"] > 300 and seen["hazard"] > 100 and seen["budget"] > 30 and overflows > 30, (seen, overflows)


def test_other_batch_widths_are_the_same_procedure():
    rng = np.random.default_rng(6)
    for case in range(60):
        bits = int(rng.integers(5, 9))
        votes = _stream(rng, 400, 1 << (bits - 1), 0.5)
        want = run_serial(votes, bits, 4 << bits, 0.9)
        for width in (1, 7, 64, 400):
            got, _ = run_batched(votes, bits, 4 << bits, 0.9, width=width)
            assert got.state() == want.state(), (case, width)
