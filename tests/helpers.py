"""Shared test helpers: the reference's evaluator workload and metrics restated for checking, tie-aware
comparisons, and small synthetic datasets.  Test-side only."""
import os
import zipfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_ZIP = "/root/reference/assets/example/example.zip"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def have_reference_assets():
    return os.path.exists(REFERENCE_ZIP)


def extract_example(tmpdir):
    with zipfile.ZipFile(REFERENCE_ZIP) as z:
        z.extractall(tmpdir)
    return str(tmpdir)


def read_test_data_evolving(path):
    """src/io.rs:40-59: group rows by session, order each session's events by (rounded) time."""
    sessions = {}
    with open(path) as f:
        next(f)
        for line in f:
            parts = line.split()
            if len(parts) < 3:
                continue
            sessions.setdefault(int(parts[0]), []).append((int(parts[1]), round(float(parts[2]))))
    return {s: [i for i, _ in sorted(v, key=lambda x: x[1])] for s, v in sessions.items()}


def evaluator_queries(test_sessions, max_items_in_session):
    """src/bin/evaluator.rs:46-56: every prefix 1..len-1, truncated to the last max_items items.
    -> list of (evolving_session, next_items)."""
    out = []
    for _sid, ev in test_sessions.items():
        for state in range(1, len(ev)):
            start = state - max_items_in_session if state > max_items_in_session else 0
            out.append((ev[start:state], ev[state:]))
    return out


def mrr_hitrate(recommendations, next_items, length=20):
    """src/metrics/mrr.rs:24-33 and src/metrics/hitrate.rs:24-33 (first next item only)."""
    mrr = hit = 0.0
    for recs, nxt in zip(recommendations, next_items):
        top = list(recs[:length])
        if nxt[0] in top:
            hit += 1.0
            mrr += 1.0 / (top.index(nxt[0]) + 1)
    n = max(1, len(recommendations))
    return mrr / n, hit / n


def flatten(sessions):
    off = np.zeros(len(sessions) + 1, np.uint32)
    off[1:] = np.cumsum([len(s) for s in sessions])
    flat = np.array([x for s in sessions for x in s], dtype=np.uint64)
    return flat, off


def assert_valid_topn(ids, scores, all_ids, all_scores, how_many, exclude=(), rtol=1e-9):
    """`ids/scores` must be A valid top-`how_many` of the full score table (ties may fall either way):
    scores match the table, are non-increasing, and nothing left out beats the last one returned."""
    table = dict(zip((int(i) for i in all_ids), (float(s) for s in all_scores)))
    for x in exclude:
        table.pop(int(x), None)
    assert len(ids) == min(how_many, len(table))
    assert len(set(int(i) for i in ids)) == len(ids)
    for i, s in zip(ids, scores):
        assert int(i) in table, "returned an item that was never scored"
        assert abs(table[int(i)] - s) <= rtol * max(1.0, abs(s)), (i, table[int(i)], s)
    for a, b in zip(scores[:-1], scores[1:]):
        assert a >= b - rtol * max(1.0, abs(a))
    if len(ids) and len(ids) < len(table):
        rest = max(v for k, v in table.items() if k not in set(int(i) for i in ids))
        assert rest <= scores[-1] + rtol * max(1.0, abs(rest))


def small_dataset(seed, n_sessions=1500, n_items=300, tied_timestamps=False, max_len=12):
    """Random training sessions (ascending de-duplicated rows) small enough for the literal oracle."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_items + 1) ** 0.9
    w /= w.sum()
    ids = rng.permutation(np.arange(1, 50 * n_items, 37, dtype=np.uint64))[:n_items] + np.uint64(10**12)
    off, items = [0], []
    for _ in range(n_sessions):
        ln = int(rng.integers(1, max_len + 1))
        row = np.unique(ids[rng.choice(n_items, size=ln, p=w)])
        items.extend(row.tolist())
        off.append(len(items))
    if tied_timestamps:
        ts = rng.integers(1000, 1000 + n_sessions // 8, size=n_sessions).astype(np.uint32)
    else:
        ts = (1000 + rng.permutation(n_sessions)).astype(np.uint32)
    return np.array(off, np.uint64), np.array(items, np.uint64), ts, ids


def random_queries(seed, ids, n, max_len=6, unknown_rate=0.05, dup_rate=0.2):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(ids) + 1) ** 0.9
    w /= w.sum()
    qs = []
    for _ in range(n):
        ln = int(rng.integers(1, max_len + 1))
        q = ids[rng.choice(len(ids), size=ln, p=w)].tolist()
        for j in range(ln):
            if rng.random() < unknown_rate:
                q[j] = int(7 + rng.integers(0, 1000))          # id the index has never seen
            elif j and rng.random() < dup_rate:
                q[j] = q[int(rng.integers(0, j))]              # repeated item
        qs.append([int(x) for x in q])
    return qs


def big_config_unavailable(config, why):
    """A BASELINE config that cannot run at full size on this box must not vanish quietly: the test FAILS (the driver then reports the config as
    untested) unless SRN_ALLOW_SKIP_BIG=1 says the smaller box is intended."""
    import pytest
    msg = "BASELINE %s cannot run at full size here: %s" % (config, why)
    if os.environ.get("SRN_ALLOW_SKIP_BIG") == "1":
        pytest.skip(msg + " (SRN_ALLOW_SKIP_BIG=1)")
    pytest.fail(msg + " -- set SRN_ALLOW_SKIP_BIG=1 to skip it on purpose")


def eight_metrics(recommendations, next_items, training_items, length=20):
    """The reference's evaluation report (src/metrics/evaluation_reporter.rs:12-117) restated: Mrr, Ndcg, HitRate, Popularity, Precision, Coverage, Recall,
    F1score @length over (recommendations, next_items) pairs; training_items = the item column of the training data (popularity.rs:20-29, coverage.rs:17-25).
    Same formulas as serenade_amd/csrc/host/evaluator.cpp (which is pinned on the reference's metric KATs); test-side only."""
    import math
    from collections import Counter
    freq = Counter(int(x) for x in training_items)
    max_freq = max(freq.values()) if freq else 1
    n = mrr = ndcg = hit = pop = prec = rec = 0.0
    covered = set()

    def dcg(top, nxt):                                                    # ndcg.rs:13-27
        return sum((1.0 if i == 0 else 1.0 / math.log2(i + 1.0)) for i, x in enumerate(top) if x in nxt)

    for recs, nxt in zip(recommendations, next_items):
        n += 1
        top = [int(x) for x in recs[:length]]
        nxt = [int(x) for x in nxt]
        if nxt[0] in top:                                                 # mrr.rs:24-33, hitrate.rs:24-33
            mrr += 1.0 / (top.index(nxt[0]) + 1)
            hit += 1.0
        nset, tset = set(nxt), set(top)
        ndcg += dcg(top, nset) / dcg(nxt[:length], nset)                  # ndcg.rs:42-56
        inter = len(nset & tset)
        prec += inter / float(length)                                     # precision.rs:31-43
        rec += inter / float(len(nxt))                                    # recall.rs:31-44
        if tset:                                                          # popularity.rs:41-58
            pop += sum(freq[x] / float(max_freq) for x in tset if x in freq) / len(tset)
        covered.update(top)                                               # coverage.rs:29-38
    n = max(n, 1.0)
    p, r = prec / n, rec / n
    f1 = 2.0 * p * r / (p + r) if p + r > 0 else 0.0                      # f1score.rs:27-36
    return {"Mrr": mrr / n, "Ndcg": ndcg / n, "HitRate": hit / n, "Popularity": pop / n, "Precision": p, "Coverage": len(covered) / float(max(1, len(freq))),
            "Recall": r, "F1score": f1}
