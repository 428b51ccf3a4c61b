"""What "item-exact against the canonical refinement" means where the benchmark runs.

The reference's k-cut iterates a hash map (src/vmisknn/vmis_index.rs:393-414), so wherever more than k sessions survive the m-cut -- 85 % of BASELINE configs[1]'s
queries -- which equal-similarity sessions close the neighbourhood is decided by hashbrown's per-process order: two runs of the Rust binary disagree with each other
there.  The product computes ONE fixed instance (ties by recency).  On configs[1] at full size the recommendation SETS of the product and of the literal restatement
(another fixed instance: its own hash order) differ on ~77 % of the queries; this test pins that number's order of magnitude AND that the evaluator's quality metrics
(Mrr@20 src/metrics/mrr.rs:24-33, HitRate@20 src/metrics/hitrate.rs:24-33, against each query's held-out next item, evaluator.rs:75) do not move: |delta| <= 0.01."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_literal_vs_canonical_quality_on_baseline_config_2():
    import serenade_amd as sa
    from serenade_amd import synth
    from oracle import oracle as O
    from oracle import refinement as R
    inter, n_items, k, m, idfw = synth.CONFIGS["cfg2"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
    qi, qo, nxt = synth.queries(2000, n_items, seed=synth.SEED + 7919, with_next=True)     # (the seed of bench.py's rank-0 query stream)
    n = 3000
    flat, q_off = qi[:qo[n]], qo[:n + 1]
    hip = sa.predict_batch(gix, (flat, q_off), k, m, synth.HOW_MANY, False)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    r = R.literal_vs_canonical(oix, flat, q_off, nxt[:n], k, m, synth.HOW_MANY, threads=8, hip=hip)
    print("\n[literal-vs-canonical] cfg2, k=%d m=%d, %d queries: %s" % (k, m, n, {a: b for a, b in r.items() if a != "note"}))
    assert r["hip_equals_canonical"]                                        # the product IS the canonical form (ids and order; scores are checked elsewhere)
    assert r["k_cut_bites"] > 0.5 and r["top_n_set_differs"] > 0.5         # the refinement is not a corner case at this config: most queries are decided by a tie rule
    assert r["hip"] == r["canonical"]
    assert 0.05 < r["literal"]["mrr_at_20"] < 0.5 and 0.1 < r["literal"]["hitrate_at_20"] < 0.9   # (the metric has signal on the synthetic stream)
    assert abs(r["delta_mrr_canonical_minus_literal"]) <= 0.01 and abs(r["delta_hitrate_canonical_minus_literal"]) <= 0.01


def test_mrr_hitrate_formulas_on_the_references_own_examples():
    """src/metrics/mrr.rs:50-63 (should_calculate_mrr) and the same list for HitRate (hitrate.rs tests): next item 3 at rank 3 of 24 recommendations."""
    from oracle import refinement as R
    ids = np.arange(1, 25, dtype=np.uint64).reshape(1, 24)
    mrr, hit = R.mrr_hitrate(ids, np.array([24], np.uint32), np.array([3], np.uint64))
    assert mrr == 0.3333333333333333 and hit == 1.0
    mrr, hit = R.mrr_hitrate(ids, np.array([24], np.uint32), np.array([21], np.uint64))      # beyond the first 20: no credit
    assert mrr == 0.0 and hit == 0.0
    mrr, hit = R.mrr_hitrate(np.array([[1, 2]], np.uint64), np.array([2], np.uint32), np.array([2], np.uint64))   # hitrate.rs should_happyflow_hitrate: [1, 2] vs next 2
    assert hit == 1.0 and mrr == 0.5
    mrr, hit = R.mrr_hitrate(ids, np.array([2], np.uint32), np.array([3], np.uint64))        # a short list: entries beyond its count are not recommendations
    assert mrr == 0.0 and hit == 0.0
