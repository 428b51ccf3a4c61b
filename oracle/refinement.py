"""What the CANONICAL refinement costs against the reference's LITERAL loops on a query sample  --  TEST INFRASTRUCTURE ONLY (see vmis_oracle.cpp's header).

The reference's k-cut iterates a hash map (src/vmisknn/vmis_index.rs:393-414): where more than k candidate sessions survive the m-cut, WHICH of the equal-similarity
sessions close the neighbourhood depends on hashbrown's per-process order, so two runs of the Rust binary differ from each other.  The product (and the canonical
oracle) implement one fixed instance: ties by recency.  This module measures, on the same queries, how far the literal restatement (its own fixed hash order: one
possible run of the reference) and the canonical form are apart -- in the recommendation lists and in the metrics the reference's evaluator reports
(Mrr@20 src/metrics/mrr.rs:24-33, HitRate@20 src/metrics/hitrate.rs:24-33; both look at next_items[0] only, src/bin/evaluator.rs:75)."""
import numpy as np


def mrr_hitrate(ids, counts, next_items, length=20):
    """ids u64[nq, how_many] (score-descending rows), counts[nq], next_items u64[nq] -> (Mrr@length, HitRate@length) over all nq (qty counts every call: mrr.rs:25)."""
    nq, hm = ids.shape
    top = min(length, hm)
    inside = np.arange(top)[None, :] < np.minimum(counts.astype(np.int64), top)[:, None]
    hit = (ids[:, :top] == next_items[:, None]) & inside
    any_hit = hit.any(axis=1)
    rank = hit.argmax(axis=1)                       # first position (ids are distinct inside a row)
    mrr = float(np.where(any_hit, 1.0 / (rank + 1.0), 0.0).sum() / max(1, nq))
    return mrr, float(any_hit.sum() / max(1, nq))


def literal_vs_canonical(oix, flat, q_off, next_items, k, m, how_many, threads=1, hip=None):
    """oix: oracle.OracleIndex; (flat, q_off): the query sample; next_items u64[nq] or None; hip = (ids, scores, counts) of the product on the same queries or None.
    -> dict of shares (of the nq queries) and metrics."""
    nq = len(q_off) - 1
    lit = oix.predict_batch("literal", flat, q_off, k, m, how_many, False, threads=threads)
    can = oix.predict_batch("canonical", flat, q_off, k, m, how_many, False, threads=threads, want_stats=True)
    col = np.arange(how_many)[None, :]
    lm, cm = col < lit["counts"][:, None].astype(np.int64), col < can["counts"][:, None].astype(np.int64)
    li, ci = np.where(lm, lit["ids"], 0), np.where(cm, can["ids"], 0)
    ranked_differs = (lit["counts"] != can["counts"]) | (li != ci).any(axis=1)
    set_differs = (lit["counts"] != can["counts"]) | (np.sort(li, axis=1) != np.sort(ci, axis=1)).any(axis=1)
    st = can["stats"]                                # P C K I D H L per query
    out = {"queries": int(nq),
           "top_n_set_differs": float(set_differs.mean()), "ranked_list_differs": float(ranked_differs.mean()),
           "k_cut_bites": float((st[:, 1] > st[:, 2]).mean()),          # more candidates than neighbours: the k-cut dropped some
           "m_cut_reached": float((st[:, 1] >= m).mean()),               # the candidate set is full: the m-cut (may have) dropped some
           "note": "literal = the reference's loops with the restatement's fixed hash order (one possible run of the Rust binary: its k-cut iterates a hash map, vmis_index.rs:393-414); "
                   "canonical = ties by recency (DESIGN.md section 1), what the product computes"}
    if next_items is not None:
        nxt = np.ascontiguousarray(next_items, np.uint64)
        out["literal"] = dict(zip(("mrr_at_20", "hitrate_at_20"), mrr_hitrate(lit["ids"], lit["counts"], nxt)))
        out["canonical"] = dict(zip(("mrr_at_20", "hitrate_at_20"), mrr_hitrate(can["ids"], can["counts"], nxt)))
        if hip is not None:
            out["hip"] = dict(zip(("mrr_at_20", "hitrate_at_20"), mrr_hitrate(np.asarray(hip[0]).view(np.uint64).reshape(nq, how_many), np.asarray(hip[2]).view(np.uint32), nxt)))
        out["delta_mrr_canonical_minus_literal"] = out["canonical"]["mrr_at_20"] - out["literal"]["mrr_at_20"]
        out["delta_hitrate_canonical_minus_literal"] = out["canonical"]["hitrate_at_20"] - out["literal"]["hitrate_at_20"]
    if hip is not None:
        hm_ = col < np.asarray(hip[2]).view(np.uint32)[:, None].astype(np.int64)
        hi = np.where(hm_, np.asarray(hip[0]).view(np.uint64).reshape(nq, how_many), 0)
        out["hip_equals_canonical"] = bool(np.array_equal(np.asarray(hip[2]).view(np.uint32), can["counts"]) and np.array_equal(hi, ci))
    return out
