"""Generates tests/golden/example_golden.npz from the reference's example data -- run in the build
container only (needs /root/reference/assets/example/example.zip; the GPU box has no /root/reference).

What is stored is DATA: the training sessions as parsed by the oracle's restatement of read_from_file
(src/vmisknn/vmis_index.rs:591-686), the 931 evaluator queries of test.txt (src/bin/evaluator.rs:46-56)
for two parameter sets, and the expected ranked ids/scores from the canonical oracle, which
tests/test_oracle_pins.py pins against the reference's own known answers first.

  a: BASELINE.json config 1  -- m=1502 k=288 last_items=4 how_many=21 idf_weighting=1
  b: shipped example.toml    -- m=500  k=50  last_items=2 how_many=21 idf_weighting=1
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from helpers import evaluator_queries, extract_example, flatten, read_test_data_evolving  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    d = extract_example(tmp)
    off, items, ts, _ = O.read_tsv(os.path.join(d, "train.txt"))
    test = read_test_data_evolving(os.path.join(d, "test.txt"))
    lens = np.diff(off.astype(np.int64))
    max_len = int(round(float(np.quantile(lens, 0.995))))
    out = dict(sess_off=off.astype(np.uint32), items=items.astype(np.uint32), ts=ts)
    # raw rows of test.txt (session, item, rounded time) so that the evaluator binary can be run on the GPU box
    rows = [l.split() for l in open(os.path.join(d, "test.txt")).read().splitlines()[1:]]
    out["test_rows"] = np.array([(int(a), int(b), round(float(c))) for a, b, c in rows], np.int64)
    for tag, (m, k, last, n, idfw) in dict(a=(1502, 288, 4, 21, 1.0), b=(500, 50, 2, 21, 1.0)).items():
        qs = evaluator_queries(test, last)
        flat, qoff = flatten([q for q, _ in qs])
        ix = O.OracleIndex(off, items, ts, m, max_len, idfw)
        r = ix.predict_batch("canonical", flat, qoff, k, m, n, business=True, threads=4)
        out["params_" + tag] = np.array([m, k, n, idfw, max_len], np.float64)
        out["q_items_" + tag] = flat.astype(np.uint32)
        out["q_off_" + tag] = qoff
        out["next_" + tag] = np.array([nx[0] for _, nx in qs], np.uint32)
        out["ids_" + tag] = r["ids"].astype(np.uint32)
        out["scores_" + tag] = r["scores"]
        out["counts_" + tag] = r["counts"]
    path = os.path.join(HERE, "example_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(ts), "sessions,", len(qoff) - 1, "queries")


if __name__ == "__main__":
    main()
