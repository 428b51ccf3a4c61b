"""bench.py's N > 1 code path on a one-GPU box: `--rehearse N` starts N ranks under torch.distributed.run, all on device 0, process group over
gloo, the shard group's collectives through application callbacks -- every line of the multi-GPU run except RCCL itself: shard cut, group creation, the parity gate through
the sharded call, the non-overlapped run, the overlapped run under its watchdog, and the assembly of the ONE JSON line.  A KeyError there must not wait for an 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _run(args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line on stdout, whatever the libraries print
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_rehearsal_of_the_multi_gpu_line(world):
    line = _run(["--rehearse", str(world), "--steps", "2", "--warmup", "1", "--config", "tiny", "--batch", "8192", "--shard-batch", "4096", "--no-sweep", "--cpu-seconds", "1"])
    for key in CONTRACT:
        assert key in line, key
    assert line["n_gpus"] == world and line["value_mode"] == "item-sharded" and line["scaling"] == "strong"
    assert line["value"] == line["value_item_sharded"] > 0 and line["value_replicas"] > 0
    cfg = line["config"]
    assert cfg["transport"] == "callbacks" and cfg["rccl_ranks"] == 0 and "rehearsal" in cfg and cfg["rehearsal"]
    probe = cfg["overlap_probe"]
    assert probe["non_overlapped"]["value"] > 0 and probe["overlapped"]["value"] > 0 and probe["timed_run"] in ("overlapped", "non-overlapped", "streaming, non-overlapped")
    assert cfg["exchange_overlapped_with_previous_batch"] == (probe["timed_run"] == "overlapped")
    third = probe["streaming_non_overlapped"]           # round 6: the streaming form of the neighbours exchange as a third run (set_postings again: a collective) -- same rows
    assert third["value"] > 0 and third["rows_equal_the_gather_form_rows"] is True and (third["streaming_form_ran"] or probe["timed_run"] != "streaming, non-overlapped")
    assert line["parity_checked"] > 0 and line["queries_served_last_step"] == 4096
    rep = line["replicas"]
    assert rep["value"] == line["value_replicas"] and rep["parity_checked"] > 0 and rep["full_batch_properties_ok"] is True and rep["kernel"]["kernel"].startswith("vmis_")
    rf = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["peak"] == 8000.0 * world and 0 < rf["frac"] < 1
    ex = line["exchange_bytes_per_query_rank0"]
    assert ex["topn_all_gather"] > 0 and ex["neighbour_lists_all_gather"] > 0 and cfg["pipeline"] == "neighbours"


def test_single_gpu_line_carries_both_modes():
    line = _run(["--steps", "2", "--warmup", "1", "--config", "tiny", "--batch", "8192", "--shard-batch", "4096", "--no-sweep", "--cpu-seconds", "1"])
    for key in CONTRACT:
        assert key in line, key
    assert line["n_gpus"] == 1 and line["value_mode"] == "replicas" and line["scaling"] == "weak" and line["value"] == line["value_replicas"] > 0
    assert line["value_item_sharded"] == line["item_sharded"]["value"] > 0 and line["item_sharded"]["rccl_ranks"] == 1 and line["item_sharded"]["transport"] == "rccl"
    assert line["item_sharded"]["parity_checked"] > 0
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["kernel"].startswith("vmis_")
    # round 5: the HBM traffic of the dominant kernel is measured INSIDE the run by default (two rocprofv3 --pmc passes of this command); a box without a working profiler
    # falls back to the committed summary, labelled
    assert rf["traffic_measured_in_this_run"] in (True, False) and (rf["traffic_measured_in_this_run"] is False or rf["traffic"] > 0)
    g8 = line["item_sharded"]["local_g8"]            # one rank's work of an 8-way item-sharded index, all 8 shards on this GPU
    assert "error" not in g8, g8
    assert g8["n_shards"] == 8 and g8["rank0_ms_total"] > 0 and g8["parity_checked"] > 0 and set(g8["rank0_ms"]) and g8["projected_node_queries_per_s_without_exchanges"] > 0
    assert line["parity_checked_positions"].startswith("uniform over batch")
    assert cb["kind"] == "port" and cb["cores"] >= 1 and set(cb["single_thread_per_call_us"]) >= {"p25", "p50", "p75", "p90", "p95", "p99_5"}


def test_default_bench_line_on_the_tiny_config_carries_every_block():
    """`python bench.py` as the driver runs it (N = 1, sweeps and CPU baseline on), on the generator's tiny config so that it takes seconds: ONE line, the contract's keys, `roofline`,
    `cpu_baseline`, the batch sweep and -- since round 4 -- `latency.long_sessions` (sessions of up to 8 / 10 items through the fast kernel's MID instantiation, oracle-gated inside the run)."""
    line = _run(["--config", "tiny", "--batch", "16384", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    for key in CONTRACT + ("value_replicas", "value_item_sharded", "value_mode"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0 and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    ls = line["latency"]["long_sessions"]
    assert [e["max_items_in_session"] for e in ls] == [8, 10, 20]
    assert ls[2]["without_long_tier"]["reached_general_kernel"] > ls[2]["mid_tier"]["reached_general_kernel"]      # (round 5: the LONG instantiation takes the sessions of 11..20 items)
    for e in ls:
        assert e["parity_checked"] == 256 and e["mid_tier"]["queries_per_s"] > 0 and e["mid_tier"]["listed_for_mid_instantiation"] > 0
    assert ls[1]["without_mid_tier"]["listed_for_mid_instantiation"] == 0 and ls[1]["without_mid_tier"]["reached_general_kernel"] == ls[1]["batch"]
    assert len(line["latency"]["batch_sweep"]) >= 5 and line["latency"]["single_query_us_p50"] > 0


def test_bench_line_through_the_avro_route():
    """`bench.py --index avro:per-item --tie-per-second 8` (round 6): the same synthetic sessions through the reference's production route -- a stand-in producer writes the
    Avro item / session index with its own order among equal timestamps, srn_index_new_from_avro loads it, the parity gate checks against the oracle's restatement of
    VMISIndex::new -- on the tiny config, so that the path stays alive between rounds."""
    line = _run(["--config", "tiny", "--batch", "16384", "--steps", "2", "--warmup", "1", "--no-sweep", "--no-cpu-baseline", "--no-measure-traffic", "--mode", "replicas",
                 "--index", "avro:per-item", "--tie-per-second", "8", "--producer-max-len", "30"])
    cfg = line["config"]
    assert cfg["index"].startswith("avro") and cfg["avro"]["producer_tie_order"] == "per-item" and cfg["avro"]["tie_order_inference"] == "on"
    assert cfg["avro"]["sessions_named_by_some_list"] < cfg["avro"]["sessions_in_the_session_index"]      # sessions beyond the producer's length cut are in no list
    assert line["parity_checked"] > 0 and line["value"] > 0 and cfg["avro"]["share_on_the_fast_kernels"] > 0.95
