"""The driver's record of a bench run (BENCH_rNN.json) keeps the scalars of the JSON line plus the FIRST 24 scalar entries of
`config`, names cut at 40 characters (BENCH_r04.json: five descriptive strings in front cost every side row its place).
bench.order_config() puts the rows the verdicts track first; this test reads a line the way the driver does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

MUST_SURVIVE = ("workload", "value_is", "host_inclusive_full_payload_alignments_per_s", "one_launch_at_a_time_alignments_per_s", "one_launch_at_a_time_kernel_ms", "host_inclusive_alignments_per_s",
                "u_full_n20_10k_junctions_per_s", "u_full_n20_2k_junctions_per_s", "ins_svt4_junctions_per_s",
                "lr_c4_align_consensus_junctions_per_s", "lr_c4_msaedlib_n15_junctions_per_s", "lr_ins_msawfa_n15_junctions_per_s",
                "sr_stage_mixed_all_svt_junctions_per_s", "lr_stress_10kb_x_20kb_junctions_per_s",
                "substitutions_2pct_alignments_per_s", "substitutions_5pct_alignments_per_s",
                "lr_c4_align_consensus_8k_junctions_per_s", "lr_c4_msaedlib_n15_3k_junctions_per_s", "lr_ins_msawfa_n15_2k_junctions_per_s",
                "u_c2_40k_alignments_per_s")


def _line_like_bench():
    """a config in the order main() builds it: descriptive strings and bookkeeping first, the side rows appended at the end"""
    cfg = {"workload": "BASELINE configs[1]: ...", "workload_detail": "...", "launches_in_flight": 2, "junctions_per_gpu": 10000,
           "resident_batches": 4, "refined_ok_min": 9900, "parallelism": "junction-sharded x1", "ranks_launched": 1, "ranks_that_ran_kernels": 1,
           "value_is": "...", "kernels_ms_per_step_rank0": 0.3, "timed_regions": 25, "value_is_region": "...", "value_min": 1.0, "value_max": 2.0,
           "value_median": 1.5, "timed_seconds_total": 0.1, "one_launch_at_a_time_alignments_per_s": 3.0e7, "one_launch_at_a_time_ms_per_step": 0.3,
           "one_launch_at_a_time_kernel_ms": 0.26, "host_inclusive_alignments_per_s": 4.0e7, "host_inclusive_wall_s": 1.0,
           "host_inclusive_batches": 4000, "host_inclusive_ms_per_batch": 0.25, "host_inclusive_payload": "...",
           "host_inclusive_full_payload_alignments_per_s": 3.7e7, "host_recut_ms_per_batch_one_thread": 1.0}
    for _, key in bench.FLAT_ROWS:
        cfg[key] = 1.0
    cfg["u_full_n20_10k_msa_deferred_junctions"] = 0
    cfg["u_full_n20_10k_host_inclusive_per_s"] = 1.0
    for name, _ in bench.SWEEP_PLAN:
        cfg["deficit_sweep_%s_alignments_per_s" % name] = 1.0
    cfg["substitutions_2pct_alignments_per_s"] = cfg["substitutions_5pct_alignments_per_s"] = 1.0
    cfg["a_nested_thing"] = {"x": 1}
    return {"metric": "m", "value": 1.0, "config": bench.order_config(cfg)}


def test_the_rows_the_verdicts_track_survive_the_drivers_cut():
    line = json.loads(json.dumps(_line_like_bench()))          # (through JSON: key order is what the driver sees)
    kept = bench.driver_view_of_config(line["config"])
    assert len(kept) == bench.DRIVER_CONFIG_KEYS
    missing = [k for k in MUST_SURVIVE if k[:40] not in kept]
    assert not missing, missing
    assert sum(isinstance(v, str) for v in kept.values()) == 2  # the workload and which-figure-is-which; every other slot is a number


def test_first_keys_fit_the_cut_and_stay_distinct():
    for first in (bench.CONFIG_FIRST, bench.CONFIG_FIRST_MULTI):
        assert len(first) <= bench.DRIVER_CONFIG_KEYS
        cut = [k[:40] for k in first]
        assert len(set(cut)) == len(cut)
    flat = {key for _, key in bench.FLAT_ROWS}
    assert {k for k in MUST_SURVIVE if k.endswith("junctions_per_s")} <= flat | {"host_inclusive_alignments_per_s"}


def test_every_side_row_has_a_batch_recipe_the_parity_test_can_rebuild():
    names = [x[0] for x in bench.SIDE_PLAN]
    assert len(set(names)) == len(names)
    for name, n, ncpu, kw in bench.SIDE_PLAN:
        assert n > 0 and ncpu >= 0 and "mode" in kw
        assert n % int(kw.get("_tiles", 1)) == 0
    assert {row for row, _ in bench.FLAT_ROWS} <= set(names)


def test_an_n_gt_1_line_starts_with_what_the_return_paths_cost():
    """bench.py --gpus N: `value` is the pipelined shared-memory return, the blocking gather is timed beside it -- both must be among
    the scalars the driver keeps (SCALE_rNN.json is built from these lines)"""
    cfg = {"workload": "BASELINE configs[1]: ...", "workload_detail": "...", "launches_in_flight": 2, "junctions_per_gpu": 10000, "resident_batches": 4,
           "refined_ok_min": 9900, "parallelism": "junction-sharded x2", "ranks_launched": 2, "ranks_that_ran_kernels": 2, "value_is": "...",
           "kernels_ms_per_step_rank0": 0.5, "return_path": "...", "value_return_path": "shm", "ms_per_step_min_rank": 0.3, "ms_per_step_max_rank": 0.32,
           "rccl_ranks": 2, "gather_transport": "rccl", "oversubscribed_one_device": False, "gather_alignments_per_s": 2.0e7, "gather_step_ms": 0.9,
           "gather_ms_per_step": 0.85, "gathered_records_on_rank0": 20000, "gathered_blob_bytes_on_rank0": 17000000, "gather_path": "...",
           "shm_return_alignments_per_s": 6.0e7, "shm_return_ms_per_step": 0.32, "shm_return_gather_ms_per_step": 0.29,
           "shm_return_records_seen_by_rank0": 20000, "shm_return_blob_bytes_seen_by_rank0": 17000000, "shm_return_path": "...",
           "host_inclusive_alignments_per_s": 8.0e7, "segments_seen_by_rank0": [{"rank": 0}], "ms_per_step_per_rank": [0.3, 0.32]}
    line = json.loads(json.dumps({"config": bench.order_config(cfg)}))
    kept = bench.driver_view_of_config(line["config"])
    for k in ("value_return_path", "gather_alignments_per_s", "gather_ms_per_step", "gather_transport", "rccl_ranks", "shm_return_alignments_per_s",
              "shm_return_ms_per_step", "ranks_launched", "ranks_that_ran_kernels", "oversubscribed_one_device", "gathered_records_on_rank0",
              "shm_return_records_seen_by_rank0", "host_inclusive_alignments_per_s"):
        assert k[:40] in kept, k
    assert list(kept)[0] == "workload" and list(kept)[1] == "value_is" and list(kept)[2] == "value_return_path"


def test_traffic_stamp_follows_the_code_not_the_comments(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC pass that carries the hash of the headline kernel's sources
    (delly_amd/build.py: headline_kernel_hash); `traffic_stale` must flip when the code changes and must NOT when a comment
    does (renewing the stamp costs a GPU pass) -- and the stamp committed with this tree must be the tree's."""
    import shutil
    from delly_amd import build
    here = build.headline_kernel_hash()
    t = bench._measured_traffic()
    assert t["traffic"] and t["stale"] is False, "profiles/rNN/pmc_traffic.json was collected on other kernel sources: %r" % (t,)
    for f in build.HEADLINE_KERNEL_SOURCES:
        shutil.copy(os.path.join(build.CSRC, f), tmp_path / f)
    monkeypatch.setattr(build, "CSRC", str(tmp_path))
    assert build.headline_kernel_hash() == here
    p = tmp_path / build.HEADLINE_KERNEL_SOURCES[0]
    src = p.read_text()
    p.write_text("// a reworded remark\n\n" + src.replace("\n", "   // trailing remark\n", 1))
    assert build.headline_kernel_hash() == here
    p.write_text(src + "\nnamespace dh { constexpr int dh_probe_of_the_stamp = 1; }\n")
    assert build.headline_kernel_hash() != here
