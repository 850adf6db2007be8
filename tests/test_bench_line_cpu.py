"""bench.py's ONE stdout line: built from a canned full result object, it must stay under 8 KB (the driver stored round 5's 27 KB line unparsed)
and carry the contract's keys, the roofline with every search kernel's own figures and the CPU baseline."""
import json

import bench


def canned(legs=3):
    pk = {k: {"ms_per_step": 45.66201343536377, "launches_per_step": 4, "rank_queries": 1650217632, "rank_queries_needed": 1650217000, "lt_lookups": 111356798,
              "sa_reads": 2098346117, "ce_windows": 3349792595, "algorithmic_bytes": 53166964224.0, "algorithmic_bytes_extended": 54948672992.0,
              "achieved_GBps": 1164.35, "frac": 0.14554484193754333, "frac_extended": 0.15042227898519472, "fetched_bytes": 87450929536.0,
              "fetched_GBps": 1915.17, "fetched_Glines_per_s": 29.92} for k in ("first", "trunk", "unit", "ce")}
    e2e = {"counts": {"pairs": 10_000_000, "Number of pairs aligned by DP": 2832711}, "phases_s": {"Alignment + output time": 2.88, "Loading time": 1.2},
           "M_pairs_per_s_reads_in_sam_out": 3.47, "dp": {"GCUPS": 375.0}, "device_ms": {"bwt_stages": 2674.2, "bwt_search_kernels": 1370.6, "dp_kernels": 868.2},
           "text": "x" * 3000}
    leg = {"pairs": 10_000_000, "end_to_end": e2e, "bwt_stage_resident": {"roofline": {"frac": 0.17}},
           "truth_check": {"pairs": 10_000_000, "pairs_with_both_mates_at_simulated_place": 9897910}, "long_text": "y" * 4000}
    out = {"metric": "M reads/sec aligned (100 bp SE, GRCh38-scale synthetic genome)", "value": 42.57186022830793, "unit": "M reads/s", "n_gpus": 1, "steps": 20,
           "warmup": 5, "ms_per_step": 234.89694710006006, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "w" * 900, "genome_bp": 3_100_000_000, "reads_per_gpu": 10_000_000, "read_len": 100, "max_mismatch": 4, "max_output_per_read": 1000,
                      "repeat_frac": 0.5, "repeat_model": "grch38-like", "repeat_families": {"a": {"unit": 300, "copies": 1117117}}, "parallelism": "reads sharded x1",
                      "lookup_chars": 16},
           "roofline": {"bound": "hbm", "achieved": 1373.5, "peak": 8000.0, "unit": "GB/s", "frac": 0.1717, "traffic": 130545241952.0, "achieved_is": "z" * 300,
                        "peak_measured": 5087.9, "frac_of_measured": 0.27, "frac_extended": 0.22, "traffic_measured_at_commit": "45e6f63af99f",
                        "algorithmic_bytes_per_launch": 72317157055.0, "launches_per_step": 4, "kernel_ms_per_step": 210.6, "rank_queries_per_step": 9779166606,
                        "rank_queries_algorithmic": 8911099556.0, "per_kernel": pk, "per_kernel_is": "t" * 700, "dominant_kernel": "unit",
                        "random_line_rate": {"achieved_Glines_per_s": 38.7, "probe_Glines_per_s": 48.2, "frac": 0.80}},
           "cpu_baseline": {"value": 0.1108, "unit": "M reads/s", "cores": 16, "kind": "port", "sample": "s" * 400, "reads_per_s_per_core": 6928.6,
                            "reference": {"kind": "reference", "what": "r" * 300, "cores": 1, "value": 0.294, "unit": "M reads/s", "port_over_reference": 0.66}},
           "truth_check": {"ok": True}, "oracle_check": {"reads": 16384, "bit_exact": True, "what": "o" * 300},
           "streamed": {"same_results_as_resident": True, "vs_resident": 0.85, "batches_of_1Mi": {"vs_resident": 0.79}},
           "occurrences": 312786869, "wall_s_total": 316.0,
           "extras": {"c3_pairs": leg, "c5_pairs": leg, "uniform_genome": {"value": 648.1, "blob": "u" * 5000},
                      "repeat_models": {m: {"M_reads_per_s": 29.1, "ms_per_step": 342.9, "blob": "b" * 1000} for m in ("young", "old", "satellite-heavy")[:legs]}}}
    return out


def test_the_line_is_short_and_complete():
    out = canned()
    out["summary"] = bench.summary_of(out)
    assert len(json.dumps(out)) > 20_000                                # the full object is what it was: tens of kilobytes
    txt = bench.compact_line(out, "bench_full.json")
    assert len(txt) < bench.COMPACT_LIMIT == 8192 and "\n" not in txt
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "summary"):
        assert k in line, k
    assert line["value"] == 42.5719 and line["steps"] == 20 and line["warmup"] == 5 and line["vs_baseline"] is None
    assert set(line["config"]) >= {"workload", "genome_bp", "reads_per_gpu", "read_len", "max_mismatch"} and len(line["config"]["workload"]) <= 400
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_per_step", "per_kernel", "traffic_measured_at_commit"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert set(roof["per_kernel"]) == {"first", "trunk", "unit", "ce"}
    for v in roof["per_kernel"].values():
        assert set(v) >= {"ms_per_step", "rank_queries", "rank_queries_needed", "algorithmic_bytes", "fetched_bytes", "frac"}
    cpu = line["cpu_baseline"]
    assert set(cpu) >= {"value", "unit", "cores", "kind", "sample", "reference"} and cpu["kind"] == "port" and cpu["reference"]["kind"] == "reference"
    assert line["summary"]["c3_pairs"]["seconds"] == 2.88 and line["summary"]["c5_pairs"]["M_pairs_per_s"] == 3.47
    assert line["summary"]["uniform_genome_M_reads_per_s"] == 648.1 and line["parity"]["oracle_check_bit_exact"] is True


def test_the_line_fits_even_when_the_legs_grow():
    out = canned()
    out["extras"]["repeat_models"] = {"model_%d" % i: {"M_reads_per_s": 1.0 + i, "ms_per_step": 100.0 + i} for i in range(200)}
    for leg in ("c3_pairs", "c5_pairs"):
        out["extras"][leg]["end_to_end"]["device_ms"] = {"k%d" % i: float(i) for i in range(100)}
    txt = bench.compact_line(out, None)
    assert len(txt) < bench.COMPACT_LIMIT
    line = json.loads(txt)
    assert "roofline" in line and "cpu_baseline" in line and line["value"] == 42.5719      # what is dropped is the summary, never the contract


def test_a_failed_leg_is_named_not_dropped():
    out = canned()
    out["extras"]["c3_pairs"] = {"error": "RuntimeError('no device memory')"}
    line = json.loads(bench.compact_line(out, None))
    assert "no device memory" in line["summary"]["c3_pairs"]["error"]
