"""The committed bench line and the rocprofv3 summaries it is judged against (profiles/<bench.ROUND>/) must tell one story: the contract's keys,
value = units / time, roofline.achieved = algorithmic bytes per launch / the launch's average duration, frac = achieved / peak, and the
kernel-trace average of the same launch shape within a few per cent of the HIP-event figure inside bench.py.  Runs on the CPU: it reads
files only -- a guard against a line refreshed without its profile, or a document quoting numbers the files no longer hold."""
import csv
import json
import os

import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = re.search(r'^ROUND = "(r\d+)"', open(os.path.join(ROOT, "bench.py")).read(), re.M).group(1)     # the round bench.py says its evidence is of
P = os.path.join(ROOT, "profiles", ROUND)


def test_the_bench_line_keeps_the_contract_and_its_arithmetic():
    d = json.load(open(os.path.join(P, "bench_default_1gpu.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    c = d["config"]
    w, h = c["tile"]
    cands = w * h * c["ndisp"] * c["tiles_per_step"]                      # candidates one step processes
    assert abs(d["value"] - cands / 1e6 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3          # Mdisp/s = units / time
    assert c["confidence"] is True and c["recursion"] == 2                # the drop-in's mode, its confidence image inside the timed region
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    alg = r["alg_bytes_per_candidate"] * w * h * c["ndisp"] * r["tiles_per_launch"]
    assert alg == r["alg_bytes_per_launch"]
    assert abs(r["achieved"] - alg / 1e9 / (r["avg_launch_ms"] * 1e-3)) / r["achieved"] < 2e-3
    assert r["traffic"] is None or 0.9 * alg < r["traffic"] < 1.5 * alg   # PMC bytes: the algorithmic ones + the hand-off rows, no wasted re-reads
    b = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in b, k
    assert b["kind"] in ("reference", "port") and b["cores"] >= 1
    pool = d["pool"]
    assert pool["workers"] >= 16 and pool["tiles_per_s"] >= 1000.0        # VERDICT r03 item 1's bar, in the line the driver records


def test_the_kernel_trace_agrees_with_the_bench_line():
    d = json.load(open(os.path.join(P, "bench_default_1gpu.json")))
    rows = list(csv.DictReader(open(os.path.join(P, "census_mgm3_b8_1024x1024x128_kernel_stats.csv"))))
    band = [r for r in rows if "k_mgm_bands" in r["Name"]]
    assert len(band) == 1 and int(band[0]["Calls"]) >= 8
    avg_ms = float(band[0]["AverageNs"]) / 1e6
    assert abs(avg_ms - d["roofline"]["avg_launch_ms"]) / avg_ms < 0.05  # rocprofv3's average and the HIP events inside bench.py (boxes differ by a few per cent)
    pmc = json.load(open(os.path.join(P, "census_mgm3_b8_1024x1024x128_pmc_fetch_write.json")))
    k = [v for n, v in pmc.items() if "k_mgm_bands" in n]
    assert len(k) == 1
    cal = json.load(open(os.path.join(P, "pmc_calibration.json")))        # known-byte calibration of the SAME round, at the band kernel's access width (8 B per lane)
    f, wf = cal["calib_read_b64_band"]["FETCH_SIZE_bytes_over_known"], cal["calib_write_b64_band_nt"]["WRITE_SIZE_bytes_over_known"]
    assert abs(f - 0.5) < 0.01 and abs(wf - 1.0) < 0.01
    traffic = (k[0]["FETCH_SIZE_KiB_avg"] / f + k[0]["WRITE_SIZE_KiB_avg"] / wf) * 1024.0
    r = d["roofline"]
    assert abs(traffic - r["traffic"]) / traffic < 1e-3                   # bench.py fills roofline.traffic from these very files
    assert r["traffic_same_round"] is True and r["traffic_calibrated"] is True and ("profiles/%s/" % ROUND) in r["traffic_source"]


def test_the_in_flight_figure_is_of_this_round_and_call_shape():
    """roofline.in_flight.measured_* may only quote a kernel trace of THIS round's headline command."""
    d = json.load(open(os.path.join(P, "bench_default_1gpu.json")))
    fl = d["roofline"]["in_flight"]
    nb = d["config"]["tiles_per_call"]
    name = "mgm_inflight_b%d_%dx%dx%d.json" % (nb, d["config"]["tile"][0], d["config"]["tile"][1], d["config"]["ndisp"])
    assert fl["measured_source"] == "profiles/%s/%s" % (ROUND, name) and fl["measured_tiles_per_launch"] == nb
    u = json.load(open(os.path.join(P, name)))
    assert abs(u["union_ms_per_launch"] - fl["measured_union_ms_per_launch"]) < 1e-6
    # the union per launch lies between the launch alone and its mean duration with three calls overlapping
    assert d["roofline"]["avg_launch_ms"] * 0.95 < u["union_ms_per_launch"] < u["mean_duration_ms_in_flight"]
    assert abs(fl["measured_frac"] - d["roofline"]["alg_bytes_per_launch"] / (u["union_ms_per_launch"] * 1e-3) / 1e9 / 8000.0) < 2e-3


def test_the_hbm_side_figure_is_labelled_a_model():
    d = json.load(open(os.path.join(P, "bench_default_1gpu.json")))
    r = d["roofline"]
    assert "frac_hbm" not in r and r["hbm_model"].startswith("MODEL, not measured") and "frac_hbm_model" in r
    ac = d["cpu_baseline"]["all_cores"]
    assert ac["cores"] == ac["host_cores"] or "bounded by memory" in ac["sample"]     # N = nproc unless memory says otherwise
