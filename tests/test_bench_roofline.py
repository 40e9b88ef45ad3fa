"""CPU tests of the measurement code in bench.py (no GPU, no oracle): the executed-work roofline arithmetic on a
synthetic counter summary, the committed round-3 PMC summaries and bench lines (their roofline blocks must be
reproducible by hand from the summaries), and the source stamp."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_executed_roofline_arithmetic_and_bound():
    # 1e10 flop over 1 GB raw fetch + 0.5 GB written in 1 ms: traffic = 2 x 1 + 0.5 = 2.5 GB -> 2.5 TB/s; AI = 4 flop/B: hbm
    pm = {"fp64_flop_issued_per_launch": 1e10, "fp64_mfma_flop_per_launch": 6e9, "hbm_fetch_bytes_per_launch": 1e9,
          "hbm_write_bytes_per_launch": 5e8, "lane_utilisation": 0.5}
    r = bench.executed_roofline(pm, 1e-3)
    assert r["traffic"] == pytest.approx(2.5e9) and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert r["achieved"] == pytest.approx(2500.0) and r["frac"] == pytest.approx(2500.0 / 8000.0) == r["hbm_frac"]
    assert r["fp64_frac"] == pytest.approx(10.0 / 78.6) and r["mfma_share_of_flop"] == pytest.approx(0.6)
    assert r["arithmetic_intensity_flop_per_byte"] == pytest.approx(4.0)
    # the same work with a tenth of the bytes sits right of the ridge (AI 40 > 9.8): the quadruple is the flop one
    pm2 = dict(pm, hbm_fetch_bytes_per_launch=1e8, hbm_write_bytes_per_launch=5e7)
    r2 = bench.executed_roofline(pm2, 1e-3)
    assert r2["bound"] == "mfma" and r2["unit"] == "TFLOP/s" and r2["frac"] == pytest.approx(10.0 / 78.6)
    # no traffic counters: the label falls back to the flop mix
    r3 = bench.executed_roofline({"fp64_flop_issued_per_launch": 1e10, "fp64_mfma_flop_per_launch": 1e9}, 1e-3)
    assert r3["bound"] == "fp64-valu" and r3["traffic"] is None


@pytest.mark.parametrize("tag,cfg", [("r03", 3), ("r03_cfg4", 4), ("r03_cfg5", 5)])
def test_committed_bench_lines_follow_from_the_committed_counters(tag, cfg):
    """profiles/<tag>_bench.json.log against profiles/<tag>_pmc_summary.json: the line names that summary, is not
    stale, and its roofline numbers are the summary's counters divided by the line's own launch time."""
    line = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json.log")).read().strip().splitlines()[-1])
    sm = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")))
    assert bench.PMC_SUMMARIES[cfg].endswith(f"{tag}_pmc_summary.json")
    roof = line["roofline"]
    assert roof["pmc_stale"] is False and roof["bound"] in ("hbm", "mfma") and 0.0 < roof["frac"] < 1.0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "reference"
    assert line["config"]["baseline_config"] == cfg - 1 and line["n_gpus"] == 1 and line["dtype"] == "f64"
    k = roof["kernel"]
    if "launches_per_step" in roof:  # the Newton linear step: all its kernels together, per step
        ks = k.split(" + ")
        flop = sum(sm["kernels"][q]["fp64_flop_issued_per_launch"] * sm["kernels"][q]["launches"] for q in ks) / sm["searches"]
        byts = sum((2.0 * sm["kernels"][q]["hbm_fetch_bytes_per_launch"] + sm["kernels"][q]["hbm_write_bytes_per_launch"])
                   * sm["kernels"][q]["launches"] for q in ks) / sm["searches"]
        assert roof["fp64_flop_issued_per_step"] == pytest.approx(flop)
        assert roof["fp64_achieved_TFLOPs"] == pytest.approx(flop / (roof["ms_per_step"] * 1e-3) / 1e12)
        assert roof.get("traffic_per_step", roof["traffic"]) == pytest.approx(byts)
        if roof["bound"] == "hbm":
            assert roof["achieved"] == pytest.approx(byts / (roof["ms_per_step"] * 1e-3) / 1e9) and roof["unit"] == "GB/s"
    else:
        pm = sm["kernels"][k]
        again = bench.executed_roofline(pm, roof["avg_launch_ms"] * 1e-3)
        for f in ("achieved", "frac", "traffic", "fp64_frac"):
            assert roof[f] == pytest.approx(again[f]), f
    # the sum of the line's kernel time classes is the step
    assert sum(line["kernels_ms_per_step"].values()) == pytest.approx(line["ms_per_step"], rel=1e-6)
    assert line["value"] > {3: 20000, 4: 15000, 5: 2000}[cfg]


def test_source_stamp_matches_the_committed_summaries():
    """The summaries carry the hash of emme_amd/csrc/*.h* they were collected on: equal to the tree's, or bench.py
    will (rightly) call them stale."""
    h = bench.kernel_source_sha16()
    for cfg, path in bench.PMC_SUMMARIES.items():
        assert json.load(open(path))["source_sha16"] == h, (cfg, path)
