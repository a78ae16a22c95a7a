"""-m gpu: the driver's contract with bench.py, checked on hardware with a shortened run: ONE JSON line on stdout (last line), the contract's
keys with the right types, `roofline` and `cpu_baseline` objects, the timed outputs checked, the bounded objects of the other BASELINE
configs, and the compact `summary` as the LAST key (the driver keeps the tail of stdout)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_workload_line_contract(dev):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--cpu-sample", "2",
                        "--no-cpu-all-cores", "--no-live-traffic", "--no-io-rates"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    d = json.loads(lines[-1])                                   # the result is the LAST line (RCCL / library banners may precede it)
    assert sum(1 for ln in lines if ln.startswith("{")) == 1
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", (int, float)),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "images/s" and "workload" in d["config"]
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01          # whole-job images/s == batch / step time
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0.2 < roof["frac"] < 1.0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    # what was timed was checked, and the parity legs look at the timed outputs
    assert d["timed_outputs_checked"] is True and d["parity"]["timed_outputs_bitwise_equal_eager"] is True
    assert d["parity"]["logit_max_abs_err"] < 2e-5 and d["parity"]["unexplained_label_mismatches"] == 0
    sp = d["second_precision"]
    assert sp["mode"] == "fast" and sp["parity"]["timed_outputs_bitwise_equal_eager"] is True and sp["parity"]["logit_max_abs_err"] < 1e-3
    # the other BASELINE configs, bounded, in the same line
    assert d["c4"]["timed_outputs_bitwise_equal_eager"] is True and d["c4"]["parity"]["logit_max_abs_err"] < 2e-5 and d["c4"]["value"] > 0
    assert d["c5"]["precision"] == "fast" and d["c5"]["parity"]["embedding_max_abs_err"] < 1e-3 and d["c5"]["roofline"]["peak"] == 2500.0
    assert d["bilateral_solver"]["batch1"]["ms_per_image"] > 0 and d["pseudo_labels"]["value"] > 0 and d["batch1"]["parity"]["category_list_identical"]
    # the pseudo-label path is measured like the headline (round 6): roofline of its dominant kernel, parity of the final mask against the
    # oracle chain, a bounded CPU baseline
    pl = d["pseudo_labels"]
    assert pl["roofline"]["bound"] == "mfma" and "attn_f16_kernel" in pl["roofline"]["kernel"] and 0.15 < pl["roofline"]["frac"] < 1.0
    assert abs(pl["roofline"]["frac"] - pl["roofline"]["achieved"] / pl["roofline"]["peak"]) < 1e-3
    assert pl["parity"]["differing_pixels_off_the_oracle_contour"] == 0 and pl["parity"]["differing_pixels"] <= 200
    assert pl["parity"]["selected_query_identical"] is True and pl["cpu_baseline"]["kind"] == "port" and pl["cpu_baseline"]["value"] > 0
    assert pl["value_noise_images"] > 0 and "natural" in pl["images"]
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) < 1500
    s = d["summary"]
    assert s["checked"] is True and s["c4"]["ok"] is True and "c5_fast" in s and "solver_ms" in s and "b1_ms" in s and s["pseudo"]["bad_px"] == 0
