"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and
`cpu_baseline` objects.  Run here on the smallest BASELINE configuration (the reference's own CPU-runnable case)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("DTW cell-updates/sec") and d["unit"] == "cell-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("cfg1") and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["cells_per_step"] - 1.0) < 1e-6     # value = cells / step time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "cell-updates/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["parity_ok"] is True and d["max_rel_err_vs_oracle"] <= 1e-4
