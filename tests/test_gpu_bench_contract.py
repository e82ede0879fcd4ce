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
    # no torch in the data path: buffers, streams and fences come from the library; the runtime is the one it links
    assert d["config"]["torch_imported"] is False and "/opt/rocm" in d["config"]["runtime"]
    assert d["config"]["collective_fallback"] is False and d["config"]["ranks_seen"] == 1
    # the full-matrix census of the default distance form against the bit-exact strict mode
    pc = d["parity_census"]
    assert pc["entries"] == 64 * 64 and pc["over_1e-4"] == 0 and pc["max_rel"] <= 1e-4
    assert pc["nonfinite_pattern_equal"] and pc["zero_pattern_equal"] and pc["strict_max_rel_err_vs_oracle"] == 0.0


def test_bench_secondary_workloads_and_in_process_multi_device():
    """`--gpus 1 --launcher inprocess`-style multi handle at one device is covered by tests/test_gpu_multi.py; here: the
    `secondary` object (cfg1 measured next to another workload in the same run)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "0", "--secondary", "on"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["workload"].startswith("cfg2") and d["parity_census"]["over_1e-4"] == 0
    for name in ("cfg2", "cfg1", "cfg4", "cfg5s"):
        s = d["secondary"][name]
        assert s["workload"].startswith(name) and s["kernel_ms"] > 0 and 0 < s["roofline_frac"] < 1 and s["parity_ok"] is True
    for name in ("cfg4", "cfg5s"):                                    # the driver-visible BASELINE shapes beyond cfg3: own census, own strict pass
        s = d["secondary"][name]
        pc = s["parity_census"]
        assert pc["entries"] == (4096 if name == "cfg4" else 512) ** 2 and pc["nonfinite_pattern_equal"] and pc["zero_pattern_equal"]
        # the default form's one known deviation: a handful of entries at most, each below 3e-3
        assert pc["over_1e-4"] <= 8 and pc["max_rel"] <= 3e-3, pc
        assert s["strict"]["kernel_ms"] > s["kernel_ms"] and s["strict"]["bitwise_equal_to_oracle_sample"] == "32 of 32"
    assert "cfg3_strict" not in d["secondary"]                        # only next to the cfg3 headline (it reuses that run's census pass)
    # the same shape measured twice in one run agrees with itself
    assert abs(d["secondary"]["cfg2"]["kernel_ms"] / d["roofline"]["kernel_ms"] - 1.0) < 0.15


def test_bench_gpus_2_in_process_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` with WORLD_SIZE unset -- the driver's N > 1 command -- end to end on a one-GPU box: two ranks of
    the in-process handle on device 0 (APD_BENCH_DEVICES), slabs gathered by the peer-copy collective (RCCL refuses two ranks on
    one device).  Checks the line's N > 1 fields; the timing means nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(APD_MULTI_COLLECTIVE="peer", APD_BENCH_DEVICES="0,0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["ranks_seen"] == 2
    # N > 1 lines: the run verified itself before timing (same batch on N devices and on device 0 alone), the CPU baseline is
    # there for every N, and the recorded single-GPU counters are withheld with the reason
    assert d["config"]["multi_selfcheck"] == "bitwise"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_withheld"]
    assert d["config"]["launch"].startswith("one process driving 2 devices") and d["config"]["torch_imported"] is False
    assert d["config"]["collective_fallback"] is True and "peer-copy" in d["config"]["collective_error"]
    assert d["parity_ok"] is True and d["parity_census"]["over_1e-4"] == 0
    assert "2 ranks" in d["config"]["sharding"] and d["roofline"]["kernel"].startswith("dtw_fused (rank 0 share")
