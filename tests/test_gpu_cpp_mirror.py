"""The C++ mirror of the reference's Rust surface (include/apd.hpp) driven by tests/cpp/harness.cpp, vs the oracle."""
import os
import subprocess

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_end_to_end(oracle, tmp_path):
    exe = os.path.join(ROOT, "build", "apd_cpp_harness")
    if not os.path.exists(exe):
        pytest.fail("build/apd_cpp_harness missing: __graft_entry__.build() compiles it")
    n, dim, pct, perc = 14, 13, 0.0625, 0.2
    frames, offsets = synth.make_sequences(n, 40, dim, seed=31, copies=0.5)
    with open(tmp_path / "in.txt", "w") as fp:
        fp.write("%d %d %r 1.0 1.0 1.0 %r\n" % (n, dim, pct, perc))
        for s in synth.split(frames, offsets):
            fp.write("%d\n%s\n" % (len(s), " ".join(repr(float(v)) for v in s.ravel())))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "audio_pattern_discovery_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, str(tmp_path / "in.txt")], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    dist = np.array(lines[0].split()[1:], dtype=np.float32).reshape(n, n)
    want = oracle.align_all(frames, offsets, pct, workers=4)
    np.testing.assert_allclose(dist, want, rtol=1e-4, atol=0)
    # clustering is compared on the GPU's own matrix (bit-exact vs the literal algorithm on identical input)
    ops, roots, _ = oracle.clustering(dist, n, perc)
    got_ops = [tuple(l.split()[1:4]) + (l.split()[5],) for l in lines if l.startswith("op ")]
    assert got_ops == [(str(o["merge_i"]), str(o["merge_j"]), str(o["into"]), str(oracle.MERGE_NAMES.index(o["operation"]))) for o in ops]
    assert [int(v) for v in [l for l in lines if l.startswith("roots")][0].split()[1:]] == roots
    sets = [[int(v) for v in l.split()[1:]] for l in lines if l.startswith("set")]
    assert sets == oracle.cluster_sets(ops, roots, n)
    pair = float([l for l in lines if l.startswith("pair01")][0].split()[1])
    assert abs(pair - want[0, 1]) <= 1e-4 * want[0, 1]
