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
    assert [l for l in lines if l.startswith("multi ")][0].split()[1:] == ["1", "1"]     # the multi-device handle on {0}: one rank, same bits
    assert [l for l in lines if l.startswith("multi_again")][0].split()[1:] == ["1", "rccl"]   # kept handle, second call, RCCL collective
    assert [l for l in lines if l.startswith("onecall")][0].split()[1:] == ["1", "1"]     # apd_dtw_all_pairs / apd_upgma: same bits


def test_cpp_mirror_with_the_references_on_disk_artefacts(oracle, tmp_path):
    """apd::AutoEncoder::from_file (bincode auto_encoder.bin, neural.rs:30-36) -> encoded() -> align_all -> clustering ->
    apd::dendrograms, parameters from apd::Discovery::from_toml (discovery.rs:28-36): main.rs:142-203 through the C++ mirror.
    PARITY UNPINNED for the file formats: no reference-made file exists; the weight file is written by the Python mirror's
    writer (same C entry point) and by hand-packed bytes in tests/test_host_abi.py."""
    import struct
    exe = os.path.join(ROOT, "build", "apd_cpp_harness")
    if not os.path.exists(exe):
        pytest.fail("build/apd_cpp_harness missing: __graft_entry__.build() compiles it")
    n, dim, latent, pct, perc = 12, 13, 8, 0.25, 0.5
    frames, offsets = synth.make_sequences(n, 36, dim, seed=77, copies=0.5)
    rng = np.random.default_rng(5)
    w = ((rng.random((dim, latent)) - 0.5) / latent).astype(np.float32)
    b = ((rng.random(latent) - 0.5) / latent).astype(np.float32)
    wd = rng.standard_normal((latent, dim)).astype(np.float32)
    bd = rng.standard_normal(dim).astype(np.float32)
    blob = b""
    for m, cols in ((w, latent), (wd, dim), (b, latent), (bd, dim)):                 # hand-packed bincode 1.x image
        blob += struct.pack("<Q", m.size) + m.astype("<f4").tobytes() + struct.pack("<Q", cols)
    (tmp_path / "auto_encoder.bin").write_bytes(blob)
    (tmp_path / "Discovery.toml").write_text(
        "dft_win = 256\ndft_step = 128\nceps_filter = 18\nauto_encoder = %d\nlearning_rate = 0.1\nepochs = 25\nepoch_drop = 5.0\n"
        "drop = 0.5\nvat_moving = 15\nvat_percentile = 0.95\nvat_min_len = 150\nwarping_band_percentage = %r  # band\n"
        "insertion_penalty = 1.0\ndeletion_penalty = 1.0\nmatch_penalty = 1.0\nalignment_workers = 4\nclustering_percentile = %r\n"
        % (latent, pct, perc))
    with open(tmp_path / "in.txt", "w") as fp:
        fp.write("%d %d 1.0 1.0 1.0 1.0 0.05\n" % (n, dim))                        # overridden by the TOML file
        for s_ in synth.split(frames, offsets):
            fp.write("%d\n%s\n" % (len(s_), " ".join(repr(float(v)) for v in s_.ravel())))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "audio_pattern_discovery_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "auto_encoder.bin"), str(tmp_path / "Discovery.toml")],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    get = lambda key: [l for l in lines if l.startswith(key + " ")][0].split()[1:]
    toml = get("toml")
    assert [int(toml[k]) for k in (0, 1, 2, 3, 5, 6, 12, 14)] == [256, 128, 18, 15, 150, 4, latent, 25]
    assert np.float32(toml[8]) == np.float32(pct) and np.float32(toml[7]) == np.float32(perc)
    assert get("latent") == [str(latent)] and get("roundtrip") == ["1"]
    enc = oracle.encode(frames, w, b)
    np.testing.assert_allclose(np.array(get("enc0"), np.float32).reshape(-1, latent), enc[:int(offsets[1])], rtol=1e-5, atol=1e-5)
    dist = np.array(get("dist"), dtype=np.float32).reshape(n, n)
    np.testing.assert_allclose(dist, oracle.align_all(enc, offsets, pct, workers=4), rtol=1e-4, atol=1e-6)
    ops, roots, _ = oracle.clustering(dist, n, float(np.float32(perc)))
    labels = ["{img%d}" % i for i in range(n)]
    results = {}
    for o in ops:
        i, j, k = o["merge_i"], o["merge_j"], o["into"]
        left = results[i] if o["operation"] in ("Cluster2Sequence", "Cluster2Cluster") else labels[i]
        right = results[j] if o["operation"] in ("Sequence2Cluster", "Cluster2Cluster") else labels[j]
        results[k] = "[.%d [%s %s ] ]" % (k, left, right)
    got = {int(l.split(" ", 2)[1]): l.split(" ", 2)[2] for l in lines if l.startswith("dendro ")}
    assert got == {r: results[r] for r in roots if r in results}
