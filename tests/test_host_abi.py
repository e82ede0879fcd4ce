"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol include/apd.h
declares (no compute call is made: there is no GPU here), host-only entry points behave like the Rust
items they replace, and the library refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "apd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(apd):
    L = apd.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(L, name), "libapd_hip.so does not export %s" % name
    bound = {n for n, _, _ in apd.SYMBOLS}
    assert set(names) == bound, "ctypes table and apd.h disagree: %s" % (set(names) ^ bound)


def test_no_cpu_fallback(apd):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(apd.ApdError) as e:
        apd.Context(0)
    assert e.value.status == apd.APD_ERR_NO_DEVICE


def test_status_strings(apd):
    L = apd.lib()
    assert L.apd_status_string(0) == b"ok"
    assert b"alignments.rs:120" in L.apd_status_string(apd.APD_ERR_EMPTY_SEQUENCE)
    assert b"numerics.rs:132" in L.apd_status_string(apd.APD_ERR_INDEX)


def test_discovery_alignment_params_matches_oracle(oracle):
    from audio_pattern_discovery_amd.discovery import Discovery
    d = Discovery(warping_band_percentage=0.0625, insertion_penalty=0.5, deletion_penalty=1.5, match_penalty=0.75)
    for n in (0, 1, 15, 16, 17, 512, 1055, 2048, 1 << 20):
        p = d.alignment_params(n)
        assert p.warping_band == oracle.warping_band(0.0625, n)
        assert (p.insertion_penalty, p.deletion_penalty, p.match_penalty) == (0.5, 1.5, 0.75)
    assert Discovery(warping_band_percentage=-1.0).alignment_params(100).warping_band == 0
    assert Discovery(warping_band_percentage=float("nan")).alignment_params(100).warping_band == 0
    assert Discovery(warping_band_percentage=0.1).alignment_params(20).warping_band == oracle.warping_band(0.1, 20)


def test_discovery_from_toml(tmp_path):
    from audio_pattern_discovery_amd.discovery import Discovery
    text = """# FFT Spec
dft_win       = 256                  # DFT window
dft_step      = 128
ceps_filter   = 32
auto_encoder  = 10
learning_rate = 0.1
epochs        = 25
epoch_drop    = 5.0
drop          = 0.5
vat_moving     = 15
vat_percentile = 0.95
vat_min_len    = 150
warping_band_percentage = 0.0625
insertion_penalty       = 1.0
deletion_penalty        = 1.0
match_penalty           = 1.0
alignment_workers       = 4
clustering_percentile   = 0.05
"""
    f = tmp_path / "Discovery.toml"
    f.write_text(text)
    d = Discovery.from_toml(str(f))
    assert d.dft_win == 256 and d.warping_band_percentage == 0.0625
    assert d.clustering_percentile == np.float32(0.05) and d.vat_percentile == np.float32(0.95)     # the reference's fields are f32
    assert (d.dft_step, d.ceps_filter, d.auto_encoder, d.epochs, d.vat_moving, d.vat_min_len, d.alignment_workers) == (128, 32, 10, 25, 15, 150, 4)
    assert (d.learning_rate, d.epoch_drop, d.drop) == (np.float32(0.1), 5.0, 0.5)
    for bad in ("dft_win = 1\n",                                        # 16 fields missing (serde: missing field)
                text + "dft_win = 3\n",                                  # duplicate key
                text.replace("dft_step      = 128", "dft_step      = 128.5"),   # a float where usize is wanted
                text.replace("epochs        = 25", "epochs        = -25"),
                text.replace("drop          = 0.5", "drop          = half")):
        (tmp_path / "bad.toml").write_text(bad)
        with pytest.raises(KeyError):
            Discovery.from_toml(str(tmp_path / "bad.toml"))
    # keys the struct does not have are IGNORED: `#[derive(Deserialize)] struct Discovery` (discovery.rs:7) has no
    # deny_unknown_fields, so toml::from_str accepts them -- and a [table] hides what follows it from the top-level fields
    for extra in (text + "colour = 3\n", "note = \"a string\"\n" + text, text + "[plots]\ndft_win = 7\nwidth = 3\n"):
        (tmp_path / "extra.toml").write_text(extra)
        e = Discovery.from_toml(str(tmp_path / "extra.toml"))
        assert (e.dft_win, e.alignment_workers, e.warping_band_percentage) == (256, 4, 0.0625)
    (tmp_path / "bad.toml").write_text(text + "colour\n")                   # not `key = value`
    with pytest.raises(KeyError):
        Discovery.from_toml(str(tmp_path / "bad.toml"))
    # the file the reference ships parses (project/config/Discovery.toml, reproduced in SURVEY.md section 5: band 1.0, workers 4)
    (tmp_path / "shipped.toml").write_text(text.replace("0.0625", "1.0       # sakoe shiba band"))
    assert Discovery.from_toml(str(tmp_path / "shipped.toml")).warping_band_percentage == 1.0
    with pytest.raises(OSError):
        Discovery.from_toml(str(tmp_path / "missing.toml"))


def test_align_work_counts_the_reference_loops(oracle):
    from audio_pattern_discovery_amd import _lib
    from audio_pattern_discovery_amd.alignments import align_work
    frames, offsets = synth.make_sequences(37, 60, 13, seed=8, jitter=20)
    for pct in (1.0, 0.0625, 0.0):
        cfg = _lib.AlignConfig(pct, 1, 1, 1)
        pairs, cells, nbytes = align_work(offsets, 13, cfg)
        want_cells = want_bytes = 0
        for i in range(37):
            for j in range(37):
                if i != j:
                    n, m = int(offsets[i + 1] - offsets[i]), int(offsets[j + 1] - offsets[j])
                    want_cells += oracle.dtw_cells(n, m, oracle.warping_band(pct, max(n, m)))
                    want_bytes += 4 * 13 * (n + m) + 4
        assert (pairs, cells, nbytes) == (37 * 36, want_cells, want_bytes)
        # shards add up
        parts = [align_work(offsets, 13, cfg, r, 3) for r in range(3)]
        assert tuple(sum(p[k] for p in parts) for k in range(3)) == (pairs, cells, nbytes)


def test_cluster_sets_host_matches_oracle(oracle):
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering, ClusteringOperation, Merge
    frames, offsets = synth.make_sequences(12, 16, 4, seed=5)
    d = oracle.align_all(frames, offsets, 1.0, workers=4)
    ops, roots, _ = oracle.clustering(d, 12, 0.4)
    mine = [ClusteringOperation(o["merge_i"], o["merge_j"], o["into"], o["distance"], Merge[o["operation"]]) for o in ops]
    assert AgglomerativeClustering.cluster_sets(mine, set(roots), 12) == oracle.cluster_sets(ops, roots, 12)
    assert AgglomerativeClustering.cluster_sets([], {0, 1, 2}, 3) == []      # singletons omitted (clustering.rs:71)


def test_autoencoder_bincode_layout(tmp_path):
    """`auto_encoder.bin` as bincode 1.x writes it (neural.rs:13-19,30-44): four Mat{flat: Vec<f32>, cols: usize}."""
    import struct
    from audio_pattern_discovery_amd.neural import AutoEncoder, Mat

    def mat(rows, cols, base):
        vals = [base + 0.5 * i for i in range(rows * cols)]
        return struct.pack("<Q", len(vals)) + struct.pack("<%df" % len(vals), *vals) + struct.pack("<Q", cols), vals

    blob, want = b"", []
    for rows, cols, base in ((3, 2, 1.0), (2, 3, -4.0), (1, 2, 10.0), (1, 3, 20.0)):   # w_encode, w_decode, b_encode, b_decode
        b, v = mat(rows, cols, base)
        blob += b
        want.append((v, cols))
    nn = AutoEncoder.from_bytes(blob)
    assert nn.w_encode.flat.tolist() == want[0][0] and nn.w_encode.cols == 2 and nn.w_encode.rows() == 3
    assert nn.w_decode.flat.tolist() == want[1][0] and nn.w_decode.cols == 3
    assert nn.b_encode.flat.tolist() == want[2][0] and nn.n_latent() == 2
    assert nn.b_decode.flat.tolist() == want[3][0]
    assert nn.to_bytes() == blob                                       # byte-exact round trip
    nn.save_file(str(tmp_path / "auto_encoder.bin"))
    assert AutoEncoder.from_file(str(tmp_path / "auto_encoder.bin")).to_bytes() == blob
    for bad in (blob[:-1], blob + b"\\0", blob[:5]):
        with pytest.raises(ValueError):
            AutoEncoder.from_bytes(bad)


def test_dendrogram_bracket_strings():
    """reporting.rs:143-168: qtree strings by Merge kind."""
    from audio_pattern_discovery_amd.clustering import ClusteringOperation, Merge, dendrograms
    ops = [ClusteringOperation(0, 1, 4, 0.1, Merge.Sequence2Sequence), ClusteringOperation(2, 4, 5, 0.2, Merge.Sequence2Cluster),
           ClusteringOperation(5, 3, 6, 0.3, Merge.Cluster2Sequence)]
    out = dendrograms(ops, {6}, ["A", "B", "C", "D"])
    assert out == {6: "[.6 [[.5 [C [.4 [A B ] ] ] ] D ] ]"}
    assert dendrograms(ops[:1], {2, 3, 4}, ["A", "B", "C", "D"]) == {4: "[.4 [A B ] ]"}


def test_dendrograms_equal_a_literal_replay_of_the_reference_loop(oracle, apd):
    """apd_dendrograms expands each root on demand; the reference builds every intermediate string in a HashMap
    (reporting.rs:142-169).  Both must give the same characters on real merge lists (oracle clustering of random batches)."""
    from audio_pattern_discovery_amd.clustering import ClusteringOperation, Merge, dendrograms
    for n, perc, seed in ((9, 0.3, 1), (30, 0.9, 2), (64, 0.6, 3), (5, 0.99, 4)):
        d = synth.make_distance_matrix(n, "points" if seed % 2 else "ties", seed=seed)
        ops, roots, _ = oracle.clustering(d, n, perc)
        labels = ["{img%d}\n" % i for i in range(n)]
        results = {}
        for o in ops:                                                   # the reference's loop, literally
            i, j, k = o["merge_i"], o["merge_j"], o["into"]
            left = results[i] if o["operation"] in ("Cluster2Sequence", "Cluster2Cluster") else labels[i]
            right = results[j] if o["operation"] in ("Sequence2Cluster", "Cluster2Cluster") else labels[j]
            results[k] = "[.%d [%s %s ] ]" % (k, left, right)
        want = {r: results[r] for r in roots if r in results}
        mine = [ClusteringOperation(o["merge_i"], o["merge_j"], o["into"], o["distance"], Merge[o["operation"]]) for o in ops]
        assert dendrograms(mine, set(roots), labels) == want
    # ids need not be unique or fresh: HashMap::insert overwrites, a string built earlier keeps its operands by value.  An op
    # whose `into` is one of its own operands, a repeated `into`, a huge id -- all finite, all equal to the literal replay.
    S2S, S2C, C2S, C2C = Merge.Sequence2Sequence, Merge.Sequence2Cluster, Merge.Cluster2Sequence, Merge.Cluster2Cluster
    odd = [[ClusteringOperation(0, 1, 5, 0.1, S2S), ClusteringOperation(5, 2, 5, 0.2, C2S)],
           [ClusteringOperation(0, 1, 5, 0.1, S2S), ClusteringOperation(2, 3, 5, 0.2, S2S), ClusteringOperation(5, 5, 6, 0.3, C2C)],
           [ClusteringOperation(0, 1, 4000000000, 0.1, S2S), ClusteringOperation(2, 4000000000, 7, 0.2, S2C), ClusteringOperation(7, 7, 7, 0.3, C2C),
            ClusteringOperation(7, 3, 4000000000, 0.3, C2S)]]
    labels = ["a", "b", "c", "d"]
    for ops in odd:
        results = {}
        for o in ops:
            left = results[o.merge_i] if o.operation in (C2S, C2C) else labels[o.merge_i]
            right = results[o.merge_j] if o.operation in (S2C, C2C) else labels[o.merge_j]
            results[o.into] = "[.%d [%s %s ] ]" % (o.into, left, right)
        assert dendrograms(ops, set(results), labels) == results
    assert dendrograms(odd[0], {5}, labels) == {5: "[.5 [[.5 [a b ] ] c ] ]"}
    # a degenerate list can describe an exponentially large string (every op doubles the last one): refused with OOM, not an abort
    doubling = [ClusteringOperation(0, 1, 4, 0.1, S2S)] + [ClusteringOperation(3 + t, 3 + t, 4 + t, 0.1, C2C) for t in range(1, 40)]
    with pytest.raises(apd.ApdError) as e:
        dendrograms(doubling, {43}, labels)
    assert e.value.status == apd.APD_ERR_OOM
    # an op that names a cluster no earlier op made: the reference's HashMap index panics
    with pytest.raises(apd.ApdError):
        dendrograms([ClusteringOperation(0, 7, 4, 0.1, Merge.Sequence2Cluster)], {4}, ["A", "B"])
    with pytest.raises(apd.ApdError):
        dendrograms([ClusteringOperation(0, 5, 4, 0.1, Merge.Sequence2Sequence)], {4}, ["A", "B"])     # a leaf without a label


def test_bench_gpus_n_starts_in_process():
    """`python bench.py --gpus 2` (how the driver starts N > 1) must reach the library's multi-device handle instead of asking
    for a launcher: without a GPU that means APD_ERR_NO_DEVICE from apd_multi_create, not a SystemExit about torchrun."""
    import subprocess
    import sys
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode != 0
    assert "torch.distributed.run" not in out.stderr and "no gfx950 HIP device" in out.stderr, out.stderr[-1500:]


def test_formats_survive_the_sanitizer_fuzz(tmp_path):
    """csrc/formats.hip is host-only: built as plain C++ under AddressSanitizer + UBSan (CPU build: the pool has no GPU sanitizer)
    and fed random / mutated op lists (repeated ids, self references), Discovery.toml texts and bincode images."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_formats")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-x", "c++",
                           os.path.join(root, "tools", "fuzz", "fuzz_formats.cpp"), "-o", exe])
    out = subprocess.run([exe, "20000", "99"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "no sanitizer report" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]


def test_rust_binding_declares_every_symbol_with_the_headers_arity():
    """bindings/rust/src/apd_sys.rs cannot be compiled here (no rustc), so it is at least kept in step with include/apd.h
    mechanically: the same set of functions, each with as many parameters as the C declaration."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "apd.h")).read(), flags=re.S)
    c_decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(apd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)}
    rust = open(os.path.join(ROOT, "bindings", "rust", "src", "apd_sys.rs")).read()
    rust = re.sub(r"//.*", "", rust)
    r_decl = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (apd_[a-z0-9_]+)\s*\(([^)]*)\)", rust, flags=re.S)}
    assert set(c_decl) == set(declared_symbols())
    assert set(r_decl) == set(c_decl), "apd_sys.rs and apd.h disagree: %s" % sorted(set(r_decl) ^ set(c_decl))

    def arity(params):
        params = params.strip()
        return 0 if params in ("", "void") else params.count(",") + 1

    for name in c_decl:
        assert arity(c_decl[name]) == arity(r_decl[name]), "%s: %d parameters in apd.h, %d in apd_sys.rs" % (name, arity(c_decl[name]), arity(r_decl[name]))


def test_rust_binding_modules_call_every_entry_point_with_the_headers_arity():
    """The drop-in modules of bindings/rust/src (alignments, clustering, spectrogram, neural -- uncompiled: no rustc here) are at
    least checked call by call: every `apd_*(...)` call passes as many arguments as include/apd.h declares, and the companions'
    modules exist with the reference's public items (spectrogram.rs:31,99,103,152,167,192; neural.rs:21,30,39,55)."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "apd.h")).read(), flags=re.S)
    c_decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(apd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)}

    def arity(params):
        params = params.strip()
        return 0 if params in ("", "void") else params.count(",") + 1

    def calls(text):
        text = re.sub(r"//.*", "", text)
        for m in re.finditer(r"\b(apd_[a-z0-9_]+)\s*\(", text):
            depth, i, args, start = 1, m.end(), 0, m.end()
            while depth and i < len(text):
                ch = text[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if ch == "," and depth == 1:
                    args += 1
                i += 1
            inner = text[start:i - 1].strip()
            yield m.group(1), (0 if not inner else args + 1)

    src = os.path.join(ROOT, "bindings", "rust", "src")
    seen = set()
    for name in ("alignments.rs", "clustering.rs", "spectrogram.rs", "neural.rs"):
        text = open(os.path.join(src, name)).read()
        for fn, n_args in calls(text):
            if fn in c_decl:
                seen.add(fn)
                assert n_args == arity(c_decl[fn]), "%s: %s called with %d arguments, apd.h declares %d" % (name, fn, n_args, arity(c_decl[fn]))
    assert {"apd_cepstrum", "apd_encode", "apd_interesting_ranges", "apd_autoencoder_parse", "apd_autoencoder_copy", "apd_autoencoder_serialize",
            "apd_multi_align_all", "apd_clustering", "apd_cluster_sets"} <= seen
    spec, neural = open(os.path.join(src, "spectrogram.rs")).read(), open(os.path.join(src, "neural.rs")).read()
    for item in ("pub fn new(fft_size: usize, fft_step: usize, filter_size: usize, raw_audio: &AudioData) -> NDSequence", "pub fn vec(&self, t: usize) -> &[f32]",
                 "pub fn encoded(&self, nn: &AutoEncoder) -> NDSequence", "pub fn len(&self) -> usize", "pub fn at(&self, t: usize, f: usize) -> f32",
                 "pub fn interesting_ranges(&self, moving_average: usize, perc: f32, min_len: usize) -> Vec<Slice>"):
        assert item in spec, item
    for item in ("pub fn n_latent(&self) -> usize", "pub fn from_file(file: &str) -> Result<AutoEncoder>", "pub fn save_file(&self, file: &str) -> Result<()>",
                 "pub fn predict(&self, x: &Mat) -> Mat"):
        assert item in neural, item
    assert "clustering.rs:175" in open(os.path.join(src, "clustering.rs")).read()      # merge's absence is stated, with the reason
