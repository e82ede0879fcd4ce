"""For a node with several MI355X (none was available to the build: the pool hands out one GPU per call):
   python tools/multi_gpu_check.py [n_devices]
drives the persistent multi-device handle (apd_multi_*: ONE process, ncclCommInitAll over the first n_devices GPUs, pair tiles
sharded, one grouped all-gather) on a cfg 2 sized batch: the matrix must equal the one-device result bit for bit and the CPU
oracle to 1e-4 on sampled entries; prints the collective in use, the ranks RCCL saw and the time of the first and the second
alignment (setup is paid once).  Also runs the peer-copy fallback (APD_MULTI_COLLECTIVE=peer).  No torch.  The same path
under a timer is `python bench.py --gpus N`; the process-per-GPU form is `python bench.py --gpus N --launcher spawn`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_pattern_discovery_amd import _lib, sharding, synth
from oracle import binding as oracle

n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print(_lib.runtime_info())
frames, offsets = synth.make_sequences(1024, 512, 13, seed=2)
cfg = _lib.AlignConfig(0.0625, 1.0, 1.0, 1.0)
one = sharding.Multi([0])
ref = one.align_all(one.batch(offsets, 13, frames=frames), cfg)
one.close()
rng = np.random.default_rng(0)
pi = rng.integers(0, 1024, 400).astype(np.uint32)
pj = (pi + 1 + rng.integers(0, 1023, 400)).astype(np.uint32) % 1024
want, _ = oracle.align_sample(frames, offsets, pi, pj, 0.0625, workers=8)
ok = True
for mode in ("rccl", "peer"):
    if mode == "peer":
        os.environ["APD_MULTI_COLLECTIVE"] = "peer"
    t0 = time.perf_counter()
    m = sharding.Multi(list(range(n_dev)))
    os.environ.pop("APD_MULTI_COLLECTIVE", None)
    mb = m.batch(offsets, 13, frames=frames)
    first = m.align_all(mb, cfg)
    t1 = time.perf_counter()
    second = m.align_all(mb, cfg)
    t2 = time.perf_counter()
    rel = float(np.max(np.abs(second[pi, pj] - want) / np.abs(want)))
    same = bool(np.array_equal(ref.view(np.uint32), first.view(np.uint32)) and np.array_equal(ref.view(np.uint32), second.view(np.uint32)))
    print("devices %d [%s]: ranks seen %d; bitwise equal to one device: %s; max rel err vs oracle %.2e; first call %.3f s (with setup), second %.4f s"
          % (n_dev, m.collective()[:60], m.ranks_seen(), same, rel, t1 - t0, t2 - t1))
    ok = ok and same and rel <= 1e-4 and m.ranks_seen() == n_dev
    m.close()
sys.exit(0 if ok else 1)
