"""For a node with several MI355X (none was available to the build: the pool hands out one GPU per call):
   python tools/multi_gpu_check.py [n_devices]
runs apd_align_all_multi -- ONE process, ncclCommInitAll over the first n_devices GPUs, pair tiles sharded, one all-gather --
on a cfg 2 sized batch and checks the matrix bit for bit against the one-device result and to 1e-4 against the CPU oracle
on sampled entries; prints the ranks RCCL saw and the wall time of each.  The process-per-GPU form is exercised by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (its JSON line reports config.collective / ranks_seen)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401  (loads the HIP runtime first)
from audio_pattern_discovery_amd import _lib, sharding, synth
from oracle import binding as oracle

n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else torch.cuda.device_count()
frames, offsets = synth.make_sequences(1024, 512, 13, seed=2)
cfg = _lib.AlignConfig(0.0625, 1.0, 1.0, 1.0)
t0 = time.perf_counter()
one, seen1 = sharding.align_all_multi([0], frames, offsets, 13, cfg)
t1 = time.perf_counter()
many, seen = sharding.align_all_multi(list(range(n_dev)), frames, offsets, 13, cfg)
t2 = time.perf_counter()
rng = np.random.default_rng(0)
pi = rng.integers(0, 1024, 400).astype(np.uint32)
pj = (pi + 1 + rng.integers(0, 1023, 400)).astype(np.uint32) % 1024
want, _ = oracle.align_sample(frames, offsets, pi, pj, 0.0625, workers=8)
rel = float(np.max(np.abs(many[pi, pj] - want) / np.abs(want)))
print("devices %d: ranks seen %d (1 device: %d); bitwise equal to one device: %s; max rel err vs oracle %.2e; %.3f s vs %.3f s incl. setup"
      % (n_dev, seen, seen1, bool(np.array_equal(one.view(np.uint32), many.view(np.uint32))), rel, t2 - t1, t1 - t0))
sys.exit(0 if (seen == n_dev and np.array_equal(one.view(np.uint32), many.view(np.uint32)) and rel <= 1e-4) else 1)
