"""Unequal penalties under a full band (the README's weighted INSERTION / DELETION / MATCH): which kernels run, how fast,
and bit-identity with the oracle on sampled pairs (run on the GPU box)."""
import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import synth, _lib
from audio_pattern_discovery_amd.alignments import align_work
from oracle import binding as oracle

n, length, dim = 256, 600, 13
pens = (1.0, 0.8, 0.5)
frames, offsets = synth.make_sequences(n, length, dim, seed=9)
ctx = _lib.Context(0); L = _lib.lib()
d_frames = torch.from_numpy(frames).cuda()
batch = C.c_void_p()
_lib.check(L.apd_batch_create(ctx.handle, C.c_void_p(d_frames.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, dim, 1, C.byref(batch)), ctx.handle)
cfg = _lib.AlignConfig(1.0, *pens)
d_out = torch.empty(n * n, dtype=torch.float32, device="cuda")
ctx.set_timing(True)
for _ in range(2):
    _lib.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), C.c_void_p(d_out.data_ptr())), ctx.handle)
    torch.cuda.synchronize()
ms = ctx.last_kernel_ms()
_, cells, _ = align_work(offsets, dim, cfg, 0, 1)
print("kernel ms %.2f  cell-updates/s %.3e" % (ms, cells / (ms * 1e-3)))
rng = np.random.default_rng(1)
pi = rng.integers(0, n, 48).astype(np.uint32); pj = ((pi + 1 + rng.integers(0, n - 1, 48)) % n).astype(np.uint32)
want, _ = oracle.align_sample(frames, offsets, pi, pj, 1.0, *pens, workers=16)
got = d_out.view(n, n)[torch.from_numpy(pi.astype(np.int64)), torch.from_numpy(pj.astype(np.int64))].cpu().numpy()
print("bit-identical:", bool(np.array_equal(got, want)), "max rel", float(np.max(np.abs(got - want) / np.abs(want))))
