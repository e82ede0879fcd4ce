"""Debug: very many short sequences (grid-size limits) -- GPU sample vs oracle."""
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import _lib
from audio_pattern_discovery_amd.alignments import Batch
from oracle import binding as oracle

n = int(sys.argv[1]); ln = int(sys.argv[2]); dim = 13
rng = np.random.default_rng(0)
lens = rng.integers(ln - 2, ln + 3, n)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
frames = rng.standard_normal((int(offsets[-1]), dim)).astype(np.float32)
ctx = _lib.Context(0)
L = _lib.lib()
d_frames = torch.from_numpy(frames).cuda()
batch = C.c_void_p()
_lib.check(L.apd_batch_create(ctx.handle, C.c_void_p(d_frames.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, dim, 1, C.byref(batch)), ctx.handle)
pct = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0625
cfg = _lib.AlignConfig(pct, 1.0, 1.0, 1.0)
d_out = torch.empty(n * n, dtype=torch.float32, device="cuda")
ctx.set_timing(True)
_lib.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), C.c_void_p(d_out.data_ptr())), ctx.handle)
torch.cuda.synchronize()
print("kernel ms", ctx.last_kernel_ms())
pi = rng.integers(0, n, 64).astype(np.uint32); pj = ((pi + 1 + rng.integers(0, n - 1, 64)) % n).astype(np.uint32)
pi = np.concatenate([pi, [0, n - 1, n - 2, 5]]).astype(np.uint32); pj = np.concatenate([pj, [n - 1, 0, n - 1, n - 7]]).astype(np.uint32)
want, _ = oracle.align_sample(frames, offsets, pi, pj, pct, workers=8)
got = d_out.view(n, n)[torch.from_numpy(pi.astype(np.int64)), torch.from_numpy(pj.astype(np.int64))].cpu().numpy()
print("max rel", np.max(np.abs(got - want) / np.abs(want)), "zeros", int((got == 0).sum()))
