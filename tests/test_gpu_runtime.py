"""Which HIP / RCCL runtime executes the library's calls.  libapd_hip.so is linked (RUNPATH) against /opt/rocm; the GPU suite
imports no torch, so that is the runtime every other test runs on.  The reverse order -- torch first, its bundled HIP and
RCCL mapped before the library loads -- is the interop case (apd_set_stream on torch's stream, torch tensors as device
buffers) and is exercised here in a CHILD process, so that one pytest process never holds both flavours."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_suite_runs_on_the_linked_runtime(apd):
    if "torch" in sys.modules:
        pytest.skip("torch was imported into this process by another test module")
    info = apd.runtime_info()
    assert "/opt/rocm" in info and "torch" not in info, info
    ctx = apd.Context(0)
    ctx.selftest()
    buf = ctx.alloc(1 << 20)
    buf.fill(0x3C)
    assert set(buf.to_numpy("u1").tolist()) == {0x3C}
    # a pointer the context did not hand out (or handed out and already released) is refused, not passed to hipFree
    import ctypes as C
    assert apd.lib().apd_device_free(ctx.handle, C.c_void_p(buf.ptr + 256)) == apd.APD_ERR_INVALID_ARG
    keep = ctx.alloc(4096)                                               # never freed by us: apd_destroy releases it
    ctx.close()
    keep.free()                                                          # after the context: a no-op in the mirror


CHILD = r"""
import ctypes as C, json, sys
import numpy as np
import torch                                   # FIRST: torch's bundled libamdhip64 / librccl are mapped before libapd_hip.so
sys.path.insert(0, %r)
from audio_pattern_discovery_amd import _lib, synth, sharding
from audio_pattern_discovery_amd.alignments import Batch
L = _lib.lib()
n = 60
frames, offsets = synth.make_sequences(n, 70, 13, seed=3)
cfg = _lib.AlignConfig(0.0625, 1.0, 1.0, 1.0)
ctx = _lib.Context(0, stream=torch.cuda.current_stream().cuda_stream)      # apd_set_stream: the caller's stream
d_frames = torch.from_numpy(frames).cuda()
batch = C.c_void_p()
_lib.check(L.apd_batch_create(ctx.handle, C.c_void_p(d_frames.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, 13, 1, C.byref(batch)), ctx.handle)
out = torch.empty(n * n, dtype=torch.float32, device="cuda")
_lib.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), C.c_void_p(out.data_ptr())), ctx.handle)
torch.cuda.synchronize()
got = out.cpu().numpy().reshape(n, n)
own = _lib.Context(0)                                                      # the library's own stream and buffers, same process
b2 = Batch(own, frames, offsets, 13)
ref = np.empty((n, n), np.float32)
_lib.check(L.apd_align_all(own.handle, b2.handle, C.byref(cfg), ref.ctypes.data_as(C.POINTER(C.c_float))), own.handle)
comm = sharding.Comm(own, sharding.Comm.unique_id(), 0, 1)                 # RCCL of whichever runtime was mapped first
print(json.dumps({"equal": bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32))), "runtime": _lib.runtime_info(),
                  "ranks": comm.count(), "finite": bool(np.isfinite(got).all())}))
"""


def test_torch_first_interop_in_a_child_process(oracle):
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["equal"] and d["finite"] and d["ranks"] == 1
    assert "torch" in d["runtime"], d["runtime"]                          # the bundled runtime really was the one in use there
