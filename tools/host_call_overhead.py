"""Host cost of the pieces of one C-ABI call from ops.py (GPU box: the stream query needs the runtime): python tools/host_call_overhead.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import _lib, ops

dev = torch.device("cuda:0")
t = torch.empty(1024, device=dev)
lib = _lib.load()
N = 20000


def bench(name, fn):
    for _ in range(1000):
        fn()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    print(f"{name:58s} {(time.perf_counter() - t0) / N * 1e6:6.2f} us")


bench("ops._stream()", ops._stream)
bench("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
bench("torch._C._cuda_getCurrentRawStream(0)", lambda: torch._C._cuda_getCurrentRawStream(0))
bench("ops._ptr(t)", lambda: ops._ptr(t))
bench("t.data_ptr()", t.data_ptr)
bench("_lib.call(host-only entry, 1 arg)", lambda: _lib.call("seg3d_debug_set_wgrad_lds", -1))
fn = lib.seg3d_debug_set_wgrad_lds
bench("bound ctypes function, 1 arg", lambda: fn(-1))
f11 = lib.seg3d_linear_fwd_x6
args_obj = (ops._ptr(t), 0, ops._ptr(t), None, None, None, 0, 128, 256, ops._ptr(t), ops._stream())
bench("ctypes call, 11 args prebuilt c_void_p (m = 0: returns at once)", lambda: f11(*args_obj))
p = t.data_ptr()
args_int = (p, 0, p, None, None, None, 0, 128, 256, p, torch._C._cuda_getCurrentRawStream(0))
bench("ctypes call, 11 args as python ints", lambda: f11(*args_int))
bench("full today: _lib.call + 4 _ptr + _stream, 11 args", lambda: _lib.call("seg3d_linear_fwd_x6", ops._ptr(t), 0, ops._ptr(t), None, None, None, 0, 128, 256, ops._ptr(t), ops._stream()))
bench("lean: bound fn + data_ptr ints + raw stream", lambda: f11(t.data_ptr(), 0, t.data_ptr(), None, None, None, 0, 128, 256, t.data_ptr(), torch._C._cuda_getCurrentRawStream(0)))
bench("torch.empty((1000, 192), device)", lambda: torch.empty((1000, 192), dtype=torch.float32, device=dev))
