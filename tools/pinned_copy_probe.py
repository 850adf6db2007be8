#!/usr/bin/env python3
"""D2H / H2D copy rates into pinned host memory by how it was allocated (hipHostMalloc flags, allocating thread), and how long the hipMemcpyAsync CALL takes
(a call that blocks for the length of the copy went through the runtime's staging path).  Run on the GPU box:  python tools/pinned_copy_probe.py"""
import ctypes as C
import threading
import time

hip = C.CDLL("libamdhip64.so")
N = 128 << 20


def chk(e, what):
    if e != 0:
        raise SystemExit("%s: hip error %d" % (what, e))


chk(hip.hipSetDevice(0), "hipSetDevice")
dev = C.c_void_p()
chk(hip.hipMalloc(C.byref(dev), C.c_size_t(N)), "hipMalloc")
streams = []
for _ in range(2):
    s = C.c_void_p()
    chk(hip.hipStreamCreateWithFlags(C.byref(s), 1), "hipStreamCreateWithFlags")      # non-blocking
    streams.append(s)


def host_alloc(flags, result):
    p = C.c_void_p()
    t0 = time.perf_counter()
    chk(hip.hipHostMalloc(C.byref(p), C.c_size_t(N), C.c_uint(flags)), "hipHostMalloc")
    result.append((p, time.perf_counter() - t0))


def timed_copy(dst, src, kind, stream):
    best_call, best_all = 1e9, 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        chk(hip.hipMemcpyAsync(dst, src, C.c_size_t(N), C.c_int(kind), stream), "hipMemcpyAsync")
        t1 = time.perf_counter()
        chk(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        t2 = time.perf_counter()
        best_call, best_all = min(best_call, t1 - t0), min(best_all, t2 - t0)
    return best_call, best_all


print("allocation                                   pin ms   D2H call ms  D2H total ms  GB/s   H2D total ms  GB/s")
for name, flags, other_thread in (("hipHostMallocDefault", 0, False), ("hipHostMallocDefault, another thread", 0, True), ("hipHostMallocPortable", 1, False),
                                  ("hipHostMallocMapped", 2, False), ("Portable | Coherent", 1 | 0x40000000, False), ("NonCoherent", 0x80000000, False)):
    res = []
    if other_thread:
        t = threading.Thread(target=host_alloc, args=(flags, res))
        t.start()
        t.join()
    else:
        host_alloc(flags, res)
    p, pin_s = res[0]
    c, a = timed_copy(p, dev, 2, streams[0])
    _, h = timed_copy(dev, p, 1, streams[1])
    print("%-42s %8.1f %12.2f %13.2f %6.1f %13.2f %6.1f" % (name, pin_s * 1e3, c * 1e3, a * 1e3, N / a / 1e9, h * 1e3, N / h / 1e9))
    chk(hip.hipHostFree(p), "hipHostFree")
import numpy as np
buf = np.empty(N, dtype=np.uint8)
c, a = timed_copy(C.c_void_p(buf.ctypes.data), dev, 2, streams[0])
print("%-42s %8s %12.2f %13.2f %6.1f" % ("pageable (numpy)", "-", c * 1e3, a * 1e3, N / a / 1e9))
