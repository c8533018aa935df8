"""Probe: what does pinning a caller's 5.2 MB wire buffer cost (hipHostRegister / hipHostUnregister), alone and from 16
threads at once, and how fast is a host-to-device copy of it pageable vs registered?"""
import ctypes
import threading
import time

import numpy as np

hip = ctypes.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
N = 5 * 32768 * 32
bufs = [np.random.randint(0, 255, N, dtype=np.uint8) for _ in range(64)]
d = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(N * 64))


def reg(b):
    return hip.hipHostRegister(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(N), 0)


def unreg(b):
    return hip.hipHostUnregister(ctypes.c_void_p(b.ctypes.data))


def copy_all():
    t0 = time.perf_counter()
    for i, b in enumerate(bufs):
        hip.hipMemcpyAsync(ctypes.c_void_p(d.value + i * N), ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(N), 1, None)
    hip.hipDeviceSynchronize()
    return time.perf_counter() - t0


copy_all()
print("pageable: 64 x 5.2 MB H2D %.1f ms" % (copy_all() * 1e3))
t0 = time.perf_counter()
rcs = [reg(b) for b in bufs]
t_reg = time.perf_counter() - t0
print("register serial: %.2f ms per buffer (rc %s)" % (t_reg / 64 * 1e3, set(rcs)))
copy_all()
print("registered: 64 x 5.2 MB H2D %.1f ms" % (copy_all() * 1e3))
t0 = time.perf_counter()
for b in bufs:
    unreg(b)
print("unregister serial: %.2f ms per buffer" % ((time.perf_counter() - t0) / 64 * 1e3))


def worker(chunk):
    for b in chunk:
        reg(b)


t0 = time.perf_counter()
ths = [threading.Thread(target=worker, args=(bufs[i::16],)) for i in range(16)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print("register from 16 threads: %.2f ms for all 64" % ((time.perf_counter() - t0) * 1e3))
for b in bufs:
    unreg(b)
