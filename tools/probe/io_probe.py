#!/usr/bin/env python3
"""Host-side probe: how fast can a batch of output frames go from a (pinned-like) buffer into a fresh region of an .npy dump?
pwrite (buffered) vs a shared mapping vs O_DIRECT, from 1 / 4 / 12 threads.  Prints GB/s.  usage: python tools/probe/io_probe.py [dir]"""
import mmap, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np

d = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
FRAME = 3840 * 1080 * 3
NB, BATCH = 6, 16
src = np.random.randint(0, 255, (BATCH, FRAME), dtype=np.uint8)
pool = ThreadPoolExecutor(16)


def run(name, T, writer, setup, teardown=None):
    path = os.path.join(d, f"io_probe_{name}.bin")
    if os.path.exists(path):
        os.unlink(path)
    st = setup(path)
    t0 = time.perf_counter()
    for b in range(NB):
        per = (BATCH + T - 1) // T
        futs = [pool.submit(writer, st, b, k, min(k + per, BATCH)) for k in range(0, BATCH, per)]
        for f in futs:
            f.result()
    dt = time.perf_counter() - t0
    if teardown:
        teardown(st)
    os.unlink(path)
    print(f"{name:10s} threads {T:2d}: {NB * BATCH * FRAME / dt / 1e9:6.2f} GB/s  ({NB * BATCH / dt:7.1f} frames/s of 12.4 MB)", flush=True)


def setup_fd(path, flags=0):
    fd = os.open(path, os.O_RDWR | os.O_CREAT | flags, 0o644)
    os.posix_fallocate(fd, 0, 4096 + NB * BATCH * FRAME)
    return fd


def w_pwrite(fd, b, k0, k1):
    mv = memoryview(src[k0:k1]).cast("B")
    off, done = 4096 + (b * BATCH + k0) * FRAME, 0
    while done < len(mv):
        done += os.pwritev(fd, [mv[done:]], off + done)


def setup_map(path):
    fd = setup_fd(path)
    m = np.memmap(path, dtype=np.uint8, mode="r+", offset=4096, shape=(NB * BATCH, FRAME))
    os.close(fd)
    return m


def w_map(m, b, k0, k1):
    np.copyto(m[b * BATCH + k0:b * BATCH + k1], src[k0:k1])


for T in (1, 4, 12):
    run("pwrite", T, w_pwrite, setup_fd, os.close)
for T in (1, 4, 12):
    run("mmap", T, w_map, setup_map)
try:
    # O_DIRECT: buffer, offset and length must be block aligned: use an aligned staging copy of the batch
    al = mmap.mmap(-1, BATCH * FRAME + 4096)
    asrc = np.frombuffer(al, dtype=np.uint8, count=BATCH * FRAME).reshape(BATCH, FRAME)
    asrc[...] = src

    def w_direct(fd, b, k0, k1):
        mv = memoryview(asrc[k0:k1]).cast("B")
        off, done = 4096 + (b * BATCH + k0) * FRAME, 0
        while done < len(mv):
            done += os.pwritev(fd, [mv[done:]], off + done)
    for T in (1, 4, 12):
        run("direct", T, w_direct, lambda p: setup_fd(p, os.O_DIRECT), os.close)
except OSError as e:
    print("O_DIRECT not usable here:", e)
os.system(f"df -T {d} | tail -1; nproc")
