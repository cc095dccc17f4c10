"""Why does the training process's host work slow down next to a running DataLoader?  Times a fixed single-thread Python + torch
launch loop (a) alone, (b) next to N busy worker PROCESSES (cgroup quota / scheduler), (c) next to a THREAD that copies 32 MB
tensors into pinned memory (the pin thread's work: GIL released inside the copy), (d) next to a thread running pure Python (GIL).
    python tools/host_contention_probe.py"""
import multiprocessing as mp
import os
import threading
import time

import torch


def burn(stop):
    x = 0
    while not stop.is_set():
        for i in range(100000):
            x += i * i


def main_loop(n=3000):
    a = torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        a.add_(1.0)                    # one tiny launch + the Python around it
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t / n * 1e6


def main():
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
        try:
            print(f, open(f).read().strip().replace("\n", " | "))
        except OSError:
            pass
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
    print("alone: %.1f us per launch" % main_loop())
    for nproc in (4, 16, 64):
        stop = mp.Event()
        ps = [mp.Process(target=burn, args=(stop,)) for _ in range(nproc)]
        for p in ps:
            p.start()
        time.sleep(1.0)
        print("next to %d busy processes: %.1f us per launch" % (nproc, main_loop()))
        stop.set()
        for p in ps:
            p.join()
    stop = threading.Event()

    def pinner():
        src = torch.empty(32 << 20, dtype=torch.uint8)
        while not stop.is_set():
            src.pin_memory()
    th = threading.Thread(target=pinner)
    th.start()
    time.sleep(0.5)
    print("next to a thread pinning 32 MB tensors in a loop: %.1f us per launch" % main_loop())
    stop.set()
    th.join()
    stop = threading.Event()

    def pinner_reuse():
        src = torch.empty(32 << 20, dtype=torch.uint8)
        dst = torch.empty(32 << 20, dtype=torch.uint8).pin_memory()
        while not stop.is_set():
            dst.copy_(src)
    th = threading.Thread(target=pinner_reuse)
    th.start()
    time.sleep(0.5)
    print("next to a thread copying 32 MB into ONE pinned buffer in a loop: %.1f us per launch" % main_loop())
    stop.set()
    th.join()
    stop = threading.Event()
    th = threading.Thread(target=burn, args=(stop,))
    th.start()
    time.sleep(0.5)
    print("next to a pure-Python thread: %.1f us per launch" % main_loop())
    stop.set()
    th.join()
    print("cpu.stat after:", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | ") if os.path.exists("/sys/fs/cgroup/cpu.stat") else "-")


if __name__ == "__main__":
    main()
