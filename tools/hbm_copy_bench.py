#!/usr/bin/env python3
"""Device copy / read bandwidth of the box ("confirm the HBM peak with a device copy
benchmark").  Plain torch ops on one GPU: d2d copy (read + write), fill (write only), sum (read
only), sizes far beyond the 256 MB MALL.  Usage: python tools/hbm_copy_bench.py [GiB]
"""
import sys
import time

import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    n = int(gib * (1 << 30)) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    a.fill_(1.0)
    nbytes = n * 4
    t = timed(lambda: b.copy_(a))
    print(f"copy  {gib:.1f} GiB: {2 * nbytes / t / 1e12:.2f} TB/s (read + write)")
    t = timed(lambda: b.fill_(2.0))
    print(f"fill  {gib:.1f} GiB: {nbytes / t / 1e12:.2f} TB/s (write)")
    t = timed(lambda: a.sum())
    print(f"sum   {gib:.1f} GiB: {nbytes / t / 1e12:.2f} TB/s (read)")


if __name__ == "__main__":
    main()
