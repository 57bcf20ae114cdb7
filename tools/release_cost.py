#!/usr/bin/env python3
"""Scratch: does a process that has just released tens of GB of device memory slow the NEXT process's allocations?
    python tools/release_cost.py [gb=50] [rounds=4]
Each child allocates <gb> GB in one hipMalloc (torch), fills it, reports how long malloc and fill took, and exits. Children
run back to back, then with a pause before them, alternately."""
import subprocess
import sys
import time

CHILD = r'''
import sys, time, torch
gb = float(sys.argv[1])
t0 = time.perf_counter(); torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
x = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t2 = time.perf_counter()
x.fill_(1); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"init {t1 - t0:.3f} s, malloc {t2 - t1:.3f} s, fill {t3 - t2:.3f} s", flush=True)
import os; os._exit(0)
'''


def child(gb):
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-c", CHILD, str(gb)], capture_output=True, text=True)
    return f"{out.stdout.strip()} (process {time.perf_counter() - t0:.2f} s)"


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 50
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    print("first:", child(gb), flush=True)
    for r in range(rounds):
        print("back to back:", child(gb), flush=True)
        time.sleep(4)
        print("after 4 s:   ", child(gb), flush=True)
    print("small after big, back to back:", child(1), flush=True)
    print("small after small:", child(1), flush=True)


if __name__ == "__main__":
    main()
