"""CPU issue time vs GPU time of one c3 step: is the host the bottleneck anywhere?
usage: python tools/issue_time.py [c2|c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tasks = ["depth"] if wl == "c2" else list(bench.ALL_TASKS)
    model, data, _ = bench.build_workload(tasks, 1 if wl == "c2" else 4, 64, dev)
    for _ in range(3):
        with torch.no_grad():
            model.forward(data, tasks)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        with torch.no_grad():
            model.forward(data, tasks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{wl}: host issue {1e3 * (t1 - t0) / K:.2f} ms/step, total {1e3 * (t2 - t0) / K:.2f} ms/step, "
          f"tail after last issue {1e3 * (t2 - t1):.1f} ms")
    # one isolated step: issue time with an empty queue
    t0 = time.perf_counter()
    with torch.no_grad():
        model.forward(data, tasks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"isolated step: issue {1e3 * (t1 - t0):.2f} ms, done {1e3 * (t2 - t0):.2f} ms")


if __name__ == "__main__":
    main()
