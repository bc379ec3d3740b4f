"""Soak: N repeated c3 forwards (all heads, B=4, per-clip tracker streams on) must be bit-identical to the first.
usage: python tools/soak_c3.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tasks = list(bench.ALL_TASKS)
    model, data, _ = bench.build_workload(tasks, 4, 64, dev)
    with torch.no_grad():
        ref = {k: v.clone() for k, v in model.forward(data, tasks).items() if torch.is_tensor(v)}
        bad = 0
        for i in range(iters):
            out = model.forward(data, tasks)
            torch.cuda.synchronize()
            diff = [k for k, v in ref.items() if not torch.equal(out[k], v)]
            if diff:
                bad += 1
                print(f"iter {i}: differs in {diff}")
    print(f"soak: {bad} of {iters} forwards differ from the first ({len(ref)} output tensors compared bitwise)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
