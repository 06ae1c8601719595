"""Host-side cost of one inference step (cfg2: bench.py's workload): cProfile over ten forward_test calls, functions by own and by cumulative time.

    python tools/diag/infer_cprofile.py > gpurun_out/infer_cprofile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    w = bench.WORKLOADS["cfg2"]
    dev = torch.device("cuda", 0)
    det = bench.build_model(w).to(dev).eval()
    batch = bench.to_device(bench.synth_batch(w, 0), dev)
    with torch.no_grad():
        for _ in range(5):
            det(return_loss=False, **batch)
        torch.cuda.synchronize()
        # host time to QUEUE a step: the deferred form returns before the detections' copy has landed
        prof = cProfile.Profile()
        n = 10
        t0 = time.perf_counter()
        prof.enable()
        for _ in range(n):
            det(return_loss=False, **batch)
        prof.disable()
        t1 = time.perf_counter()
    print(f"# {n} steps in {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / n:.2f} ms per step, profiler overhead included)")
    for key in ("tottime", "cumtime"):
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats(key).print_stats(40)
        print(f"# by {key}")
        print("\n".join(l[:190] for l in buf.getvalue().splitlines() if l.strip()))


if __name__ == "__main__":
    main()
