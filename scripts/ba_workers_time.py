"""stand-alone throughput of the native local-BA pipeline worker(s) (libov2host.so BaPipelineNative: restore -> set-up ->
solve -> update per keyframe job on DISTINCT device-resident maps), nothing else on the GPU.
usage: python scripts/ba_workers_time.py [seqs] [workers] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import synth_ba


def main():
    seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    kf, lm = int(os.environ.get("BA_KFS", 50)), int(os.environ.get("BA_LMS", 10000))
    specs = synth_ba.sequence_window_specs(seqs, seed=20211, n_kf=kf, n_lm=lm, spread=float(os.environ.get("BA_SPREAD", 0.2)))
    wins = synth_ba.make_windows_parallel(specs, min(32, len(os.sched_getaffinity(0))))
    from ov2slam_amd import frontend as fe, device_map as DM, host_map
    share = [seqs // workers + (1 if k < seqs % workers else 0) for k in range(workers)]
    ctxs, pipes, i = [], [], 0
    for k in range(workers):
        c = fe.Context(0)
        ms = [DM.DeviceMap.from_problem(c, P, isobs="newest") for P in wins[i:i + share[k]]]
        for m in ms:
            m.save_state()
        i += share[k]
        ctxs.append((c, ms))
        pipes.append(host_map.EstimatorPipeline(c, ms, wins[0], max_batch=int(os.environ.get("BA_BATCH", 64))))
    for p in pipes:
        p.submit_all()
    time.sleep(1.0)   # warm-up batch
    for p in pipes:
        p.set_counting(True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for p in pipes:
            p.submit_all()
        time.sleep(0.002)
    el = time.perf_counter() - t0
    for p in pipes:
        p.set_counting(False)
    tot = [p.stats() for p in pipes]
    solves, iters = sum(t["solves"] for t in tot), sum(t["iters"] for t in tot)
    nb = sum(t["batches"] for t in tot)
    print(f"seqs {seqs} workers {workers}: {solves / el:.0f} solves/s, {iters / el:.0f} LM it/s, {nb} batches in {el:.2f} s; per batch: "
          f"set-up {1e3 * sum(t['setup_s'] for t in tot) / max(nb, 1):.2f} ms, solve {1e3 * sum(t['solve_s'] for t in tot) / max(nb, 1):.2f}, "
          f"update {1e3 * sum(t['update_s'] for t in tot) / max(nb, 1):.2f}; status {[t['last_status'] for t in tot]}", flush=True)
    for p in pipes:
        p.close()


if __name__ == "__main__":
    main()
