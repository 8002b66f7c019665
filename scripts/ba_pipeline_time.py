"""stand-alone timing of the whole local-BA keyframe job on B DISTINCT device-resident maps:
restore -> ov2_map_local_ba_setup_batch -> ov2_ba_solve_batch_dev -> ov2_map_local_ba_update_batch -> sync.
usage: python scripts/ba_pipeline_time.py [B] [reps]     (BA_KFS / BA_LMS / BA_SPREAD: nominal window and spread)"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import synth_ba


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    kf, lm = int(os.environ.get("BA_KFS", 50)), int(os.environ.get("BA_LMS", 10000))
    spread = float(os.environ.get("BA_SPREAD", 0.2))
    specs = synth_ba.sequence_window_specs(B, seed=20211, n_kf=kf, n_lm=lm, spread=spread)
    t = time.perf_counter()
    wins = synth_ba.make_windows_parallel(specs, min(32, len(os.sched_getaffinity(0))))
    print(f"{B} windows generated in {time.perf_counter() - t:.1f} s", flush=True)
    from ov2slam_amd import frontend as fe, local_ba, device_map as DM
    ctx = fe.Context(0)
    maps = [DM.DeviceMap.from_problem(ctx, P, isobs="newest") for P in wins]
    for m in maps:
        m.save_state()
    o = local_ba.default_options()
    best = None
    prof = getattr(ctx.lib, "ov2_debug_chol_prof", None) if os.environ.get("OV2_CHOL_PROF") else None   # -DOV2_CHOL_PROF builds only
    for rep in range(reps):
        ctx.synchronize()
        if prof is not None:
            buf = (C.c_ulonglong * 8)()
            prof(buf, 1)
        t0 = time.perf_counter()
        DM.restore_state_batch(ctx, maps)
        views = DM.setup_batch(ctx, maps, calib_l=wins[0].calib_l)
        t1 = time.perf_counter()
        pcs, rcs = DM.problems_of(views, wins[0], True)
        st = ctx.lib.ov2_ba_solve_batch_dev(ctx.h, B, pcs, C.byref(o), rcs)
        assert st == 0, ctx.lib.ov2_last_error(ctx.h)
        t2 = time.perf_counter()
        DM.update_batch(ctx, maps, views, cur_kfid=[m.newkf for m in maps], want_lists=False)
        ctx.synchronize()
        t3 = time.perf_counter()
        cur = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
        if prof is not None:
            prof(buf, 0)
            print("   Cholesky phases of window 0 (us): update %.0f, diagonal %.0f, substitution %.0f, write-back %.0f, backward %.0f"
                  % tuple(v / 100.0 for v in list(buf)[:5]), flush=True)
        if rep and (best is None or cur[0] < best[0]):
            best = cur
    n1 = [r.n_log_robust - 1 for r in rcs]
    n2 = [(r.n_log - r.n_log_robust - 1) if r.l2_done else 0 for r in rcs]
    its = sum(n1) + sum(n2)
    blocks = [int(v.n_res) for v in views]
    tot, ts, tv, tu = best
    print(f"B={B}: {1e3 * tot:.2f} ms per batch = set-up {1e3 * ts:.2f} + solve {1e3 * tv:.2f} + update {1e3 * tu:.2f}; "
          f"{1e3 * tot / B:.3f} ms per window, {B / tot:.0f} solves/s, {its / tot:.0f} LM it/s", flush=True)
    print(f"   residual blocks min/mean/max {min(blocks)}/{np.mean(blocks):.0f}/{max(blocks)}; robust iterations "
          f"{np.bincount(n1, minlength=6).tolist()}, L2 iterations {np.bincount(n2, minlength=11).tolist()}; slowest window "
          f"{max(a + b for a, b in zip(n1, n2))} iterations", flush=True)
    it_bytes = sum((a + b) * (nb * 256.0 + 2.0 * (6.0 * (v.n_pose - 1)) ** 2 * 8.0 + (7 * v.n_pose + v.n_lm) * 8.0)
                   for a, b, nb, v in zip(n1, n2, blocks, views))
    print(f"   algorithmic bytes (SURVEY 8d) {it_bytes / 1e9:.2f} GB per batch -> {it_bytes / tot / 1e9:.0f} GB/s = "
          f"{it_bytes / tot / 8e12:.3f} of the HBM roofline", flush=True)


if __name__ == "__main__":
    main()
