"""How often the wave loop of report2_kernel (kernels/report_packed.h, duo_loop) enters its conditional blocks: one bench step
of config 2 with the counting build (scripts/build_prof_var.sh loopc -DDAMAR_LOOPC -> build/prof_loopc/).
   python3 scripts/loop_blocks.py [bench flags]        (the profiles/rNN_loop_blocks.txt table)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import damar_amd.lib as dl
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dl.lib_path = lambda: os.path.join(root, "build", "prof_loopc", "libdamar_hip.so")
import bench
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-trace", "--no-e2e", "--no-legs"] + sys.argv[1:]
bench.main()
L = dl.load()
out = (ctypes.c_ulonglong * 16)()
L.damar_loopc_read(out)
rows = [(0, "iterations (both halves of a wavefront in one)"),
        (12, "some lane passes the old best (the prefix maximum and the record breakers are computed in EVERY iteration)"),
        (6, "  some record breaker has popcount >= ave: trim test (table look-ups)"),
        (7, "    the trim test holds for some lane: per-lane trim slot written to LDS"),
        (4, "pebbles: some lane crosses a trace mark (one combined round for the A and the B chain)"),
        (5, "  something left after the combined round: the two leftover loops"),
        (10, "    A-side leftover loop iterations"),
        (11, "    B-side leftover loop iterations"),
        (1, "band moved to the middle of its lanes (6 ds_bpermute, lane constants recomputed)"),
        (3, "window continuations, counted per LANE: all 16 bases equal and more than 16 left (a second pair of loads)"),
        (2, "byte path: some lane's window would leave a read"),
        (8, "a sequence end reached: ends / clip block"),
        (9, "lasta reduced over the lanes (the lazy bound failed or the band is empty)")]
n = float(out[0]) if out[0] else 1.
print("%-112s %14s %9s" % ("block", "entered", "per iter"))
for i, what in rows:
    print("%-112s %14d %9.4f" % (what, out[i], out[i] / n))
