"""Band-width / time-split profile of the report kernel (library built with -DDAMAR_PROF into
build/prof/).  Prints the counters of one bench step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import damar_amd.lib as dl
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dl.lib_path = lambda: os.path.join(root, "build", os.environ.get("DAMAR_PROF_DIR", "prof"), "libdamar_hip.so")
import bench
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-trace", "--no-e2e", "--no-legs"] + sys.argv[1:]
bench.main()
L = dl.load()
out = (ctypes.c_ulonglong * 32)()
L.damar_prof_read(out, 0)
names = ["deal_lanes_used", "deal_slots_done", "steps_in_passes_fit16", "steps_before_over16", "steps_in_passes_fit32",
         "steps_before_over32", "clk_firstLA", "clk_laterLA", "clk_pair", "n_firstLA", "n_laterLA",
         "passes_fit16", "passes_fit32", "clk_wave_mem", "clk_finish", "clk_wave_reg",
         "pairs", "seeds_scanned", "panels", "clk_to_scan_end", "clk_pass1", "clk_pass2_incl_LA", "clk_pass3",
         "clk_waves_busy_sum", "clk_wave_max(last launch max)", "waves",
         "deals", "deal_slots_running", "pk_overflows", "pk_passes", "deal_slots_served", "deal_band_lanes"]
for n, v in zip(names, out):
    print("%-24s %d" % (n, v))
