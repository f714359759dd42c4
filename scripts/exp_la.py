"""One bench step (report kernel alone: --no-overlap) with an experiment build of the library:
python3 scripts/exp_la.py build/exp_<name> [bench flags].  The parity check of the step is off: such builds are wrong on purpose."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import damar_amd.lib as dl
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libdir = sys.argv[1]
dl.lib_path = lambda: os.path.join(root, libdir, "libdamar_hip.so")
import bench
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-trace", "--no-e2e", "--no-legs"] + sys.argv[2:]
bench.main()
