"""bench.py against an alternative build of the library: python scripts/bench_lib.py <libdir> [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import damar_amd.lib as dl
libdir = os.path.abspath(sys.argv[1])
dl.lib_path = lambda: os.path.join(libdir, "libdamar_hip.so")
import bench
sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()
