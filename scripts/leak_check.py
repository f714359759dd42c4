"""Peak RSS of the bench process for 2 and for 8 timed steps: a host-side leak shows as growth."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import resource, sys, os
sys.argv = ["bench.py", "--no-cpu", "--warmup", "1", "--steps", sys.argv[1]]
sys.path.insert(0, %r)
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
print(sys.argv[-1], "steps: peak RSS %%.0f MB" %% (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.))
''' % ROOT
for steps in ("2", "8"):
    subprocess.run([sys.executable, "-c", code, steps], check=True)
