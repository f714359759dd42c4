#!/bin/bash
# where a cold GPU command's time goes: tools/startup.hip (bare runtime bring-up and teardown in three exit modes), then
# the contract command with its timeline (DAMAR_CLIPROF=1).   gpurun -- bash scripts/gpu_startup.sh
cd "$(dirname "$0")/.."
python3 - <<'PY'
import subprocess, time, os
exe = "damar_amd/bin/startup"
for mode, what in ((0, "free everything, return"), (1, "return with everything allocated"), (2, "_exit(0) with everything allocated")):
    for env in ({}, {"GPU_MAX_HW_QUEUES": "2"}):
        for rep in range(2):
            t0 = time.time()
            out = subprocess.run([exe, "512", str(mode)], stdout=subprocess.PIPE, text=True, env=dict(os.environ, **env)).stdout
            wall = (time.time() - t0) * 1e3
            if rep == 1:
                print("== mode %d (%s) %s" % (mode, what, env))
                print(out, end="")
                print("%8.1f ms  process gone" % wall)
            time.sleep(0.7)
PY
python3 scripts/tidyprof.py
