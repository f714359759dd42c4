#!/bin/bash
# config 4's first N blocks (default 24: 300 block pairs) through one `daligner -P` command under a list of environment
# settings: wall, phases and the library's host profile.   bash scripts/gpu_c4_env.sh [N] -- "VAR=val VAR2=val" "..." ...
P=$PWD
N=24
if [ "$1" != "--" ] && [ -n "$1" ]; then N=$1; shift; fi
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
W=$(mktemp -d /dev/shm/c4e.XXXX)
damar_amd/bin/simdb $W SIM 248 -c80 -m15000 -s3000 -e.15 -r4 -S78 -N$N > /dev/null || exit 1
python3 - "$W" $N <<'PY'
import sys
w, n = sys.argv[1], int(sys.argv[2])
open(w + "/plan.txt", "w").write("".join("daligner -k14 -j8 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))) for a in range(1, n + 1)))
open(w + "/keep.txt", "w").write("nothing-is-kept\n")
PY
: > gpurun_out/c4env.txt
for setting in "" "$@"; do
  for rep in 1 2; do
    ( cd $W && env $setting DAMAR_HOSTPROF=1 DAMAR_LAS_KEEP=$W/keep.txt DAMAR_PLAN_TIDY=1 DAMAR_PLAN_STATS=$W/stats.json \
      timeout -k 10 120 $P/damar_amd/bin/daligner -P plan.txt > /dev/null 2> $W/err.txt ) || { echo "[$setting] failed"; tail -3 $W/err.txt; continue; }
    python3 - "$setting" $W <<'PY' | tee -a gpurun_out/c4env.txt
import json, sys
st = json.load(open(sys.argv[2] + "/stats.json"))
print("[%s] wall %.0f ms  phases %s\n     host %s" % (sys.argv[1], st["wall_ms"], {k: round(v) for k, v in st["phase_ms"].items()}, {k: round(v) for k, v in st["host_wall_ms"].items()}))
for ln in open(sys.argv[2] + "/err.txt"):
    if ln.startswith("damar host") or ln.startswith("damar: host") or "scratch grow" in ln:
        print("     " + ln.strip()[:400])
PY
  done
done
rm -rf $W
