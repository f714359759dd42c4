#!/usr/bin/env python3
"""BASELINE config 4 as a WHOLE on one GPU: the 19.8 Gbp database (255 blocks of 78 Mbp) is generated on the box, the whole
HPCdaligner plan -- n (n + 1) / 2 = 32 640 block pairs for n = 255 -- is run by ONE `daligner -P` command, and
the .las files of the 8 + 8 sampled block pairs of tests/golden/config4_ref_md5.txt are compared with the reference's md5s.
The other ~55 GB of .las go through the writers into /dev/null (DAMAR_LAS_KEEP, host/las.c): the compute rate and the output
rate are stated separately.  Usage (GPU box, repo root):  python3 scripts/gpu_c4_whole.py [max_blocks]  -> gpurun_out/c4_whole.json"""
import hashlib, json, os, shutil, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from damar_amd import api
import bench

def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for c in iter(lambda: f.read(1 << 24), b""):
            h.update(c)
    return h.hexdigest()

def main():
    maxb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    work = tempfile.mkdtemp(prefix="damar_c4w_", dir="/dev/shm")
    stop = threading.Event()
    t_start = time.time()
    def tick():
        while not stop.wait(45):
            print("[c4 whole] +%.0f s" % (time.time() - t_start), flush=True)
    threading.Thread(target=tick, daemon=True).start()
    try:
        t0 = time.time()
        nb = api.sim_write_db(work, "SIM", 248., coverage=80., seed=4, rmean=15000, rsdev=3000, block_mbp=78, max_blocks=maxb)
        t_gen = time.time() - t0
        print("[c4 whole] %d blocks generated in %.0f s" % (nb, t_gen), flush=True)
        run = os.path.join(work, "run")
        bench.link_db(work, "SIM", run)
        with open(os.path.join(run, "plan.txt"), "w") as f:
            f.write(bench.plan_text("SIM", nb, "-k14 -j8"))
        want = {}
        for ln in open(os.path.join(ROOT, "tests", "golden", "config4_ref_md5.txt")):
            if ln.startswith("#"):
                continue
            m, tag, a, b, rel = ln.split()
            if int(a) <= nb and int(b) <= nb:
                want[rel] = m
        with open(os.path.join(run, "keep.txt"), "w") as f:
            f.write("".join(rel + "\n" for rel in sorted(want)))
        env = dict(os.environ, DAMAR_LAS_KEEP=os.path.join(run, "keep.txt"), DAMAR_PLAN_STATS=os.path.join(run, "stats.json"),
                   DAMAR_PLAN_TIDY="1", DAMAR_CLIPROF="1")
        t0 = time.time()
        with open(os.path.join(ROOT, "gpurun_out", "c4_whole.err"), "w") as err:
            r = subprocess.run([api.daligner_binary(), "-P", "plan.txt"], cwd=run, env=env, stdout=subprocess.DEVNULL, stderr=err)
        wall = time.time() - t0
        if r.returncode != 0:
            print("daligner -P failed: rc", r.returncode)
            print(open(os.path.join(ROOT, "gpurun_out", "c4_whole.err")).read()[-3000:])
            sys.exit(1)
        st = json.loads(open(os.path.join(run, "stats.json")).read())
        bad = [rel for rel, m in sorted(want.items()) if not os.path.exists(os.path.join(run, rel)) or md5(os.path.join(run, rel)) != m]
        npairs = nb * (nb + 1) // 2
        bp = st["aligned_bp"]
        res = {"workload": "config 4 WHOLE: simulator 248 -c80 -m15000 -s3000 -e.15 -r4, DBsplit -s78 -> %d blocks; the whole HPCdaligner plan "
                           "(%d block pairs x 2 orientations) by one cold `daligner -P plan` command on ONE MI355X, daligner -k14 -j8" % (nb, npairs),
               "blocks": nb, "block_pairs": npairs, "block_pairs_run": st["block_pairs"],
               "wall_s": wall, "ms_per_block_pair": 1e3 * wall / npairs, "aligned_bp": bp, "value": bp / wall, "unit": "aligned bp/s",
               "records": st["records"], "seed_pairs": st["seed_pairs"], "local_alignments": st["local_alignments"],
               "index_builds": st["index_builds"], "block_loads": st["block_loads"], "tile": st["tile"], "bases_resident": st["bases_resident"],
               "budget_gb": st["budget_gb"], "phase_ms": st["phase_ms"], "host_wall_ms": st["host_wall_ms"],
               "output": {"las_bytes": st["las_bytes"], "las_files": st["las_files"], "las_GB_per_s_at_this_rate": st["las_bytes"] / wall / 1e9,
                          "what": "bytes the writers assembled and handed to write(): the %d sampled files to tmpfs, the rest to /dev/null "
                                  "(DAMAR_LAS_KEEP); the compute wall above therefore excludes the file system, not the record assembly" % len(want)},
               "parity": {"files": len(want), "identical": not bad, "differing": bad[:5],
                          "against": "md5 of the reference daligner's files (tests/golden/config4_ref_md5.txt, samples 'lead' and 'full')"},
               "db_generation_s": t_gen, "plan_stats": st}
        line = json.dumps(res)
        open(os.path.join(ROOT, "gpurun_out", "c4_whole.json"), "w").write(line + "\n")
        print(line[:1500])
        sys.exit(0 if not bad else 2)
    finally:
        stop.set()
        shutil.rmtree(work, ignore_errors=True)

if __name__ == "__main__":
    main()
