import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.getcwd())
from damar_amd import api
base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
work = tempfile.mkdtemp(prefix="tanprof_", dir=base)
nb = api.sim_write_db(work, "SIM", 27., coverage=20., seed=2, block_mbp=135, tandem_frac=0.3)
exe = os.path.join(os.getcwd(), "damar_amd", "bin", "datander")
blocks = ["SIM.%d" % i for i in range(1, nb + 1)]
for env in ({}, {"DAMAR_HOSTPROF": "1"}, {"DAMAR_PLAN_TIDY": "1"}, {}):
    shutil.rmtree(os.path.join(work, "tan"), ignore_errors=True)
    e = dict(os.environ); e.update(env)
    t0 = time.time()
    r = subprocess.run([exe, "-j16"] + blocks, cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(env, "wall %.3f" % (time.time() - t0), "rc", r.returncode)
    if env.get("DAMAR_HOSTPROF"): print(r.stderr[-3000:])
    time.sleep(0.5)
# strace-ish: time of a trivial HIP init
t0 = time.time(); subprocess.run([os.path.join(os.getcwd(), "damar_amd", "bin", "roofcal"), "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print("roofcal 1 (hip init + tiny kernels) %.3f" % (time.time() - t0))
shutil.rmtree(work)
