"""Does RCCL accept a communicator whose two ranks drive the SAME device?  (It does not -- "Duplicate GPU detected" -- which is
why the two-process tests on a one-GPU box run the gather protocol on the hostlink transport.)  Run on the GPU box:
python tools/rccl_dup_probe.py -> one line per rank."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    d = tempfile.mkdtemp()
    ps = [subprocess.Popen([sys.executable, __file__, str(r), d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(ps):
        try:
            out = p.communicate(timeout=90)[0]
        except subprocess.TimeoutExpired:
            p.kill()
            out = p.communicate()[0] + "\n[timed out]"
        print("rank %d: %s" % (r, " | ".join(x for x in out.strip().splitlines()[-6:])))
    sys.exit(0)

rank, d = int(sys.argv[1]), sys.argv[2]
from delly_amd import refine
ctx = refine.Context(device=0)
idf = os.path.join(d, "id")
if rank == 0:
    uid = refine.comm_unique_id()
    with open(idf + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idf + ".tmp", idf)
else:
    while not os.path.exists(idf):
        time.sleep(0.05)
    uid = open(idf, "rb").read()
try:
    comm = refine.Comm(ctx, rank, 2, uid)
    print("RCCL communicator of two ranks on one device: created (%r)" % (comm.info(),))
    comm.close()
except refine.DellyHipError as e:
    print("RCCL refused: %s" % e)
