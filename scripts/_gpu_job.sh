cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_adaptive.py tests/test_parity_refine.py tests/test_capi_load.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import sys, time; sys.path.insert(0, "/root/repo")
import torch
from artemis_amd.driver import Simulation
for deck, ov in (("blast/blast_amr.in", []), ("blast/blast_amr.in", ["parthenon/mesh/nx1=512", "parthenon/mesh/nx2=512", "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16"])):
    s = Simulation("/root/repo/inputs/" + deck, ov)
    s.evolve(20); torch.cuda.synchronize()
    t = time.time(); r0 = s.remeshes; n = s.evolve(200); torch.cuda.synchronize(); w = time.time() - t
    print(deck, ov[:2], "blocks", s.nblocks, "zones", s.total_zones, "%.2f ms/cycle" % (1e3 * w / n), "remeshes", s.remeshes - r0, "zc/s %.3e" % (s.total_zones * n / w))
PY
