"""Adaptive mesh refinement: the host logic (tagging, tree update, hand-over of the conserved state, the rebuilt
block-graph exchange) on the CPU test double, 1 rank.  The GPU tests are tests/test_adaptive.py."""
import os

import numpy as np

from test_multilevel import _run_workers

LINWAVE = dict(deck=["linwave", "linear_wave_amr.in"], cycles=45,
               overrides=["problem/nperiod=1", "parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=8",
                          "parthenon/meshblock/nx2=8"])


def test_linear_wave_amr_on_cpu_double_conserves_across_remeshes(tmp_path):
    """linear_wave_amr.in at half resolution, 45 cycles: the refined band follows the crest (blocks are created and,
    after derefine_count cycles, merged), the leaves tile the root mesh, and mass / momentum / energy after the run
    equal those of the initial (adaptively refined) state to round-off."""
    a = _run_workers(1, LINWAVE, tmp_path, "amr")[0]
    i = _run_workers(1, dict(LINWAVE, cycles=0), tmp_path, "ini")[0]
    lv = a["meta"]["levels"]
    assert a["meta"]["remeshes"] >= i["meta"]["remeshes"] + 2 and set(lv) == {0, 1}
    assert sum(4.0 ** (-l) for l in lv) == 32.0 == sum(4.0 ** (-l) for l in i["meta"]["levels"])
    assert a["meta"]["ncycle"] == 45 and i["meta"]["ncycle"] == 0
    scale = np.abs(i["hist"]).max()
    assert np.allclose(a["hist"][:5], i["hist"][:5], rtol=0, atol=2e-13 * scale), a["hist"][:5] - i["hist"][:5]
    for _, prim in a["blocks"]:
        assert np.isfinite(prim).all()
