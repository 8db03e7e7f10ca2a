"""INTEGRATION.md's adapter is a real file (integration/artemis_hip_adapter.hpp): check that it is complete,
well-formed C++17 against the C ABI of include/artemis_hip.h.  Parthenon is not available, so the handful of
upstream names it uses are DECLARED (no behaviour) by tests/mock_parthenon/artemis.hpp and the compiler is run
with -fsyntax-only: this verifies every artemis_pack_t field and entry-point signature the adapter touches, not
the upstream API."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapter_compiles_against_the_c_abi(tmp_path):
    tu = tmp_path / "tu.cpp"
    tu.write_text('#include "artemis_hip_adapter.hpp"\nint parthenon::Globals::nghost = 2;\nint main() { return 0; }\n')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "mock_parthenon"),
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"), str(tu)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_adapter_fills_every_field_of_the_pack():
    """Every member of artemis_pack_t / artemis_fluid_pack_t must be assigned somewhere in the adapter."""
    import re
    hdr = open(os.path.join(ROOT, "include", "artemis_hip.h")).read()
    txt = open(os.path.join(ROOT, "integration", "artemis_hip_adapter.hpp")).read()
    fluid = re.search(r"typedef struct artemis_fluid_pack \{(.*?)\} artemis_fluid_pack_t;", hdr, re.S).group(1)
    pack = re.search(r"typedef struct artemis_pack \{(.*?)\} artemis_pack_t;", hdr, re.S).group(1)
    strip = lambda s: re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    names = lambda s: re.findall(r"[\*\s](\w+)(?:\[\d\])?\s*[;,]", strip(s))
    for n in names(fluid):
        if n in ("siefloor", "de_switch", "pflux", "vface", "diff_flux"):  # gas only
            assert re.search(r"p\.gas\.%s\b" % n, txt), n
        else:
            assert re.search(r"p\.gas\.%s\b" % n, txt) and re.search(r"p\.dust\.%s\b" % n, txt), n
    for n in names(pack):
        if n not in ("gas", "dust"):
            assert re.search(r"p\.%s\b" % n, txt), n
