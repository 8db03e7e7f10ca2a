"""Reference-held known answers for the hot path, loaded from tests/golden/reference_pins.json
(numbers copied from the reference's regression scripts, cited there per entry)."""
import json
import os

PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_pins.json")))
LINWAVE, ADVECTION, DRAG, SSHEET, DISK = PINS["linwave"], PINS["advection"], PINS["drag"], PINS["ssheet"], PINS["disk"]
LANDINGS = PINS["oracle_landings"]


def linwave_waves():
    return [(w["wave_flag"], w["vflow"]) for w in LINWAVE["waves"]]


def advection_history():
    h = ADVECTION["history_n32"]
    g = h["gas"]
    out = [g["mass"], *g["momentum"], g["energy"], g["internal_energy"]]
    for k in ("dust1", "dust2"):
        out += [h[k]["mass"], *h[k]["momentum"]]
    return out
