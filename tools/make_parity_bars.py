#!/usr/bin/env python3
"""tests/golden/parity_bars.json from an observation run of the GPU tests:
    IA_PARITY_OBSERVE=gpurun_out/parity_obs.json python -m pytest tests -m gpu -q          (on the MI355X)
    python tools/make_parity_bars.py gpurun_out/parity_obs.json [profiles/r05_parity_observed.json]
bar = 3 x observed per component (floor 1e-7: a difference of exactly zero still leaves room for a last-bit change); the tests
apply their own hard caps on top (tests/parity_bars.py)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    obs = json.load(open(sys.argv[1]))
    bars = {k: [max(3.0 * x, 1e-7) for x in v] for k, v in sorted(obs.items()) if len(v) == 3}
    json.dump(dict(note="3 x the MI355X observation (tools/make_parity_bars.py); (max, p99, mean) of the per-row absolute difference",
                   bars=bars), open(os.path.join(ROOT, "tests", "golden", "parity_bars.json"), "w"), indent=0, sort_keys=True)
    if len(sys.argv) > 2:
        shutil.copy(sys.argv[1], sys.argv[2])
    print(len(bars), "bars")


if __name__ == "__main__":
    main()
