#!/usr/bin/env python3
"""tests/golden/grad_parity_bars.json from an observation run of the backward-golden comparison:
    python -m tests.test_gpu_backward_golden > gpurun_out/r06_grad_parity.json          (on the MI355X)
    python tools/make_grad_parity_bars.py gpurun_out/r06_grad_parity.json [profiles/r06_grad_parity.json]
Per parameter group (rel_max, cos_dist) and per table statistic: 3 x the observation, with floors (1e-4 / 1e-8 for the groups, 1e-4 for the
table statistics) so that a last-bit change of a reduction order does not trip a bar that was observed at 1e-6; the test applies its
hard caps on top (tests/test_gpu_backward_golden.py CAP / TABLE_CAP).  Only the forward_train_ route's observations make bars; the
forward_backward_phys route is held to the same bars."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    obs = json.load(open(sys.argv[1]))
    groups, tables = {}, {}
    for name, rep in obs.items():
        if "/" in name:
            continue
        for p, st in rep["groups"].items():
            if st["ref_max"] > 1e-30:
                groups[f"{name}/{p}"] = [max(3.0 * st["rel_max"], 1e-4), max(3.0 * st["cos_dist"], 1e-8)]
        for p, st in rep["tables"].items():
            tables[f"{name}/{p}"] = {k: max(3.0 * st[k], 1e-4) for k in ("level_sum", "level_l1", "level_l2", "level_probe", "sub_rel_max", "sub_cos_dist")}
    note = (f"3 x the MI355X observation of {sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]} (python -m tests.test_gpu_backward_golden, "
            "tools/make_grad_parity_bars.py; floors 1e-4 / 1e-8); the test also applies hard caps")
    json.dump(dict(groups=groups, tables=tables, note=note), open(os.path.join(ROOT, "tests", "golden", "grad_parity_bars.json"), "w"), indent=0, sort_keys=True)
    if len(sys.argv) > 2:
        shutil.copy(sys.argv[1], sys.argv[2])
    print(len(groups), "group bars,", len(tables), "table bars")


if __name__ == "__main__":
    main()
