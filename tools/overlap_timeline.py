#!/usr/bin/env python3
"""How much do kernels actually CO-RUN on the device, and which pairs?  From rocprofv3 --kernel-trace CSVs of one process (two streams /
host threads inside it) or of several processes sharing the GPU (one CSV each; the timestamps are on one clock).

    python tools/overlap_timeline.py <kernel_trace.csv> [<kernel_trace.csv> ...] [--steps K]

The analysed window is the last K steps of every trace (a step ends with the fused Adam kernel), intersected over the traces.  Output
(JSON): span, time with >= 1 / >= 2 kernels in flight, kernel-time sum (sum / span = average kernels in flight), the co-running pairs by
time (short names; 'a | a' = two launches of the same kernel), hardware queues used per trace, and per kernel family its time alone vs
together with another kernel."""
import argparse
import collections
import csv
import json
import re


def short(name):
    m = re.search(r"(?:anonymous namespace\)::)?([A-Za-z_0-9]+)(?:<[^(]*)?\(", name)
    s = m.group(1) if m else name[:40]
    if s.startswith("vectorized_elementwise") or s.startswith("elementwise_kernel") or "at::native" in name:
        return "ATen"
    return s


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def window(rows, k, skip=0):
    ends = [e for s, e, n, q in rows if n == "adam_kernel"]
    if skip:
        ends = ends[:-skip]
    if len(ends) < k + 1:
        return rows[0][0], rows[-1][1]
    return ends[-(k + 1)], ends[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("traces", nargs="+")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--skip-last", type=int, default=0, help="leave out the last K steps of the trace (instrumented repeats after the timed steps)")
    args = ap.parse_args()
    traces = [load(p) for p in args.traces]
    wins = [window(t, args.steps, args.skip_last) for t in traces]
    w0, w1 = max(w[0] for w in wins), min(w[1] for w in wins)
    evs = []
    queues = []
    for ti, t in enumerate(traces):
        q = collections.Counter()
        for s, e, n, qid in t:
            if e <= w0 or s >= w1:
                continue
            s, e = max(s, w0), min(e, w1)
            evs.append((s, 1, n, ti))
            evs.append((e, -1, n, ti))
            q[qid] += 1
        queues.append(dict(q))
    evs.sort(key=lambda x: (x[0], x[1]))
    active = collections.Counter()
    n_active = 0
    last = w0
    t_ge1 = t_ge2 = ksum = 0
    pair = collections.Counter()
    alone = collections.Counter()
    together = collections.Counter()
    for ts, d, name, ti in evs:
        dt = ts - last
        if dt > 0 and n_active > 0:
            t_ge1 += dt
            ksum += dt * n_active
            names = sorted(n for n, c in active.items() for _ in range(c))
            if n_active >= 2:
                t_ge2 += dt
                a, b = names[0], names[1]
                pair[f"{a} | {b}"] += dt
                for n_ in set(names):
                    together[n_] += dt
            else:
                alone[names[0]] += dt
        last = ts
        active[name] += d
        if active[name] == 0:
            del active[name]
        n_active += d
    span = w1 - w0
    ms = lambda v: round(v / 1e6, 3)      # noqa: E731
    fam = {}
    for n_ in set(alone) | set(together):
        fam[n_] = dict(alone_ms=ms(alone[n_]), with_another_ms=ms(together[n_]))
    top = sorted(fam.items(), key=lambda kv: -(kv[1]["alone_ms"] + kv[1]["with_another_ms"]))[:14]
    print(json.dumps(dict(traces=args.traces, steps=args.steps, span_ms=ms(span), per_step_span_ms=ms(span / max(args.steps, 1)),
                          busy_ge1_ms=ms(t_ge1), busy_ge2_ms=ms(t_ge2), frac_ge2=round(t_ge2 / max(span, 1), 4), idle_frac=round(1 - t_ge1 / max(span, 1), 4),
                          kernel_time_sum_ms=ms(ksum), avg_kernels_in_flight=round(ksum / max(span, 1), 3),
                          queues_used=queues, top_pairs_ms=[(k, ms(v)) for k, v in pair.most_common(14)], kernel_families=dict(top))))


if __name__ == "__main__":
    main()
