#!/usr/bin/env python3
"""Which kernels of the two-step job run at the same time (window lanes): from a rocprofv3 --kernel-trace CSV, the time each kernel
class is on the GPU, the time it runs alone, and the time it shares with every other class.
usage: python tools/timeline_overlap.py <kernel_trace.csv> [t_from_frac t_to_frac]"""
import csv
import sys
from collections import defaultdict


def cls(name):
    for key, c in (("k_bm_scan", "scan"), ("k_self_select", "select"), ("k_stereo_argmin", "argmin"), ("k_group_pos", "pos"), ("k_group_shape", "shape"),
                   ("_list", "group_list"), ("k_group", "group"), ("k_aggregate", "aggregate"), ("k_copy", "copy"), ("k_symetrize", "window"), ("k_window", "window")):
        if key in name:
            return c
    return "other"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])) for r in rows]
    t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
    if len(sys.argv) > 3:
        a, b = float(sys.argv[2]), float(sys.argv[3])
        t0, t1 = t0 + int(a * (t1 - t0)), t0 + int(b * (t1 - t0))
        ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    pts = sorted([(s, 1, c) for s, e, c in ev] + [(e, -1, c) for s, e, c in ev])
    live = defaultdict(int)
    alone, total, pair, idle, busy = defaultdict(int), defaultdict(int), defaultdict(int), 0, 0
    prev = pts[0][0]
    for t, d, c in pts:
        dt = t - prev
        if dt > 0:
            act = sorted(k for k, v in live.items() if v > 0)
            if not act:
                idle += dt
            else:
                busy += dt
                for k in act:
                    total[k] += dt
                if len(act) == 1:
                    alone[act[0]] += dt if live[act[0]] == 1 else 0
                    if live[act[0]] > 1:
                        pair[(act[0], act[0])] += dt
                for i, k in enumerate(act):
                    for m in act[i + 1:]:
                        pair[(k, m)] += dt
        live[c] += d
        prev = t
    dur = defaultdict(int)
    for s, e, c in ev:
        dur[c] += e - s
    span = t1 - t0
    print(f"span {span / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms, sum of kernel durations {sum(dur.values()) / 1e6:.2f} ms")
    print("class        sum_ms  on_gpu_ms  alone_ms")
    for c in sorted(dur, key=lambda k: -dur[k]):
        print(f"{c:12s} {dur[c] / 1e6:7.2f} {total[c] / 1e6:9.2f} {alone[c] / 1e6:9.2f}")
    print("pairs on the GPU together (ms):")
    for (a, b), v in sorted(pair.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  {a:10s} + {b:10s} {v / 1e6:7.2f}")


if __name__ == "__main__":
    main()
