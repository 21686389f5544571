"""Concurrent residency of the tone kernel's launches in a rocprofv3 --kernel-trace CSV: per launch geometry (= per
sub-bank size) the number of launches and queues, the mean kernel duration, the span from the first start to the last
end, the time covered by at least one / at least two launches, and the step (span / ticks).
Usage: python3 tools/trace_overlap.py <kernel_trace.csv> [kernel name substring]"""
import collections
import csv
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    merge = "--merge" in sys.argv           # all matching kernels as ONE group (banks on streams of their own: a tick is one launch of each)
    path = args[0]
    pat = args[1] if len(args) > 1 else "tone_fast_kernel"
    rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
    if not rows:
        print("no launches of", pat)
        return
    gkey = "Grid_Size_X" if "Grid_Size_X" in rows[0] else ("Grid_Size" if "Grid_Size" in rows[0] else None)
    groups = collections.OrderedDict()
    for r in rows:
        g = ("kernels matching '%s'" % pat, 0) if merge else (r["Kernel_Name"][:70], int(r[gkey]) if gkey else 0)
        groups.setdefault(g, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
    for (name, grid), ls in groups.items():
        ls.sort()
        # drop the first tenth (warm-up, digest launches); merged: the last third only (the timed region of tools/bench_paths.py)
        ls = ls[2*len(ls)//3:] if merge else ls[len(ls)//10:]
        queues = sorted(set(q for _, _, q in ls))
        dur = [e - s for s, e, _ in ls]
        ev = []
        for s, e, _ in ls:
            ev.append((s, 1))
            ev.append((e, -1))
        ev.sort()
        depth = 0
        last = ev[0][0]
        cover = collections.Counter()
        for t, d in ev:
            cover[depth] += t - last
            last = t
            depth += d
        span = max(e for _, e, _ in ls) - min(s for s, _, _ in ls)
        ge1 = sum(v for k, v in cover.items() if k >= 1)
        ge2 = sum(v for k, v in cover.items() if k >= 2)
        print("%s grid %d: %d launches on %d queue(s) %s" % (name, grid, len(ls), len(queues), queues))
        print("   mean kernel %.2f us (min %.2f, max %.2f); span %.1f us; >= 1 resident %.1f%%, >= 2 resident %.1f%%, idle %.1f%%"
              % (sum(dur)/len(dur)/1e3, min(dur)/1e3, max(dur)/1e3, span/1e3, 100.0*ge1/span, 100.0*ge2/span, 100.0*cover[0]/span))
        print("   span / launches = %.2f us per launch; x %d queue(s) = %.2f us per tick of the whole bank"
              % (span/len(ls)/1e3, len(queues), span/len(ls)/1e3*len(queues)))


if __name__ == "__main__":
    main()
