"""Summarise a rocprofv3 kernel trace: per-kernel mean duration and mean gap to the previous kernel on the same queue."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    k = r["Kernel_Name"].split("<")[0].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[k].append(e - s)
    if prev_end is not None: gap[k].append(s - prev_end)
    prev_end = e
for k in dur:
    d = sorted(dur[k]); g = sorted(gap[k]) or [0]
    print(f"{k[:60]:60s} calls={len(d):5d} median_dur={d[len(d)//2]/1e3:8.2f} us  median_gap_before={g[len(g)//2]/1e3:8.2f} us")
