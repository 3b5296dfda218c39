#!/usr/bin/env python3
"""Reads the traces of profiles/small_trace.py (kernel_trace.csv, memory_copy_trace.csv): period per call, GPU time per kernel and per stream, one call's timeline."""
import csv, re, collections, sys
k, m = sys.argv[1], sys.argv[2]
rows = []
for r in csv.DictReader(open(k)):
    mm = re.search(r'csdr::(\w+)', r['Kernel_Name'])
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), mm.group(1) if mm else 'other', int(r['Stream_Id'])))
for r in csv.DictReader(open(m)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY_' + r['Direction'][-14:], int(r['Stream_Id'])))
rows.sort()
chan = [i for i, r in enumerate(rows) if r[2].startswith('chan_analyze')]
i0, i1 = chan[-120], chan[-20]
print('calls traced %d; period per call %.1f us' % (len(chan), (rows[i1][0] - rows[i0][0]) / 100 / 1e3))
dur = collections.defaultdict(list); busy = collections.defaultdict(float)
for s, e, n, st in rows[i0:i1]:
    dur[n].append((e - s) / 1e3); busy[st] += (e - s) / 1e3
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('%-28s per call %.2f x %.1f us' % (n, len(v) / 100, sum(v) / len(v)))
print('stream busy per call (us):', {kk: round(v / 100, 1) for kk, v in busy.items()})
c0, c1 = chan[-60], chan[-58]
base = rows[c0][0]
for s, e, n, st in rows[c0:c1]:
    print('%8.1f %8.1f  %6.1f  stream %d  %s' % ((s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3, st, n))
