#!/usr/bin/env python3
"""Per-step timeline of one steady-state unit (the encoder's GOPs in lock-step: 30 at 1080p = three passes of the clip) from a rocprofv3 --kernel-trace CSV:
usage: python tools/timeline.py <dir> [n_steps_to_print]
Prints, for the last full pass in the trace, every kernel of the first steps with start (us since the pass began),
duration and the gap to the previous kernel on the same queue, then per-kernel totals and the main-chain period."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
    r['q'] = r.get('Queue_Id', '?')
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if r['n'].startswith('k_intra_fwd')]
if len(idx) < 3:
    raise SystemExit("need at least three units in the trace")
mid = len(idx) // 2
a, b = idx[mid], idx[mid + 1]    # a pass from the middle of the run (steady state): from its k_intra_fwd to the next one
seg = rows[a:b]
t0 = seg[0]['s']
print("unit span %.1f us, %d kernels" % ((rows[b]['s'] - t0) / 1e3, len(seg)))
last_end = {}
me_starts = [r['s'] for r in seg if r['n'].startswith('k_me_pool')]
lim = me_starts[nshow] if len(me_starts) > nshow else seg[-1]['e']
for r in seg:
    gap = (r['s'] - last_end[r['q']]) / 1e3 if r['q'] in last_end else 0.0
    last_end[r['q']] = r['e']
    if r['s'] < lim:
        print("  q%-3s %-16s start %8.1f  dur %7.1f  gap %6.1f  grid %s" % (r['q'], r['n'], (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap, r.get('Grid_Size', '')))
tot = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    tot[r['n']][0] += (r['e'] - r['s']) / 1e3; tot[r['n']][1] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-16s n %3d  total %8.1f us  avg %7.1f" % (k, v[1], v[0], v[0] / v[1]))
per = [(me_starts[i + 1] - me_starts[i]) / 1e3 for i in range(len(me_starts) - 1)]
print("main chain period (k_me_pool start to start): mean %.1f min %.1f max %.1f" % (sum(per) / len(per), min(per), max(per)))
# busy union of all queues
ev = sorted([(r['s'], 1) for r in seg] + [(r['e'], -1) for r in seg])
busy = 0; depth = 0; prev = None
for t, d in ev:
    if depth > 0: busy += t - prev
    depth += d; prev = t
print("GPU busy (any kernel running) %.1f us of %.1f" % (busy / 1e3, (rows[b]['s'] - t0) / 1e3))
