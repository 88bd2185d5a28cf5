#!/usr/bin/env python3
"""Per-batch average kernel time per round from the RB2_HIP_TRACE=1 lines on stderr ([rb2_hip] round N kernel ms; every tenth
round is printed).  usage: trace_summary.py <stderr file> [first_round]   (rounds below first_round are skipped: default 100)"""
import re, collections, sys
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
batches, acc = [], collections.defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    if 'batch done' in line:
        batches.append(acc); acc = collections.defaultdict(lambda: [0, 0.0]); continue
    m = re.match(r'\[rb2_hip\] round\s+(\d+) (\S+)\s+([\d.]+) ms', line)
    if m and int(m.group(1)) >= first:
        a = acc[m.group(2)]; a[0] += 1; a[1] += float(m.group(3))
for i, b in enumerate(batches):
    rounds = b['k_advance'][0] if 'k_advance' in b else 0      # (k_tscan is timed twice per round: counting phase + setup)
    per = {k: round(v[1] / max(1, rounds) * 1000, 1) for k, v in b.items() if k != 'k_relayout'}
    print("batch %d: us per round %s  sum %.0f" % (i, per, sum(per.values())))
