#!/usr/bin/env python3
"""Instruction mix of every loop (backward branch) of one kernel in a .s file.
usage: loopmix.py file.s kernel-name-substring"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(\S*%s\S*):" % re.escape(pat), s, re.M)
a = m.start()
b = s.index(".Lfunc_end", a)
body = s[a:b].split("\n")
labels = {}
for i, l in enumerate(body):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        labels[mm.group(1)] = i
for i, l in enumerate(body):
    mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        lo = labels[mm.group(1)]
        c = Counter()
        for x in body[lo:i + 1]:
            x = x.strip()
            if not x or x.startswith((".", ";")) or x.endswith(":"):
                continue
            c[x.split()[0]] += 1
        print("loop", mm.group(1), "lines", lo, i, "instrs", sum(c.values()))
        for k, v in c.most_common(30):
            print("   ", k, v)
