#!/usr/bin/env python3
"""one-line summary of a bench.py JSON line: `show_bench.py FILE [label]` (or `- label` to read stdin)"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "-"
text = sys.stdin.read() if src == "-" else open(src).read()
d = json.loads(text.strip().splitlines()[-1])
h = d["hamming"]
print(f"{sys.argv[2] if len(sys.argv) > 2 else ''} pdq {d['value']/1e6:.3f} M/s frac {d['roofline']['frac']:.3f} ({d['roofline']['kernel_ms']:.2f} ms, "
      f"{d['config'].get('pdq_kernel')}) | hamming {h['value']:.0f} Gpairs/s ({h['roofline']['kernel_ms']:.2f} ms, edges {h['edges_found']}/{h['edges_expected']})")
