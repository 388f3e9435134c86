#!/usr/bin/env python3
"""tools/show_bench.py FILE... -- one line per bench.py JSON result"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    parts = [path.split("/")[-1]]
    if d.get("value"):
        r = d["roofline"]
        parts.append(f"pdq {d['value'] / 1e6:.3f} M/s frac {r['frac']:.3f} ({r['kernel_ms']:.2f} ms, {d['config']['pdq_kernel']})")
    if "e2e" in d:
        e = d["e2e"]
        parts.append(f"e2e {e['images_per_s'] / 1e6:.3f} M img/s ({e['seconds_per_run'] * 1e3:.1f} ms, {e['groups']} groups)")
    if "hamming" in d:
        h = d["hamming"]
        parts.append(f"hamming thr {h['threshold']} {h['value']:.0f} Gpairs/s frac {h['roofline']['frac']:.3f} ({h['roofline']['kernel_ms']:.2f} ms, "
                     f"PW {h['roofline']['prefix_dwords']}, edges {h['edges_found']}/{h['edges_expected']}, allgather {h['allgather_ms']:.2f} ms)")
    if "hamming_10m" in d:
        h = d["hamming_10m"]
        parts.append(f"config 5: {h['n_hashes']} hashes {h['value']:.0f} Gpairs/s ({h['ms_per_step']:.0f} ms per sweep, sweep max {h['sweep_ms_max_over_ranks']:.0f} ms, "
                     f"allgather {h['allgather_ms_max_over_ranks']:.2f} ms, {h['ranks_in_collective']} rank(s), edges {h['edges_found']}/{h['edges_expected']})")
    if "jpeg" in d and "device_entropy" in d["jpeg"]:
        j = d["jpeg"]
        parts.append(f"jpeg {j['device_entropy']['files_per_s'] / 1e3:.0f} k files/s device entropy, {j['host_entropy']['files_per_s'] / 1e3:.1f} k host entropy, "
                     f"libjpeg-turbo 1 thread {j['cpu_baseline']['value'] / 1e3:.2f} k"
                     + (f", photos {j['photos_baseline']['files_per_s'] / 1e3:.1f} k / progressive {j['photos_progressive']['files_per_s'] / 1e3:.1f} k files/s" if "photos_baseline" in j else ""))
    parts.append(f"valid={d.get('valid')}")
    print(" | ".join(parts))
