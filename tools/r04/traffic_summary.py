#!/usr/bin/env python3
"""Joins the FETCH_SIZE / WRITE_SIZE passes of tools/r03/traffic.sh into the JSON bench.py reports as roofline.traffic.
FETCH_SIZE is doubled (gfx950 rocprofv3 counts the 128-byte requests of 16-B-per-lane reads at 64 B:
MI355X_MICROARCH.md, HBM section; the same trace's row-maxima kernels over buffers of known size confirm the factor),
both counters are in KiB.  The counters sit on the L2's fabric side: Infinity-Cache hits are included (upper bound of HBM)."""
import csv
import json
import sys


def per_kernel(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and ('conv3x3_kernel' in r['Kernel_Name'] or 'conv3x3_ring_kernel' in r['Kernel_Name']):
            rows.append(float(r['Counter_Value']) * 1024.0)
    return rows


def main():
    fetch, write, replay = sys.argv[1:4]
    layers = [json.loads(l) for l in open(replay) if l.startswith('{"layer"')]
    f = [2.0 * v for v in per_kernel(fetch, 'FETCH_SIZE')]
    w = per_kernel(write, 'WRITE_SIZE')
    assert len(f) == len(w) == len(layers), (len(f), len(w), len(layers))
    per = []
    for l, fb, wb in zip(layers, f, w):
        per.append(dict(layer=l['layer'], read_bytes=fb, write_bytes=wb, algorithmic_read_bytes=l['algorithmic_read_bytes'],
                        algorithmic_write_bytes=l['algorithmic_write_bytes'], read_ratio=fb / l['algorithmic_read_bytes'],
                        write_ratio=wb / l['algorithmic_write_bytes']))
    n = len(layers)
    tot = sum(p['read_bytes'] + p['write_bytes'] for p in per)
    alg = sum(p['algorithmic_read_bytes'] + p['algorithmic_write_bytes'] for p in per)
    print(json.dumps(dict(kernel='pl::conv3x3_ring_kernel (11 layers) + pl::conv3x3_kernel (conv1_2)', launches=n, bytes_per_launch=tot / n, algorithmic_bytes_per_launch=alg / n,
                          ratio=tot / alg, read_bytes_per_step=sum(p['read_bytes'] for p in per),
                          write_bytes_per_step=sum(p['write_bytes'] for p in per), per_layer=per,
                          note='split-K partial sums are written by the conv kernel and re-read by pl::reduce_kernel (not in the read figure)')))


if __name__ == '__main__':
    main()
