#!/usr/bin/env python3
"""Join the PMC passes of tools/traffic_run.sh with the launch list of tools/conv_traffic.cpp into the summary bench.py
reports as roofline.traffic.

    python tools/traffic_summary.py profiles/r01_conv_traffic > profiles/r01_conv_traffic_summary.json

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3's FETCH_SIZE (KiB) counts
the 128-byte requests of wide (16 B / lane) coalesced reads at 64 B on gfx950 -> x2; WRITE_SIZE is "uncalibrated", so
both are calibrated here on the last launch of the list, a streaming kernel with a known byte count (act_bwd: reads
2 x 256 MiB, writes 256 MiB).  The counters sit on the L2's fabric side: Infinity-Cache hits are included, i.e. this
is an upper bound of the HBM bytes."""
import csv
import json
import sys


def per_launch(path):
    out = []
    for r in csv.DictReader(open(path)):
        if 'conv3x3_nhwc_kernel' in r['Kernel_Name'] or 'act_bwd_kernel' in r['Kernel_Name']:
            out.append((float(r['Counter_Value']) * 1024.0, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3))
    return out


def main(prefix):
    launches = [json.loads(l) for l in open(prefix + '_launches.jsonl')]
    fetch, write = per_launch(prefix + '_fetch_size.csv'), per_launch(prefix + '_write_size.csv')
    assert len(fetch) == len(write) == len(launches)
    cal = launches[-1]
    k_rd = cal['read_bytes_algorithmic'] / fetch[-1][0]          # 2.0 on gfx950 for 16 B / lane loads
    k_wr = cal['write_bytes_algorithmic'] / write[-1][0]         # 1.0 for 16 B / lane stores
    rows = []
    for l, (f, t), (w, _) in zip(launches[:-1], fetch[:-1], write[:-1]):
        rows.append({'name': l['name'], 'shape': [l['B'], l['H'], l['W'], l['Cin'], l['Cout']], 'us_under_pmc': round(t, 1),
                     'read_bytes': round(f * k_rd), 'write_bytes': round(w * k_wr),
                     'read_bytes_algorithmic': l['read_bytes_algorithmic'], 'write_bytes_algorithmic': l['write_bytes_algorithmic'],
                     'splitk_partials': l['splitk_ws_bytes'] > 0})
    n = len(rows)
    tot = sum(r['read_bytes'] + r['write_bytes'] for r in rows)
    alg = sum(r['read_bytes_algorithmic'] + r['write_bytes_algorithmic'] for r in rows)
    print(json.dumps({
        'what': 'fabric-side (L2 miss) bytes of conv3x3_nhwc_kernel, one launch per shape of the bench step; rocprofv3 --pmc '
                'FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic_run.sh)',
        'fetch_correction': round(k_rd, 4), 'write_correction': round(k_wr, 4), 'launches': n,
        'bytes_per_launch': tot / n, 'algorithmic_bytes_per_launch': alg / n, 'ratio': tot / alg, 'per_launch': rows}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'profiles/r01_conv_traffic')
