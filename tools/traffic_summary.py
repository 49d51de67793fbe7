#!/usr/bin/env python3
"""Join the PMC passes of tools/traffic_run.sh with the launch list of tools/conv_traffic.cpp / tools/gemm_traffic.cpp into
the summary bench.py reports as roofline.traffic (conv) and DESIGN.md quotes for the GEMMs.

    python tools/traffic_summary.py <launches.jsonl> <FETCH_SIZE.csv> <WRITE_SIZE.csv> <kernel name substring> > summary.json

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3's FETCH_SIZE (KiB) counts the
128-byte requests of wide (16 B / lane) coalesced reads at 64 B on gfx950 -> x2; WRITE_SIZE is "uncalibrated", so both are
calibrated here on the last launch of the list, a streaming kernel with a known byte count (act_bwd: reads 2 x 256 MiB,
writes 256 MiB).  The counters sit on the L2's fabric side: Infinity-Cache hits are included, i.e. this is an upper bound
of the HBM bytes.  Kernels of the same call that are not the named one (row-exponent passes, partial-sum reductions) are
listed per launch under `other_kernels_read_bytes` / `..._write_bytes` (attributed to the preceding named launch)."""
import csv
import json
import sys


def rows_of(path):
    return [(r['Kernel_Name'], float(r['Counter_Value']) * 1024.0, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
            for r in csv.DictReader(open(path))]


def group(rows, kernel):
    """[(named-kernel bytes, us, bytes of the other kernels up to the next named launch)] + the calibration launch last"""
    out, cur = [], None
    for name, val, us in rows:
        if kernel in name or 'act_bwd_kernel' in name:
            cur = [val, us, 0.0]
            out.append(cur)
        elif cur is not None:
            cur[2] += val
    return out


def main(launch_path, fetch_path, write_path, kernel):
    launches = [json.loads(l) for l in open(launch_path)]
    fetch, write = group(rows_of(fetch_path), kernel), group(rows_of(write_path), kernel)
    assert len(fetch) == len(write) == len(launches), (len(fetch), len(write), len(launches))
    cal = launches[-1]
    k_rd = cal['read_bytes_algorithmic'] / fetch[-1][0]          # 2.0 on gfx950 for 16 B / lane loads
    k_wr = cal['write_bytes_algorithmic'] / write[-1][0]         # 1.0 for 16 B / lane stores
    rows = []
    for l, (f, t, fo), (w, _, wo) in zip(launches[:-1], fetch[:-1], write[:-1]):
        shape = [l[k] for k in ('B', 'H', 'W', 'Cin', 'Cout') if k in l] or [l[k] for k in ('M', 'N', 'K')]
        rows.append({'name': l['name'], 'shape': shape, 'us_under_pmc': round(t, 1), 'tflops_under_pmc': round(l['flops'] / t / 1e6, 1),
                     'read_bytes': round(f * k_rd), 'write_bytes': round(w * k_wr),
                     'read_bytes_algorithmic': l['read_bytes_algorithmic'], 'write_bytes_algorithmic': l['write_bytes_algorithmic'],
                     'read_ratio': round(f * k_rd / l['read_bytes_algorithmic'], 2),
                     'other_kernels_read_bytes': round(fo * k_rd), 'other_kernels_write_bytes': round(wo * k_wr),
                     'splitk_partials': l.get('splitk_ws_bytes', 0) > 0})
    n = len(rows)
    tot = sum(r['read_bytes'] + r['write_bytes'] for r in rows)
    alg = sum(r['read_bytes_algorithmic'] + r['write_bytes_algorithmic'] for r in rows)
    print(json.dumps({
        'what': 'fabric-side (L2 miss) bytes of %s, one launch per shape; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in '
                'separate passes (tools/traffic_run.sh)' % kernel,
        'fetch_correction': round(k_rd, 4), 'write_correction': round(k_wr, 4), 'launches': n,
        'bytes_per_launch': tot / n, 'algorithmic_bytes_per_launch': alg / n, 'ratio': tot / alg, 'per_launch': rows}, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:5])
