#!/usr/bin/env python3
"""Join the two PMC passes of tools/traffic_run.sh gemm (gpurun r05_c19) with the launch list of tools/gemm_traffic.cpp into
profiles/r05_gemm_traffic_summary.json -- what bench.py reports as roofline.traffic when the matrix products are the class that
takes more of the step.

    python tools/r05/gemm_traffic_summary.py profiles/r05_gemm_traffic_launches.jsonl profiles/r05_gemm_traffic_fetch_size.csv \\
           profiles/r05_gemm_traffic_write_size.csv > profiles/r05_gemm_traffic_summary.json

Like tools/traffic_summary.py (corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE counts the 128-byte
requests of 16 B / lane reads at 64 B on gfx950, WRITE_SIZE is uncalibrated -- both are calibrated on the last launch, a streaming
kernel of known byte count), but a call of mh_gemm_f32 is several kernels here: the product on plane images
(pl::gemm_ring_kernel or pl::gemm_kernel), in front of it the operand preparation (absmax + planes: in the step most of it is
cached or shared between products), behind it the K-slice reduction where there is one.  The counters sit on the L2's fabric side:
Infinity-Cache hits are included, i.e. an upper bound of the HBM bytes."""
import csv
import json
import sys


def rows_of(path):
    return [(r['Kernel_Name'], float(r['Counter_Value']) * 1024.0, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
            for r in csv.DictReader(open(path))]


def calls(rows):
    """[{'product': (bytes, us, name), 'prep': bytes, 'reduce': bytes}] per mh_gemm_f32 call + the calibration launch last"""
    out, prep = [], 0.0
    for name, val, us in rows:
        if 'pl::gemm' in name:
            out.append({'product': (val, us, name.split('(')[0][-60:]), 'prep': prep, 'reduce': 0.0})
            prep = 0.0
        elif 'act_bwd_kernel' in name:
            out.append({'product': (val, us, 'calibration'), 'prep': 0.0, 'reduce': 0.0})
        elif 'splitk_reduce' in name or 'reduce_kernel' in name:
            out[-1]['reduce'] += val
        elif 'fillBuffer' not in name:
            prep += val
    return out


def main(launch_path, fetch_path, write_path, tag='r05_c19'):
    launches = [json.loads(l) for l in open(launch_path)]
    fetch, write = calls(rows_of(fetch_path)), calls(rows_of(write_path))
    assert len(fetch) == len(write) == len(launches), (len(fetch), len(write), len(launches))
    k_rd = launches[-1]['read_bytes_algorithmic'] / fetch[-1]['product'][0]
    k_wr = launches[-1]['write_bytes_algorithmic'] / write[-1]['product'][0]
    rows = []
    for l, f, w in zip(launches[:-1], fetch[:-1], write[:-1]):
        rd, wr = f['product'][0] * k_rd, w['product'][0] * k_wr
        rows.append({'name': l['name'], 'shape': [l['M'], l['N'], l['K']], 'kernel': f['product'][2], 'k_slices': l['splitk'],
                     'us_under_pmc': round(f['product'][1], 1), 'read_bytes': round(rd), 'write_bytes': round(wr),
                     'read_bytes_algorithmic': l['read_bytes_algorithmic'],
                     'write_bytes_algorithmic': l['write_bytes_algorithmic'] * max(l['splitk'], 1) if l['splitk_ws_bytes'] else l['write_bytes_algorithmic'],
                     'read_ratio': round(rd / l['read_bytes_algorithmic'], 2),
                     'operand_preparation_read_bytes': round(f['prep'] * k_rd), 'operand_preparation_write_bytes': round(w['prep'] * k_wr),
                     'slice_reduction_read_bytes': round(f['reduce'] * k_rd), 'slice_reduction_write_bytes': round(w['reduce'] * k_wr)})
    n = len(rows)
    tot = sum(r['read_bytes'] + r['write_bytes'] for r in rows)
    alg = sum(r['read_bytes_algorithmic'] + r['write_bytes_algorithmic'] for r in rows)
    print(json.dumps({
        'what': 'fabric-side (L2 miss) bytes of the step\'s big matrix products on plane images, one launch per shape (fc6 forward / '
                'input gradient / weight gradient at 1536 rows, the 120-row object fc6, fc7); rocprofv3 --pmc FETCH_SIZE and --pmc '
                'WRITE_SIZE in separate passes (tools/traffic_run.sh gemm, gpurun %s)' % tag,
        'fetch_correction': round(k_rd, 4), 'write_correction': round(k_wr, 4), 'launches': n,
        'bytes_per_launch': tot / n, 'algorithmic_bytes_per_launch': alg / n, 'ratio': tot / alg, 'per_launch': rows}, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:5])
