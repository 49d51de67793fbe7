#!/usr/bin/env python3
"""Per-layer PMC table of the trunk's convolution launches: joins the counter passes of tools/r04/pmc.sh
over `pl_check --conv-replay [N]` (the trunk layers of the cfg2 step, each with the epilogue the step uses, N launches per layer: the
last one is tabulated) with the replay's
launch list.

    python tools/r06/pmc_table.py <dir with set1.csv set2.csv set3.csv> <replay.jsonl> > profiles/r06_pmc_ring_table.txt

clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; matrix pipe busy = 32 cycles x SQ_INSTS_MFMA (v_mfma_f32_32x32x16_f16: 8 passes of 4
cycles) / (1024 SIMDs x duration x clock); x/M = instructions of class x per MFMA instruction (VALU: without the MFMAs themselves); wait_any = SQ_WAIT_ANY /
SQ_WAVE_CYCLES; LDS confl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  A launch that cuts its K range into slices or runs a separate
reduction shows the sum of its kernels under `us` (counters: the conv kernel alone).  Profiled passes clock lower than unprofiled runs."""
import collections
import csv
import json
import re
import sys


def launches(path, pattern):
    """[{counter: value}, us, kernel name] per dispatch of a matching kernel, in dispatch order (counter rows of one dispatch summed)"""
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if not re.search(pattern, r['Kernel_Name']):
            continue
        d = by.setdefault(int(r['Dispatch_Id']), [collections.defaultdict(float), 0.0, r['Kernel_Name']])
        d[0][r['Counter_Name']] += float(r['Counter_Value'])
        d[1] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    return [by[k] for k in sorted(by)]


def main(d, replay):
    layers = [json.loads(l) for l in open(replay) if l.startswith('{"layer"')]
    pat = r'conv3x3_ring_kernel|pl::conv3x3_kernel'
    sets = [launches('%s/set%d.csv' % (d, i), pat) for i in (1, 2, 3)]
    reps = len(sets[0]) // len(layers)                  # pl_check --conv-replay N: N launches per layer, the last one (warm) is tabulated
    assert reps >= 1 and all(len(s) == reps * len(layers) for s in sets), ([len(s) for s in sets], len(layers))
    sets = [s[reps - 1::reps] for s in sets]
    print(__doc__.split('\n\n')[2].replace('\n', '\n# ').join(['# ', '']))
    print('%-9s %-13s %-26s %7s %7s %6s %9s %7s %7s %7s %7s %9s %9s' % (
        'layer', 'epilogue', 'kernel', 'us', 'TF/s', 'GHz', 'pipe busy', 'VALU/M', 'SALU/M', 'LDS/M', 'VMEM/M', 'wait_any', 'LDS confl'))
    for l, a, b, c in zip(layers, *sets):
        us = a[1]
        ghz = a[0]['GRBM_GUI_ACTIVE'] / 8.0 / us * 1e-3
        mf = c[0]['SQ_INSTS_MFMA']
        busy = 32.0 * mf / (1024.0 * c[1] * 1e3 * (ghz if ghz > 0 else 1.0)) if mf else 0.0
        kern = re.sub(r'.*conv3x3_(ring_)?kernel<', '', a[2]).split('>(')[0].replace('mh::pl::', '')[:26]
        print('%-9s %-13s %-26s %7.1f %7.1f %6.2f %9.2f %7.2f %7.2f %7.2f %7.2f %9.2f %9.3f' % (
            l['layer'], l['epilogue'], kern, us, l['flops'] / us * 1e-6, ghz, busy, (c[0]['SQ_INSTS_VALU'] - mf) / mf, c[0]['SQ_INSTS_SALU'] / mf,
            c[0]['SQ_INSTS_LDS'] / mf, c[0]['SQ_INSTS_VMEM_RD'] / mf, a[0]['SQ_WAIT_ANY'] / max(a[0]['SQ_WAVE_CYCLES'], 1.0),
            c[0]['SQ_LDS_BANK_CONFLICT'] / max(c[0]['SQ_LDS_IDX_ACTIVE'], 1.0)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
