#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Compile the reference's three CUDA kernel files for the CPU, from where they lie under
/root/reference, into oracle/_ref/*.so (git-ignored build outputs; no reference source enters the repository).

    python oracle/build_ref_cuda.py            (also: make -C oracle ref)

Per file: the source text is read, every kernel launch `name<<<cfg>>>(args);` is rewritten into
`cuda_cpu::launch([&]() { name(args); }, cfg);` (C++ cannot parse the chevrons; nothing else is touched), the result is
written to a temporary file under oracle/_ref/, compiled by g++ with oracle/cuda_cpu/cuda_on_cpu.h force-included
(-ffp-contract=off, see that header), and the temporary is deleted.  Entry points used by oracle/ref_cuda.py are the
reference's own extern "C" functions: ApplyNMSGPU, ROIAlignForwardLaucher / ROIAlignBackwardLaucher,
highway_lstm_forward_ongpu / highway_lstm_backward_ongpu.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('REF', '/root/reference')
OUT = os.path.join(HERE, '_ref')
SHIM = os.path.join(HERE, 'cuda_cpu')
FILES = {
    'nms_kernel': 'lib/fpn/nms/src/cuda/nms_kernel.cu',
    'roi_align_kernel': 'lib/fpn/roi_align/src/cuda/roi_align_kernel.cu',
    'highway_lstm_kernel': 'lib/lstm/highway_lstm_cuda/src/highway_lstm_kernel.cu',
}
LAUNCH = re.compile(r'(\b\w+)\s*<<<(.*?)>>>\s*\((.*?)\)\s*;', re.S)


def rewrite(text):
    text, n = LAUNCH.subn(lambda m: '::cuda_cpu::launch([&]() { %s(%s); }, %s);' % (m.group(1), m.group(3), m.group(2)), text)
    return text, n


def build(force=False):
    if not os.path.isdir(REF):
        raise RuntimeError('%s is not available: oracle/_ref can only be built in the build container' % REF)
    os.makedirs(OUT, exist_ok=True)
    built = []
    for name, rel in FILES.items():
        src = os.path.join(REF, rel)
        so = os.path.join(OUT, 'ref_%s.so' % name)
        deps = [src, os.path.join(SHIM, 'cuda_on_cpu.h'), os.path.abspath(__file__)]
        if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
            built.append(so)
            continue
        text, n = rewrite(open(src).read())
        assert n >= 1, 'no kernel launch found in %s' % src
        tmp = os.path.join(OUT, '_tmp_%s.cpp' % name)
        with open(tmp, 'w') as f:
            f.write(text)
        try:
            subprocess.check_call(['g++', '-x', 'c++', '-std=c++14', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math', '-w',
                                   '-DCUDA_ON_CPU_DEFINE_GLOBALS', '-I', SHIM, '-I', os.path.dirname(src),
                                   '-include', os.path.join(SHIM, 'cuda_on_cpu.h'), tmp, '-o', so, '-lm'])
        finally:
            os.remove(tmp)
        built.append(so)
    return built


if __name__ == '__main__':
    for so in build(force='--force' in sys.argv):
        print(so)
