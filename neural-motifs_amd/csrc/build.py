#!/usr/bin/env python3
"""Build libmotifs_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python neural-motifs_amd/csrc/build.py [--force] [--verbose]

exact_ops.hip is compiled with -ffp-contract=off (bit-exact NMS / RoIAlign / rasteriser, mirroring
oracle/native_ops.c); the MFMA files use the default contraction.  The shared object is written next to the
sources (in-tree, git-ignored) so that it travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libmotifs_hip.so')
ARCH = 'gfx950'
EXACT = ('exact_ops.hip',)
SOURCES = ('exact_ops.hip', 'gemm.hip', 'pl_gemm.hip', 'pl_conv.hip', 'conv.hip', 'lstm.hip', 'optim.hip', 'tower.hip')
HEADERS = ('common.h', 'mfma_tile.h', 'pl_tile.h', 'pl_ring.h', os.path.join('..', '..', 'include', 'motifs_hip.h'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
PACKED_OK = ('conv.hip', 'pl_gemm.hip', 'pl_conv.hip')     # see build(): packed FP32 VALU ops only here
# (gemm.hip left this list in round 5: its bf16x6 split is subtractions of a widened bf16 from the fp32 value -- the compiler
# pairs them into v_pk_add_f32 with a neg modifier, the very instruction class that went wrong under MFMA co-residency, and
# the small products run on the side stream next to the main stream's chip-filling MFMA kernels all the time)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_flags(src):
    """the device-code-relevant flags of one source file (shared by build() and tests/test_cabi.py's ISA check)"""
    cmd = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-Wall', '-Wno-unused-function']
    if src in EXACT:
        cmd += ['-ffp-contract=off']
    # No packed-FP32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) outside the MFMA tile engines.  On
    # MI355X the packed ops of the RoIAlign kernel (SLP-vectorised bilinear interpolation: v_pk_add_f32 with neg
    # modifiers, v_pk_mul_f32 with op_sel broadcasts) returned wrong LOW halves in lanes 48-63 while waves of an
    # MFMA + packed-VALU kernel from ANOTHER HIP stream shared its SIMD: 44 of 45 concurrent launches wrong next to the
    # in-loop-split conv, 0 of 45 with the scalar forms; alone every kernel is bit-exact.  The two-stream evaluation
    # forward was wrong by 1e-2 because of it.  Evidence and the search that led here: profiles/r03_packed_f32_
    # coresidency.txt, tools/r03/diag2..diag9.  The tile engines (PACKED_OK) keep their packed ops: the in-loop-split
    # kernels are VALU-bound (-12 % step throughput without them) and as VICTIMS they came out clean in the
    # co-residency matrix; everything else is latency- or HBM-bound and loses nothing.  MH_PACKED_F32=1 / 0 forces
    # the compiler default / the scalar forms for every file (A/B builds).
    packed = os.environ.get('MH_PACKED_F32')
    if packed == '0' or (packed != '1' and src not in PACKED_OK):
        cmd += ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
    for knob in ('MH_MINW', 'MH_CONV_TAP_MAJOR', 'MH_BAR_SLEEP', 'MH_F16_VALU', 'MH_ATOMIC_ALWAYS'):
        if os.environ.get(knob):
            cmd += ['-D%s=%s' % (knob, os.environ[knob])]
    return cmd


def build(force=False, verbose=False, out_dir=None):
    """out_dir (or $MH_OUT): where objects and the .so go -- default next to the sources.  A variant build (a build knob
    below, MH_PACKED_F32) belongs in its own directory, e.g. csrc/_variants/pk, and is loaded with MOTIFS_HIP_LIB=<that .so>
    (lib/_hip.py)."""
    out_dir = os.path.abspath(out_dir or os.environ.get('MH_OUT') or HERE)
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, os.path.basename(SO))
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(out_dir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs + [os.path.abspath(__file__)]):
            cmd = [HIPCC] + compile_flags(src) + ['-fPIC', '-c', s, '-o', o]
            if verbose:
                cmd += ['-Rpass-analysis=kernel-resource-usage']
                print(' '.join(cmd))
            jobs.append(cmd)
    # the translation units are independent: up to MH_JOBS (default 4) hipcc processes at a time
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, int(os.environ.get('MH_JOBS', '4')))) as pool:
        for rc in pool.map(lambda c: subprocess.call(c), jobs):
            if rc:
                raise subprocess.CalledProcessError(rc, 'hipcc')
    if force or _stale(so, objs):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', so] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return so


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
