"""C-ABI checks that need no GPU: the shared object builds/loads and exports exactly the symbols
include/motifs_hip.h declares; the product path refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'motifs_hip.h')


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mh_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def so_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location('mh_build', os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_header_symbols_are_exported(so_path):
    lib = ctypes.CDLL(so_path)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), 'header declares %s but the library does not export it' % name
    assert lib.mh_version() >= 100


def test_python_binding_lists_every_header_symbol(so_path):
    from lib import _hip
    assert sorted(_hip.SYMBOLS) == _declared_symbols()
    _hip.lib()


def test_argument_validation_needs_no_device(so_path):
    lib = ctypes.CDLL(so_path)
    lib.mh_last_error.restype = ctypes.c_char_p
    lib.mh_gemm_ws_bytes.restype = ctypes.c_size_t
    lib.mh_nms_ws_bytes.restype = ctypes.c_size_t
    assert lib.mh_gemm_auto_splitk(120, 4096, 25088) >= 2          # 32 tiles: split K to fill 256 CUs
    assert lib.mh_gemm_auto_splitk(8214, 512, 4608) == 1 or lib.mh_gemm_auto_splitk(8214, 512, 4608) >= 1
    assert lib.mh_gemm_ws_bytes(120, 4096, 25088, 4) >= 4 * 120 * 4096 * 4
    assert lib.mh_nms_ws_bytes(6000) >= 6000 * 94 * 8
    # bad arguments are reported, never abort()/exit()
    rc = lib.mh_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, None, ctypes.c_size_t(0), None)
    assert rc == -1 and b'bad argument' in lib.mh_last_error()
    # size queries and validation of the entry points added for the widened scope (no kernel is launched)
    for name in ('mh_conv3x3_packed_floats', 'mh_conv3x3_wgrad_ws_bytes', 'mh_hwcell_seq_ws_bytes', 'mh_hwlstm_fwd_ws_bytes'):
        getattr(lib, name).restype = ctypes.c_size_t
    assert lib.mh_mfma_split() in (0, 3, 6)
    assert lib.mh_split_f16() == 1 and lib.mh_mfma_split() == 3      # the shipped default since round 2: f16x3
    assert lib.mh_split_rne() == 0             # (MH_SPLIT_RN=1 is a knob of the bf16x6 variant build)
    if lib.mh_split_f16():
        # two f16 planes (= the fp32 byte count) + one dword per (tap, output channel) holding the channel exponents
        assert lib.mh_conv3x3_packed_floats(128, 64) == 9 * 128 * ((64 // 16) * 16 + 1)
    elif lib.mh_mfma_split():
        assert lib.mh_conv3x3_packed_floats(128, 64) == 9 * 128 * (64 // 16) * 24     # bf16 planes: 1.5 x the fp32 weights
    else:
        assert lib.mh_conv3x3_packed_floats(128, 64) == 9 * 128 * 64
    assert lib.mh_conv3x3_wgrad_ws_bytes(6, 37, 37, 512, 512) >= 6 * 37 * 37 * 2          # at least the tap masks
    assert lib.mh_hwcell_seq_ws_bytes() >= 4
    assert lib.mh_hwlstm_fwd_ws_bytes(4424, 512, 6, 2, 20) > 20 * 6 * 6 * 512 * 4
    assert lib.mh_conv3x3_wgrad(None, None, 1, 8, 8, 16, 16, None, None, ctypes.c_size_t(0), None) in (-1, -2)
    assert lib.mh_triplet_match(None, None, 3, None, None, 3, ctypes.c_double(0.5), None, None, None) == -1
    assert lib.mh_hwcell_seq_fwd(513, 4, 3, None, None, None, None, None, None, None, None, None, ctypes.c_size_t(0), None) == -1
    assert lib.mh_bn_apply_nhwc(None, ctypes.c_longlong(8), 16, None, None, None, None, None, 0, None, None) == -1
    assert lib.mh_maxpool2x2_bwd_nhwc(None, None, 1, 4, 4, 8, None, None) == -1
    # optimizer table: nothing to expand is fine, a record list without a destination (or with a null tensor) is not
    assert lib.mh_opt_build_chunks(None, 0, None, 0, None) == 0
    assert lib.mh_opt_build_chunks(None, 2, None, 2, None) == -1
    rec = (ctypes.c_uint64 * 4)(0, 0, 0, 0)                    # {p, g, buf} = NULL, n = 0: rejected before any launch
    assert lib.mh_opt_build_chunks(rec, 1, ctypes.c_void_p(1 << 20), 1, None) == -1
    assert lib.mh_roi_align_bwd_det(None, 1, 2048, 4, 4, 1, None, 1, 7, 7, ctypes.c_float(1.0), None, None) == -1
    # the direct first convolution of the mask tower: sizes, and the shapes it refuses (the caller then takes im2col + GEMM)
    ll = ctypes.c_longlong
    lib.mh_tower_conv1_padded_bytes.restype = ctypes.c_size_t
    lib.mh_tower_conv1_wgrad_ws_bytes.restype = ctypes.c_size_t
    assert lib.mh_tower_conv1_out_size(27) == 14
    assert lib.mh_tower_conv1_padded_bytes(ll(1536), 27) >= 1536 * 33 * 33 * 2 * 4
    assert lib.mh_tower_conv1_wgrad_ws_bytes(ll(1536), 256) >= 512 * 99 * 256 * 4
    one = ctypes.c_void_p(1 << 20)
    assert lib.mh_tower_conv1_fwd(one, ll(4), 27, one, None, 128, one, None) == -1          # C0 not a multiple of 256
    assert lib.mh_tower_conv1_fwd(one, ll(4), 28, one, None, 256, one, None) == -1          # windows do not tile the padded mask
    assert lib.mh_tower_conv1_fwd(one, ll(4), 39, one, None, 256, one, None) == -1          # 20 outputs per row: beyond the register row
    assert lib.mh_tower_conv1_fwd(None, ll(4), 27, one, None, 256, one, None) == -1
    assert lib.mh_tower_conv1_wgrad(one, one, ll(4), 27, 256, one, one, ctypes.c_size_t(16), None) == -1      # workspace too small
    assert lib.mh_tower_conv1_pad(one, ll(0), 27, one, None) == -1


def test_direct_tower_convolution_is_chosen_only_for_the_shapes_it_covers():
    from lib import _hip
    w = torch.zeros(256, 2, 7, 7)
    assert _hip.tower_conv1_supported(torch.zeros(5, 27, 27, 2), w)
    assert _hip.tower_conv1_supported(torch.zeros(5, 27, 27, 2), torch.zeros(512, 2, 7, 7))       # the ResNet tower
    assert not _hip.tower_conv1_supported(torch.zeros(0, 27, 27, 2), w)                           # no pairs: the old path handles it
    assert not _hip.tower_conv1_supported(torch.zeros(5, 27, 27, 2), torch.zeros(128, 2, 7, 7))   # default dim = 256 -> 128 channels
    assert not _hip.tower_conv1_supported(torch.zeros(5, 27, 27, 2), torch.zeros(256, 2, 3, 3))
    assert not _hip.tower_conv1_supported(torch.zeros(5, 28, 28, 2), w)
    assert not _hip.tower_conv1_supported(torch.zeros(5, 39, 39, 2), w)
    assert not _hip.tower_conv1_supported(torch.zeros(5, 27, 27, 2, dtype=torch.float64), w)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_hot_path_fails_loudly_without_gpu(so_path):
    from lib import _hip
    with pytest.raises(_hip.HipKernelError):
        _hip.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    with pytest.raises(_hip.HipKernelError):
        _hip.nms(torch.zeros(4, 4), 0.5)


def test_no_packed_fp32_valu_outside_the_tile_engines(tmp_path):
    """csrc/build.py's packed-FP32 policy: every source outside the MFMA tile engines carries the flag
    that removes v_pk_{add,mul,fma}_f32, and the device assembly of the bit-exact operator file (RoIAlign: the kernel whose
    packed interpolation went wrong next to MFMA waves of another stream) really contains none."""
    import importlib.util
    import re
    import subprocess
    spec = importlib.util.spec_from_file_location('mh_build_flags', os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert not os.environ.get('MH_PACKED_F32'), 'an A/B build knob is set in the environment'
    for src in mod.SOURCES:
        has = '-packed-fp32-ops' in mod.compile_flags(src)
        assert has == (src not in mod.PACKED_OK), src
    assert set(mod.PACKED_OK) == {'conv.hip', 'pl_gemm.hip', 'pl_conv.hip'}
    out = str(tmp_path / 'exact_ops.s')
    subprocess.check_call([mod.HIPCC] + mod.compile_flags('exact_ops.hip') + ['-S', '--cuda-device-only', '-w', '-o', out,
                                                                             os.path.join(mod.HERE, 'exact_ops.hip')],
                          stderr=subprocess.DEVNULL)
    asm = open(out).read()
    assert 'roi_align_fwd_nhwc' in asm
    assert not re.search(r'\bv_pk_(add|mul|fma)_f32\b', asm)


def _device_asm(mod, src, tmp_path):
    import subprocess
    out = str(tmp_path / (src + '.s'))
    subprocess.check_call([mod.HIPCC] + mod.compile_flags(src) + ['-S', '--cuda-device-only', '-w', '-o', out, os.path.join(mod.HERE, src)],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_body(asm, mangled_fragment):
    """text of the first kernel whose symbol contains `mangled_fragment`, from its label to s_endpgm"""
    m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm' % re.escape(mangled_fragment), asm, flags=re.S | re.M)
    assert m, mangled_fragment
    return m.group(2)


def test_latency_critical_kernels_keep_their_load_structure(tmp_path):
    """What the measured speed of three kernels rests on is decided by the compiler, not by the source alone (DESIGN.md 5.3,
    7.4): the NMS sweep must issue its 24 mask reads per thread back to back (no branch / wait between them), the direct tower
    convolution must take its mask values through the scalar cache into SGPR operands and keep its 98 weights in registers
    (no scratch).  A compiler that decides otherwise makes them several times slower without failing a single numerics test."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('mh_build_flags2', os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = _device_asm(mod, 'exact_ops.hip', tmp_path)
    sweep = _kernel_body(asm, 'nms_sweep_kernel')
    runs = [len(r.split('global_load_dwordx2')) - 1 for r in re.split(r's_waitcnt vmcnt|s_cbranch|s_barrier', sweep)]
    assert max(runs) >= 24, 'the sweep no longer fetches a whole chunk (24 reads per thread) without a wait or branch in between: %s' % runs
    assert 'ds_or_b64' in sweep and 's_ff1_i32_b64' in sweep
    mask = _kernel_body(asm, 'nms_mask_kernel')
    assert 'v_readlane' not in mask and re.search(r's_load_dwordx(8|16)', mask)
    asm = _device_asm(mod, 'tower.hip', tmp_path)
    for frag, fmas in (('tower_conv1_fwd_kernel', 196), ('tower_conv1_wgrad_kernel', 196)):
        body = _kernel_body(asm, frag)
        assert len(re.findall(r'\bv_fma(c)?_f32', body)) >= fmas, frag
        assert re.search(r's_load_dwordx16', body), frag + ': mask values no longer come through the scalar cache'
        assert re.search(r'v_fmac_f32_e32 v\d+, s\d+, v\d+', body), frag + ': no SGPR operand in the FMAs'
    for name in ('tower_conv1_fwd_kernel', 'tower_conv1_wgrad_kernel'):
        m = re.search(r'\.amdhsa_kernel \w*%s\w*\n(.*?)\.end_amdhsa_kernel' % name, asm, flags=re.S)
        assert m and re.search(r'\.amdhsa_private_segment_fixed_size 0\b', m.group(1)), name + ' spills to scratch'
    # round 6, the same layer on the matrix cores: 7 k-tiles x 2 channel blocks x 3 f16 terms per 32-pixel tile in the forward kernel
    # (its weight fragments stay in registers: no scratch), 4 tap blocks x 2 channel blocks x 6 bf16 terms per output row in the
    # weight gradient; the window rows arrive as 16-byte / 8-byte loads, not as 14 scalar ones
    fwd = _kernel_body(asm, 'tower_conv1_mfma_fwd_kernel')
    assert len(re.findall(r'\bv_mfma_f32_32x32x16_f16\b', fwd)) == 42 and len(re.findall(r'\bglobal_load_dwordx4\b', fwd)) >= 7
    assert 'v_cvt_pk_f16_f32' in fwd and not re.search(r'\bv_cvt_pkrtz', fwd)            # round-to-nearest split
    wg = _kernel_body(asm, 'tower_conv1_mfma_wgrad_kernelILi2E')
    assert len(re.findall(r'\bv_mfma_f32_32x32x16_bf16\b', wg)) == 48 and 'ds_read_b128' in wg and 's_barrier' in wg
    for name in ('tower_conv1_mfma_fwd_kernel', 'tower_conv1_mfma_wgrad_kernelILi2E', 'tower_conv1_mfma_wgrad_kernelILi1E'):
        m = re.search(r'\.amdhsa_kernel \w*%s\w*\n(.*?)\.end_amdhsa_kernel' % name, asm, flags=re.S)
        assert m and re.search(r'\.amdhsa_private_segment_fixed_size 0\b', m.group(1)), name + ' spills to scratch'
