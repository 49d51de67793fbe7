"""`python bench.py --gpus N` starts its N ranks itself when torchrun's environment is absent (the driver may call it either
way; the reference's multi-GPU mode is one command, lib/rel_model.py:549-560).  CPU + gloo here; RCCL on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HIP_VISIBLE_DEVICES'] = ''            # the self-test must not depend on a device being present
    return env


def test_plain_invocation_starts_two_ranks_and_prints_one_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest'], env=_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split('\n') if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['sum_of_ranks_plus_1'] == 3.0 and d['backend'] == 'gloo'


def test_torchrun_invocation_is_taken_as_is():
    """started by torch.distributed.run (RANK / WORLD_SIZE set): no second level of processes"""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29631', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest']
    out = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split('\n') if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2


def test_a_group_of_another_size_is_refused():
    env = dict(_env(), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, text=True)
    assert out.returncode != 0 and 'asked for 2 ranks' in (out.stderr + out.stdout)


def test_missing_devices_fail_loudly():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, text=True)
    assert out.returncode != 0 and 'HIP device(s) visible' in (out.stderr + out.stdout)
    assert not [l for l in out.stdout.split('\n') if l.startswith('{')]


def test_dry_rehearsal_of_the_rank_launch():
    """`bench.py --gpus N --dry`: the rank command line, the REAL parameter list through the reducer's planning code on every rank
    (plans compared through the process group), the step's 25 collectives at their real sizes and order over gloo, host threads.
    N = 2 here (the 8-rank rehearsal is the same code: `python bench.py --gpus 8 --dry`, 36 s on 8 cores)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry'], env=_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split('\n') if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['dry'] and d['ok'] and d['n_gpus'] == 2 and all(d['checks'].values()), d['checks']
    cmd = d['rank_command']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--dry' not in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert d['rank_env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # the plan of the real cfg2 parameter list: fc6 (411 MB) of the two trainable heads in <= 64 MB row ranges, nothing larger
    assert d['collective_units'] == len(d['bucket_mb']) >= 20 and max(d['bucket_mb']) <= 64.0
    assert abs(sum(d['bucket_mb']) - d['gradient_mb']) < 0.5 and d['gradient_mb'] > 1000
    assert d['lr'] == 1e-3 * 2 * 6
