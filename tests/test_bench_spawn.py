"""CPU-only: ``python bench.py --gpus N`` (N > 1) started as ONE process must launch its N ranks itself -- as fresh child
processes under torch.distributed.run, before this process imports the product package or touches a GPU -- and a rank
started by that launcher (WORLD_SIZE in the environment) must not spawn again."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_spawns_its_ranks(monkeypatch):
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '5', '--warmup', '2'])
    loaded_before = 'robust_e2e_gan_amd.joint_train' in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # exits with the launcher's code
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '5', '--warmup', '2'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert ('robust_e2e_gan_amd.joint_train' in sys.modules) == loaded_before     # spawned before the product was imported


def test_rank_process_does_not_spawn_again(monkeypatch):
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    monkeypatch.setattr(subprocess, 'call', lambda *a, **k: pytest.fail('a rank must not spawn'))
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    from robust_e2e_gan_amd import dist as rdist
    monkeypatch.setattr(rdist, 'init_from_env', lambda: (0, 2, 0))
    with pytest.raises(AssertionError, match='needs a GPU'):   # gets as far as the GPU check (none here), without spawning
        bench.main()
