import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
os.environ.setdefault('RE2E_DEBUG_HOOKS', '1')      # re2e_debug_force_abort / re2e_debug_occupy answer only with this set (csrc/lstm.hip)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def host_threads():
    """Threads for the CPU oracle: scheduler affinity capped by the cgroup CPU quota.  The GPU box shows 256 logical cores to a
    container whose quota is 16: sizing torch's pool by the affinity alone oversubscribes the quota 4-16x and the full-size oracle
    steps take several times longer (round 3: 182 s / 484 s for the config-4 / config-5 step with 64 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, 64))
