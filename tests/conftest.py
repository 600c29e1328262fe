import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oraclebind
    oraclebind.build()
    return oraclebind


# ---- a helper process for the tests that start OTHER programs (tests/test_gpu_multiprocess.py: torch.distributed.run).  On the GPU pool a process that has initialised the
# GPU must not exec another program, and a forked child of such a process inherits that state: the helper is created HERE, when pytest loads this file -- before any test has
# imported torch or the HIP library --, never touches the GPU itself, and starts what it is asked to start.
_HELPER = None
_HELPER_CODE = r"""
import json, subprocess, sys
for line in sys.stdin:
    req = json.loads(line)
    try:
        p = subprocess.run(req["cmd"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=req["timeout"])
        rc, out = p.returncode, p.stdout.decode(errors="replace")
    except subprocess.TimeoutExpired as e:
        rc, out = 124, (e.stdout or b"").decode(errors="replace") + "\n(timeout)"
    sys.stdout.write(json.dumps({"rc": rc, "out": out}) + "\n")
    sys.stdout.flush()
"""


def _start_helper():
    global _HELPER
    if _HELPER is None:
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _HELPER = subprocess.Popen([sys.executable, "-c", _HELPER_CODE], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, text=True)
    return _HELPER


def spawn_helper_run(cmd, timeout=600):
    import json
    h = _start_helper()
    h.stdin.write(json.dumps({"cmd": cmd, "timeout": timeout}) + "\n")
    h.stdin.flush()
    ans = json.loads(h.stdout.readline())
    return ans["rc"], ans["out"]


def pytest_sessionstart(session):
    if session.config.getoption("-m") and "gpu" in session.config.getoption("-m") and "not gpu" not in session.config.getoption("-m"):
        _start_helper()


def pytest_sessionfinish(session, exitstatus):
    global _HELPER
    if _HELPER is not None:
        try:
            _HELPER.stdin.close()
            _HELPER.wait(timeout=10)
        except Exception:   # noqa: BLE001
            _HELPER.kill()
        _HELPER = None
