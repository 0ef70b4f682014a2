import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

DDP_JOBS = {}       # mode -> dict(procs=[Popen], outdir=str, logs=[path])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def pytest_sessionstart(session):
    """The data-parallel HIP-path check needs fresh rank processes.  They are started here, before this process has
    touched the GPU (a GPU-initialised process must not start other programs on this pool), run beside the other GPU
    tests and are collected by tests/test_ddp_hip.py."""
    markexpr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr or os.environ.get("SMIN_SKIP_DDP_TEST"):
        return
    import torch
    if torch.cuda.device_count() < 1:                      # device_count() does not initialise the GPU
        return
    outdir = tempfile.mkdtemp(prefix="smin_ddp_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    worker = os.path.join(ROOT, "tests", "ddp_hip_worker.py")
    for mode, world in (("gloo2", 2), ("rccl1", 1)):
        port = _free_port()
        procs, logs = [], []
        for rank in range(world):
            log = os.path.join(outdir, f"{mode}_rank{rank}.log")
            procs.append(subprocess.Popen([sys.executable, worker, mode, str(rank), str(world), port, outdir], env=env, cwd=ROOT,
                                          stdout=open(log, "w"), stderr=subprocess.STDOUT))
            logs.append(log)
        DDP_JOBS[mode] = dict(procs=procs, outdir=outdir, logs=logs, world=world)


def pytest_sessionfinish(session, exitstatus):
    for job in DDP_JOBS.values():
        for p in job["procs"]:
            if p.poll() is None:
                p.kill()                                   # exact children started above


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
