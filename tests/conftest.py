import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- the 2-rank data-parallel GPU test needs two FRESH processes on the GPU.  They are started here, at session start,
# before this pytest process has initialised the GPU (a process that has must not exec another program on the GPU pool);
# they run beside the other GPU tests and tests/test_ddp_gpu.py collects their results.
_DDP = {"workers": None}


def _gpu_selected(config):
    m = config.getoption("-m") or ""
    return "gpu" in m and "not gpu" not in m


def pytest_collection_finish(session):
    """Runs after collection and before any test (so before this process touches the GPU).  The workers are only started
    when tests/test_ddp_gpu.py is among the selected items: a ``-k`` subset without it runs with the GPU to itself."""
    if not _gpu_selected(session.config) or os.environ.get("CRCT_NO_DDP_WORKERS") or _DDP["workers"]:
        return
    if not any(os.path.basename(str(it.fspath)) == "test_ddp_gpu.py" for it in session.items):
        return
    try:
        import torch
        n_gpus = torch.cuda.device_count()         # counting devices does not initialise the GPU
        if n_gpus < 1:
            return
    except Exception:
        return
    import socket
    import subprocess
    import tempfile
    outdir = tempfile.mkdtemp(prefix="crct_ddp_")
    procs = {}
    cases = ("tiny", "full") + (("tiny_rccl", "full_rccl") if n_gpus >= 2 else ())      # RCCL between two processes needs two GPUs
    for case in cases:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs[case] = []
        for rank in range(2):
            log_path = os.path.join(outdir, "%s_rank%d.log" % (case, rank))
            log = open(log_path, "w")
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
            argv = [sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(rank), "2", str(port), outdir, case.split("_")[0]]
            if case.endswith("_rccl"):
                argv.append("nccl")
            pr = subprocess.Popen(argv, stdout=log, stderr=subprocess.STDOUT, env=env)
            pr.log_path = log_path
            procs[case].append(pr)
    _DDP["workers"] = (outdir, procs)


def pytest_sessionfinish(session, exitstatus):
    if _DDP["workers"]:
        for prs in _DDP["workers"][1].values():
            for pr in prs:
                if pr.poll() is None:
                    pr.kill()


@pytest.fixture(scope="session")
def ddp_workers():
    return _DDP["workers"]
