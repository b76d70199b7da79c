import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes ~1 min on 8 CPU cores")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- the two-process sharded run on the GPU box (tests/test_pipeline_gpu.py::test_two_process_sharded_sequence).  Its processes must be
# started from a process that has NOT initialised the GPU, so the helper is launched here, before the first test runs, in the background;
# the test waits for it.  Only when the gpu tests are selected and a device exists (counting devices does not initialise one).
TWO_PROC = {"proc": None, "dir": os.path.join(ROOT, "gpurun_out", "two_process"), "log": None}


def pytest_collection_finish(session):
    # (after deselection: the helper is started only when the test that waits for it is going to run -- a `-k` subset that leaves it out
    # does not pay for two extra GPU processes; round-5 advisor)
    if not any(item.name == "test_two_process_sharded_sequence" for item in session.items) or TWO_PROC["proc"] is not None:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import shutil
    import subprocess
    shutil.rmtree(TWO_PROC["dir"], ignore_errors=True)
    os.makedirs(TWO_PROC["dir"], exist_ok=True)
    TWO_PROC["log"] = open(os.path.join(TWO_PROC["dir"], "log.txt"), "w")
    # its own session = its own process group: the helper's mp.spawn workers are its children and go down with it (sessionfinish)
    TWO_PROC["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_process_shard.py"), TWO_PROC["dir"]],
                                        stdout=TWO_PROC["log"], stderr=subprocess.STDOUT, cwd=ROOT, start_new_session=True)
    # Round 6: the helper runs to its end BEFORE the first test touches the GPU (~40 s).  Rounds 4-5 let it run beside the suite; with the
    # calibration now ~90 forwards per process that overlap cost the test its meaning twice in four runs: beside other processes allocating on the
    # same GPU a forward could come out with other bits than its rerun, and one such forward inside a calibration moves a borderline decision -- the
    # three processes then run different arithmetic and cannot be bit-equal.  It was located later in the round (the log-binomial kernel's 4- / 8-byte
    # LDS gathers beside another stream's MFMA kernel; the form that ships reads 16 bytes and has not shown it); the order is kept: one process per GPU is how the path is deployed, and the suite's own tests are not
    # what this test is about.  BODYSLAM_TEST_HELPER_CONCURRENT=1: the rounds 4-5 behaviour.
    if os.environ.get("BODYSLAM_TEST_HELPER_CONCURRENT") == "1":
        return
    try:
        TWO_PROC["proc"].wait(timeout=600)
    except subprocess.TimeoutExpired:
        pass                                   # (the test reports the hang)


def pytest_sessionfinish(session, exitstatus):
    p = TWO_PROC["proc"]
    if p is not None and p.poll() is None:
        import signal
        try:
            os.killpg(p.pid, signal.SIGKILL)        # the launcher AND its workers (exactly the group started above)
        except (ProcessLookupError, PermissionError):
            p.kill()
    if TWO_PROC["log"] is not None:
        TWO_PROC["log"].close()


@pytest.fixture(scope="session")
def two_process_run():
    return TWO_PROC
