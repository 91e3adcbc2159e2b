import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _heartbeat():
    """A cold `import torch` on a fresh GPU box can take minutes, and pytest captures the tests' output: keep a
    sign of life under gpurun_out/ (and on the real stderr) so that a slow start is not taken for a hang."""
    import threading
    import time

    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "pytest_heartbeat.log")
    except OSError:
        path = None
    err = os.fdopen(os.dup(2), "w", buffering=1)
    t0 = time.time()

    def beat():
        while True:
            time.sleep(45)
            msg = "[heartbeat] pytest alive, %d s\n" % (time.time() - t0)
            try:
                if path:
                    with open(path, "a") as fh:
                        fh.write(msg)
                err.write(msg)
            except OSError:
                pass

    threading.Thread(target=beat, daemon=True).start()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        _heartbeat()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
