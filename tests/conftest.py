import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Import torch before any test timer runs, with a heartbeat on the real stderr.  On a fresh box the first
    `import torch` pages the image in and can take minutes; inside a test that looks like a hang (a full GPU run was
    once killed for 7 minutes of silence at the first torch-importing test), here it is visible and untimed."""
    import threading
    import time

    capman = session.config.pluginmanager.getplugin("capturemanager")
    if capman is not None:
        capman.suspend_global_capture(in_=True)
    done = threading.Event()

    def beat():
        t0 = time.time()
        while not done.wait(20):
            sys.__stderr__.write("[conftest] importing torch ... %d s\n" % (time.time() - t0))
            sys.__stderr__.flush()

    th = threading.Thread(target=beat, daemon=True)
    th.start()
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    finally:
        done.set()
        th.join()
        if capman is not None:
            capman.resume_global_capture()


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)
