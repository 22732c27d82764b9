import faulthandler
import os
import signal
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

# Every test runs under a time limit: a hung kernel (the one GPU-only bug of round 4 was a hang in the search kernel's fetch state
# machine) must fail ITS test within minutes instead of burning the whole `pytest -x` step.  The library has its own device-side
# bound (GM_ERR_INTERNAL after SearchArgs::iterCap wavefront iterations, gm_kernels.h); this is the belt to those braces.
# SIGALRM interrupts the main thread even inside a ctypes call that waits on the device (the wait is a futex/ioctl: EINTR) -- and
# if the interpreter cannot get control back, faulthandler's watchdog thread dumps every stack and exits the process.
DEFAULT_LIMIT_S = int(os.environ.get("GM_TEST_TIMEOUT", "600"))
HARD_EXTRA_S = 60


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "time_limit(seconds): per-test wall-clock limit (default GM_TEST_TIMEOUT or 600 s)")


class TestTimeout(Exception):
    pass


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    m = item.get_closest_marker("time_limit")
    limit = int(m.args[0]) if m and m.args else DEFAULT_LIMIT_S
    if limit <= 0 or not hasattr(signal, "SIGALRM"):
        yield
        return

    def on_alarm(signum, frame):
        raise TestTimeout(f"{item.nodeid} exceeded its time limit of {limit} s")

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(limit)
    faulthandler.dump_traceback_later(limit + HARD_EXTRA_S, exit=True)   # the alarm could not be delivered: dump all stacks, exit
    try:
        yield
    finally:
        signal.alarm(0)
        faulthandler.cancel_dump_traceback_later()
        signal.signal(signal.SIGALRM, old)
