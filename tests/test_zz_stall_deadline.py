"""The stall watchdog of tools/abi_allgather_check.py, exercised on the CPU.

This file sorts LAST in the CPU suite on purpose: in this container, running the test in the middle of the suite left every later
torch CPU matmul of the pytest process ~80x slower (the suite went from 30 s to 9 min; the trigger is the pair of short-lived child
processes, the mechanism was not found in the time given to it and does not reproduce outside pytest) - at the end nothing runs after it."""
import os

from conftest import ROOT


def test_stall_deadline_fires_while_the_main_thread_is_blocked_inside_c():
    """tools/abi_allgather_check.py's hard deadline (the backstop of the one path that has never run with N > 1 ranks): a child arms it
    for one second, names its step and then blocks in libc sleep(30) through ctypes - where no Python signal handler could run.  The
    watchdog thread must end the child with exit code 3 well inside the sleep and say which step it stalled in; a disarmed one must not fire."""
    import subprocess
    import sys
    import time
    code = ("import ctypes, sys; sys.path.insert(0, %r); import abi_allgather_check as a; "
            "a.STEP[0] = 'blocked in C'; d = a.arm_deadline(1.0); "
            "sys.argv[1] == 'disarm' and d(); ctypes.CDLL(None).sleep(int(sys.argv[2])); sys.exit(0)") % os.path.join(ROOT, "tools")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code, "armed", "30"], capture_output=True, text=True, timeout=25)
    assert r.returncode == 3 and time.monotonic() - t0 < 15, (r.returncode, r.stderr)
    assert "stalled in step 'blocked in C'" in r.stderr
    r = subprocess.run([sys.executable, "-c", code, "disarm", "2"], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0 and "stalled" not in r.stderr, (r.returncode, r.stderr)
