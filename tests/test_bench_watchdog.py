"""bench.py's N > 1 step watchdog: a phase that makes no progress for longer than the limit ends the process with
a non-zero status (os._exit — never an exec of a process that has touched the GPU), naming the phase; a run that keeps beating, or
one that stops the watchdog, is left alone."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import sys, time
sys.path.insert(0, {root!r})
import bench
bench.WATCHDOG.start(0.6, 1)
bench.WATCHDOG.beat("timed loop")
mode = sys.argv[1]
t0 = time.monotonic()
while time.monotonic() - t0 < 3.0:
    time.sleep(0.05)
    if mode == "beats":
        bench.WATCHDOG.beat("still going")
    if mode == "stopped":
        bench.WATCHDOG.stop()
print("finished")
"""


def _run(mode):
    return subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT), mode], capture_output=True, text=True, timeout=120)


def test_watchdog_exits_nonzero_on_a_stalled_phase():
    p = _run("stalls")
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    assert "watchdog: rank 1 made no progress" in p.stderr and "'timed loop'" in p.stderr
    assert "finished" not in p.stdout


def test_watchdog_leaves_a_progressing_or_finished_run_alone():
    for mode in ("beats", "stopped"):
        p = _run(mode)
        assert p.returncode == 0 and "finished" in p.stdout, (mode, p.returncode, p.stderr[-500:])
