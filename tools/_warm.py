"""Clock warm-up for the GPU benches.

An MI355X box that has idled for the second or so a bench spends drawing its input on the host is back in a low-power
state (rocm-smi says so on every fresh box), and its clocks take tens of milliseconds of work to come back: the same launch
measured 1.08, 0.99, 0.97, 0.95, 0.93, 0.92 ms in six consecutive groups of five (profiles/r04_clock_ramp.txt).  Timing five
launches right after the upload therefore measures the ramp, not the kernel.  `warm(run, sync)` repeats the launch untimed
until `seconds` have passed; the timed launches that follow see the clocks a long-running job sees.
"""
import time


def warm(run, sync, seconds=0.3):
    """Call run() (asynchronous launches) untimed for about `seconds`; returns the number of calls."""
    t0 = time.perf_counter()
    calls = 0
    while True:
        for _ in range(8):
            run()
        calls += 8
        sync()
        if time.perf_counter() - t0 >= seconds:
            return calls
