"""Global wall-clock timer pair, lib/utils/tictoc.py:3-14."""
import time

_start = None


def tic():
    global _start
    _start = time.time()
    return _start


def toc():
    if _start is None:
        return None
    return time.time() - _start
