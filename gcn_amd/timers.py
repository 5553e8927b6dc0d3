"""Host / device timers with the surface of the reference's pygcn/perf/dmk.py (Timer :44-68,
CTimer :71-117, HCTimer :32-42, Timers registry :16-30): ``with timers.hc.af: ...`` around the
SpMM (gcn6.py:135,186), ``.avms()`` for the per-layer print (gcn6.py:401-410).  Device time comes
from HIP events on the current stream (torch.cuda.Event on ROCm), harvested lazily so that
timing never synchronises the stream inside the loop."""
import time

import torch


class Timer:
    def __init__(self):
        self.dur_ns, self.t_start, self.n_calls = 0, None, 0

    def reset(self):
        self.dur_ns, self.t_start, self.n_calls = 0, None, 0

    def start(self):
        self.t_start = time.perf_counter_ns()
        return self

    def stop(self):
        assert self.t_start is not None
        self.dur_ns += time.perf_counter_ns() - self.t_start
        self.t_start = None
        self.n_calls += 1
        return self

    __enter__ = lambda self: self.start()

    def __exit__(self, *exc):
        self.stop()

    def ns(self): return self.dur_ns
    def us(self): return self.dur_ns * 1e-3
    def ms(self): return self.dur_ns * 1e-6
    def s(self): return self.dur_ns * 1e-9
    def avns(self): return self.dur_ns / self.n_calls if self.n_calls else 0
    def avus(self): return self.avns() * 1e-3
    def avms(self): return self.avns() * 1e-6
    def avs(self): return self.avns() * 1e-9


class CTimer:
    """Device timer: a (start, stop) HIP-event pair per interval, summed when the stop event has
    completed (non-blocking harvest, dmk.py:103-111)."""

    def __init__(self, on=True):
        self.dur_ms, self.n_calls = 0.0, 0
        self.now, self.pending, self.free = None, [], []
        self.on = on and torch.cuda.is_available()

    def reset(self):
        self.harvest(True)
        self.dur_ms, self.n_calls = 0.0, 0

    def start(self):
        assert self.now is None
        # (under HIP-graph capture events can neither be queried nor timed: the interval is counted, not measured)
        if not self.on or torch.cuda.is_current_stream_capturing():
            self.now = ()
            return self
        self.harvest()
        self.now = self.free.pop() if self.free else (torch.cuda.Event(enable_timing=True),
                                                      torch.cuda.Event(enable_timing=True))
        self.now[0].record()
        return self

    def stop(self):
        assert self.now is not None
        self.n_calls += 1
        if self.on and self.now:
            self.now[1].record()
            self.pending.append(self.now)
        self.now = None
        return self

    def harvest(self, all=False):
        while self.pending and (all or self.pending[0][1].query()):
            a, b = self.pending.pop(0)
            if all:
                b.synchronize()
            self.dur_ms += a.elapsed_time(b)
            self.free.append((a, b))

    __enter__ = lambda self: self.start()

    def __exit__(self, *exc):
        self.stop()

    def ms(self):
        self.harvest(True)
        return self.dur_ms

    def s(self): return self.ms() * 1e-3
    def us(self): return self.ms() * 1e3
    def avms(self): return self.ms() / self.n_calls if self.n_calls else 0
    def avs(self): return self.avms() * 1e-3
    def avus(self): return self.avms() * 1e3


class HCTimer:
    def __init__(self, h=None, c=None):
        self.h, self.c = h or Timer(), c or CTimer()

    def reset(self):
        self.h.reset(); self.c.reset()

    def __enter__(self):
        self.h.start(); self.c.start()

    def __exit__(self, *exc):
        self.c.stop(); self.h.stop()


class _Ns:
    def __init__(self, get): self._get = get
    def __getattr__(self, name): return self._get(name)


class Timers:
    """``t = Timers(); with t.hc.af: ...; t.h.af.avms(); t.c.af.avms()`` (dmk.py:16-30)."""

    def __init__(self):
        self._h, self._c = {}, {}
        self.h = _Ns(lambda n: self._h.setdefault(n, Timer()))
        self.c = _Ns(lambda n: self._c.setdefault(n, CTimer()))
        self.hc = _Ns(lambda n: HCTimer(self._h.setdefault(n, Timer()), self._c.setdefault(n, CTimer())))

    def reset(self):
        for t in list(self._h.values()) + list(self._c.values()):
            t.reset()
