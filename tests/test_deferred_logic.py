"""CPU tier: the ticket bookkeeping of compute_one_deferred / collect / set_deferred_depth in the Python mirror (engine.py), on engines
whose device side is replaced by a recorder -- which engine a pair runs on, what a change of depth or of a setting does to pairs in
flight, which tickets stay collectable.  (The same rules with real engines: tests/test_gpu_parity.py, tests/soak/deferred_soak.py;
the C++ twin host/turbo_metrics.cpp is exercised through the CLI's --loop deferred --in-flight N.)"""
import pytest

from tm_pkg import tm


class _Lib:
    def __getattr__(self, name):
        return lambda *a: 0


@pytest.fixture
def fake(monkeypatch):
    made = []

    def init(self, width, height, metrics, batch=1, full_sums=False):
        self._L, self._h, self._keep = _Lib(), object(), {}
        self.width, self.height, self.batch, self._metrics = width, height, batch, metrics
        self.launched, self.closed, self.mode = [], False, "pooled"
        self._cur = self._fly = self._res = None
        made.append(self)

    def set_pair(self, slot, fref, fdis):
        assert slot == 0 and self._fly is None, "a pair handed over while this engine's launch is still in flight"
        self._cur = (fref, fdis)

    def compute_async(self, n=1):
        self._fly = (self._cur, self.mode)
        self.launched.append(self._cur)

    def sync(self):
        if self._fly is not None:
            self._res, self._fly = self._fly, None

    def scores(self, slot):
        if self._res is None:
            raise tm.TmError(tm.ffi.TM_ERR_STATE, "no results")
        return ("scores of", self._res)

    def close(self):
        d = getattr(self, "_def", None)
        if d is not None:
            self._def = None
            for p in d["peers"]:
                p.close()
        self.closed = True

    real_mode = tm.TurboMetrics.set_channel_mode

    def set_channel_mode(self, first):
        real_mode(self, first)  # (retires the pairs in flight, records the setting for engines made later, tells the peers)
        self.mode = "first" if first else "pooled"

    for name, fn in dict(__init__=init, set_pair=set_pair, compute_async=compute_async, sync=sync, scores=scores, close=close,
                         set_channel_mode=set_channel_mode).items():
        monkeypatch.setattr(tm.TurboMetrics, name, fn)
    monkeypatch.setattr(tm.TurboMetrics, "__del__", lambda self: None, raising=False)
    return made


def test_tickets_take_turns_on_as_many_engines_as_the_depth(fake):
    eng = tm.TurboMetrics(64, 64, tm.Metrics(ssimulacra2=True), batch=1)
    t = [eng.compute_one_deferred("r%d" % k, "d%d" % k) for k in range(2)]
    assert len(fake) == 2                                      # the second engine came with the second ticket
    assert fake[0].launched == [("r0", "d0")] and fake[1].launched == [("r1", "d1")]
    t.append(eng.compute_one_deferred("r2", "d2"))             # a third submission finishes the oldest pair first and keeps its scores
    assert fake[0].launched == [("r0", "d0"), ("r2", "d2")]
    assert [eng.collect(x) for x in reversed(t)] == [("scores of", (("r%d" % k, "d%d" % k), "pooled")) for k in (2, 1, 0)]
    with pytest.raises(tm.TmError):
        eng.collect(t[0])                                      # once
    with pytest.raises(tm.TmError):
        eng.collect(99)
    eng.set_deferred_depth(4, create_now=True)
    assert len(fake) == 4
    t = [eng.compute_one_deferred(k, k) for k in range(4)]
    assert [len(e.launched) for e in fake] == [3, 2, 1, 1]   # tickets 3 .. 6 on engines 3, 0, 1, 2
    assert all(e._fly is not None for e in fake)               # four in flight, nothing waited for
    assert [eng.collect(x)[1][0] for x in t] == [(k, k) for k in range(4)]


def test_a_change_of_depth_or_of_a_setting_keeps_what_is_in_flight_collectable(fake):
    eng = tm.TurboMetrics(64, 64, tm.Metrics(ssimulacra2=True), batch=1)
    eng.set_deferred_depth(3)
    old = [eng.compute_one_deferred(k, k) for k in range(3)]
    assert len(fake) == 3
    eng.set_deferred_depth(2)                                  # finishes the three, frees the third engine
    assert fake[2].closed and not fake[1].closed
    new = [eng.compute_one_deferred(10 + k, 10 + k) for k in range(2)]
    eng.set_channel_mode(True)                                 # the pairs in flight are scored with the mode they were submitted under
    assert all(e.mode == "first" for e in fake[:2])
    late = eng.compute_one_deferred(20, 20)
    assert [eng.collect(x)[1] for x in new] == [((10, 10), "pooled"), ((11, 11), "pooled")]
    assert eng.collect(late)[1] == ((20, 20), "first")
    assert [eng.collect(x)[1][0] for x in old] == [(0, 0), (1, 1), (2, 2)]
    eng.set_deferred_depth(5)
    eng.compute_one_deferred(30, 30); eng.compute_one_deferred(31, 31); eng.compute_one_deferred(32, 32)
    assert all(e.mode == "first" for e in fake if not e.closed)  # engines made later start with the settings made before them
    for bad in (1, 9, 0):
        with pytest.raises(tm.TmError):
            eng.set_deferred_depth(bad)
    assert eng.compute_one(40, 40)[1] == ((40, 40), "first")   # the blocking call beside it: what is in flight is finished and kept
    eng.close()
    assert all(e.closed for e in fake)


def test_deferred_needs_a_one_pair_engine(fake):
    eng = tm.TurboMetrics(64, 64, tm.Metrics(ssimulacra2=True), batch=4)
    with pytest.raises(ValueError):
        eng.compute_one_deferred(0, 0)
    with pytest.raises(ValueError):
        eng.set_deferred_depth(3, create_now=True)
