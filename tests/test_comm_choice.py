"""CPU: the choice between the library-side RCCL collectives and the all-reduce callback (engine.EngineNet._install_comm) is itself
collective -- every rank takes the library-side path only if EVERY rank bound librccl and holds two communicators that passed the
known-answer check; otherwise every rank takes the callback (or, with DRS_COMM=rccl, every rank raises).  Driven here with a stand-in
for the process group and for the library, one "rank" at a time; the real thing needs a multi-GPU node."""
import types

import pytest

from drs_amd import engine


class FakeComm(object):
    backend = "nccl"

    def __init__(self, world, rank, others_ok=True):
        self.world, self.rank, self.others_ok = world, rank, others_ok
        self.asked = []

    def all_true(self, flag):                 # MIN over the ranks of "my flag"
        self.asked.append(bool(flag))
        return bool(flag) and self.others_ok


def make_net(comm, rccl_raises=None):
    calls = []
    net = types.SimpleNamespace(comm=comm, h=object(), collectives=None, _rccl=["small", "big"])

    def install_rccl():
        calls.append("rccl")
        if rccl_raises:
            raise rccl_raises
    net._install_rccl = install_rccl
    net._install_callback = lambda: calls.append("callback")
    return net, calls


@pytest.fixture
def lib(monkeypatch):
    state = dict(available=1, set_rccl=0)
    monkeypatch.setattr(engine._lib, "query", lambda name, *a: state["available"] if name == "drs_rccl_available" else 0)

    def call(name, *a):
        assert name == "drs_net_set_rccl"
        state["set_rccl"] += 1
    monkeypatch.setattr(engine._lib, "call", call)
    monkeypatch.delenv("DRS_COMM", raising=False)
    return state


def test_every_rank_fine_takes_the_library_path(lib):
    net, calls = make_net(FakeComm(8, 3))
    engine.EngineNet._install_comm(net)
    assert calls == ["rccl"] and lib["set_rccl"] == 1 and net.collectives.startswith("rccl (asynchronous")      # two communicators were made
    net, calls = make_net(FakeComm(8, 3))
    net._rccl = ["one"]                                # the default: one communicator, the inline form
    engine.EngineNet._install_comm(net)
    assert net.collectives.startswith("rccl (inline")
    assert net.comm.asked == [True, True]          # two collective questions: bound everywhere?  working everywhere?


def test_a_failure_on_another_rank_sends_this_rank_to_the_callback_too(lib):
    net, calls = make_net(FakeComm(8, 3, others_ok=False))
    engine.EngineNet._install_comm(net)
    assert "callback" in calls and lib["set_rccl"] == 0


def test_a_failed_known_answer_check_here_is_reported_to_the_others_and_falls_back(lib):
    net, calls = make_net(FakeComm(2, 1), rccl_raises=engine._lib.DrsError("library-side RCCL all-reduce at world 2 returned ..."))
    engine.EngineNet._install_comm(net)
    assert calls == ["rccl", "callback"] and lib["set_rccl"] == 0
    assert net.comm.asked == [True, False]          # the others learn of it through the second question


def test_librccl_not_bound_skips_the_communicators_but_still_answers_both_questions(lib):
    lib["available"] = 0
    net, calls = make_net(FakeComm(4, 0))
    engine.EngineNet._install_comm(net)
    assert calls == ["callback"] and net.comm.asked == [False, False]


def test_forced_library_path_raises_instead_of_falling_back(lib, monkeypatch):
    monkeypatch.setenv("DRS_COMM", "rccl")
    net, calls = make_net(FakeComm(2, 0, others_ok=False))
    with pytest.raises(engine._lib.DrsError):
        engine.EngineNet._install_comm(net)
    assert "callback" not in calls


def test_torch_choice_and_gloo_take_the_callback_without_asking(lib, monkeypatch):
    monkeypatch.setenv("DRS_COMM", "torch")
    net, calls = make_net(FakeComm(2, 0))
    engine.EngineNet._install_comm(net)
    assert calls == ["callback"] and net.comm.asked == []
    monkeypatch.delenv("DRS_COMM")
    c = FakeComm(2, 0)
    c.backend = "gloo"
    net, calls = make_net(c)
    engine.EngineNet._install_comm(net)
    assert calls == ["callback"] and c.asked == []


def test_a_collective_call_that_never_returns_becomes_an_error_on_this_rank():
    """drs_rccl_comm_create is ncclCommInitRank: collective.  If another rank never gets there, this rank's call never returns; the
    helper turns that into an exception after a rank-local limit so that _install_comm reaches its collective question and every rank
    falls back together (`cli.main*` under torch.distributed.run would hang)."""
    import threading
    import time
    assert engine.call_with_timeout(lambda: 7, 5.0, "quick") == 7
    with pytest.raises(ZeroDivisionError):
        engine.call_with_timeout(lambda: 1 // 0, 5.0, "raises")
    gate = threading.Event()
    t0 = time.time()
    with pytest.raises(engine._lib.DrsError, match="did not return within"):
        engine.call_with_timeout(gate.wait, 0.3, "stuck")
    assert time.time() - t0 < 3.0
    gate.set()
