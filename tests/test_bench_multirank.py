"""bench.py's multi-rank failure handling on CPU: two gloo ranks run bench.attach_with_agreement -- the function
bench.run uses to attach the library's RCCL transport -- with injected failures.  No GPU, no RCCL: `attach` / `prove` are the
injected stand-ins, the agreement all-reduce and the watchdog are the real ones.

  * attach fails on ONE rank   -> BOTH ranks fall back to the torch transport together (no hang), destroy() ran on both,
                                   and with --require-rccl the exit code is 4 on every rank;
  * nothing fails              -> both report ("rccl", world), exit code 0;
  * a proving evaluation fails on one rank -> that rank exits 5 at once and its peer, left inside the exchange, is ended by the
                                   watchdog with 6: every rank exits non-zero within the watchdog's bound, none re-executes anything.
"""
import os
import socket
import sys
import time

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, ret):
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"destroy": 0}

    def attach():
        if mode == "attach_fails_on_rank1" and rank == 1:
            raise RuntimeError("injected: ncclCommInitRank failed")
        return world

    def destroy():
        calls["destroy"] += 1

    def prove():
        if mode == "prove_fails_on_rank1":
            if rank == 1:
                raise RuntimeError("injected: evaluation over RCCL failed")
            time.sleep(30)          # the peer sits in a receive that never completes; the watchdog (2 s) ends it

    def _exit(code):
        ret[rank] = ("exit", code)
        os._exit(code)

    t0 = time.time()
    transport, n, note = bench.attach_with_agreement(attach, destroy, prove, dist, "cpu", rank, watchdog_s=2.0, _exit=_exit)
    ret[rank] = (transport, n, bool(note), calls["destroy"], bench.exit_code(world, True, transport, n),
                 bench.exit_code(world, False, transport, n), time.time() - t0)
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, join=True):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.spawn(_worker, args=(2, port, mode, ret), nprocs=2, join=False)
    t0 = time.time()
    codes = None
    while time.time() - t0 < 60:
        if all(not p.is_alive() for p in ctx.processes):
            codes = [p.exitcode for p in ctx.processes]
            break
        time.sleep(0.1)
    for p in ctx.processes:        # (exact PIDs we started; never a pattern)
        if p.is_alive():
            p.terminate()
    assert codes is not None, "ranks did not finish: the failure handling hangs"
    return dict(ret), codes, time.time() - t0


def test_attach_failure_on_one_rank_makes_every_rank_fall_back_together():
    ret, codes, dt = _run("attach_fails_on_rank1")
    assert codes == [0, 0]
    for r in (0, 1):
        transport, n, has_note, destroyed, code_req, code_plain, took = ret[r]
        assert transport == "torch" and n == 0 and has_note and destroyed == 1
        assert code_req == 4 and code_plain == 0       # --require-rccl turns the fallback into exit code 4
        assert took < 10


def test_clean_attach_reports_rccl_on_every_rank():
    ret, codes, dt = _run("none")
    assert codes == [0, 0]
    for r in (0, 1):
        transport, n, has_note, destroyed, code_req, code_plain, took = ret[r]
        assert transport == "rccl" and n == 2 and not has_note and destroyed == 0 and code_req == 0 and code_plain == 0


def test_failed_proving_evaluation_ends_every_rank_nonzero_within_the_watchdog():
    ret, codes, dt = _run("prove_fails_on_rank1")
    assert ret[1] == ("exit", 5)                        # the failing rank says why and leaves at once
    assert ret[0] == ("exit", 6)                        # its peer is ended by the watchdog, not by a hang
    assert codes[0] == 6 and codes[1] == 5 and dt < 30
