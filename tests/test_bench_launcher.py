"""bench.py's own N > 1 launcher (`python bench.py --gpus N` without torchrun): every rank is watched - one that dies takes the job down
at once, with its stderr tail and a non-zero exit code, instead of leaving the others in a collective until somebody's timeout."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STUB = '''
import os, sys, time
rank = int(os.environ["RANK"])
assert os.environ["WORLD_SIZE"] == sys.argv[1] and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[2]
if mode == "dies" and rank == 2:
    time.sleep(0.5)
    sys.stderr.write("rank 2: simulated RCCL initialisation failure\\n")
    sys.exit(7)
if mode == "dies":
    open(os.path.join(sys.argv[3], "alive%d" % rank), "w").write(str(os.getpid()))
    time.sleep(600)          # "waiting in a collective"
if rank == 0:
    print('{"ok": true}')
sys.stderr.write("rank %d done\\n" % rank)
'''


def _launch(tmp_path, mode, capsys):
    import bench
    script = tmp_path / "stub_rank.py"
    script.write_text(STUB)
    t0 = time.time()
    rc = bench.self_launch(types.SimpleNamespace(gpus=4), ["4", mode, str(tmp_path)], script=str(script), grace_s=5.0)
    return rc, time.time() - t0, capsys.readouterr()


def test_dead_rank_ends_the_job(tmp_path, capsys):
    rc, dt, io = _launch(tmp_path, "dies", capsys)
    assert rc == 7 and dt < 30.0
    assert "rank 2 of 4 exited with code 7" in io.err and "simulated RCCL initialisation failure" in io.err
    assert io.out == ""                                   # no JSON line from a job that failed
    time.sleep(0.2)
    for r in (0, 1, 3):                                   # the surviving ranks were terminated, not left behind
        f = tmp_path / ("alive%d" % r)
        if f.exists():
            pid = int(f.read_text())
            try:
                os.kill(pid, 0)
                alive = True
            except OSError:
                alive = False
            assert not alive, "rank %d (pid %d) still runs" % (r, pid)


def test_clean_run_passes_rank0_stdout_through(tmp_path, capsys):
    rc, dt, io = _launch(tmp_path, "ok", capsys)
    assert rc == 0 and io.out.strip() == '{"ok": true}'
    assert "rank 0 done" in io.err and "[rank 3] rank 3 done" in io.err
