"""bench.py as its own launcher: `bench.py --gpus N` without a launcher's WORLD_SIZE starts N fresh
ranks (the reference gets every GPU from one command, train/main.py:869-874).  Driven here with a
stub child so no GPU is needed."""
import io
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "stub_rank.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launcher_sets_rank_env_and_relays_rank0(tmp_path):
    cmd = _stub(tmp_path, """
        import json, os, sys
        rec = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        open(os.path.join(os.path.dirname(__file__), "seen_%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        print(json.dumps({"n_gpus": int(rec["WORLD_SIZE"]), "rank": int(rec["RANK"])}), flush=True)
    """)
    out = io.StringIO()
    assert bench.launch_ranks(3, cmd, ndev=3, out=out) == 0
    lines = [json.loads(l) for l in out.getvalue().splitlines() if l.strip()]
    assert lines == [{"n_gpus": 3, "rank": 0}]          # only rank 0's line reaches stdout
    seen = [json.loads((tmp_path / f"seen_{r}.json").read_text()) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"]
    assert [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert {s["WORLD_SIZE"] for s in seen} == {"3"}
    assert {s["MASTER_ADDR"] for s in seen} == {"127.0.0.1"}
    assert len({s["MASTER_PORT"] for s in seen}) == 1


def test_launcher_refuses_more_ranks_than_devices(tmp_path):
    cmd = _stub(tmp_path, "print('should not run')")
    out = io.StringIO()
    assert bench.launch_ranks(2, cmd, ndev=1, out=out) != 0
    assert out.getvalue() == ""


def test_launcher_propagates_a_failing_rank_and_stops_the_others(tmp_path):
    cmd = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)          # a rank left waiting in a collective
    """)
    import time
    t0 = time.monotonic()
    assert bench.launch_ranks(2, cmd, ndev=2, out=io.StringIO()) == 7
    assert time.monotonic() - t0 < 60


def test_bench_cli_without_devices_exits_nonzero():
    """`bench.py --gpus 2` with no launcher above it and fewer than two visible GPUs must fail, not
    record one rank's work under an N = 2 label."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
