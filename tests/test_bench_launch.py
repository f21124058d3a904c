"""`python bench.py --gpus N` as the driver invokes it (no outer torchrun): the parent starts the ranks as child processes
through torch.distributed.run before it has imported torch, and exits with the launcher's code (What's missing
#2; the reference's multi-GPU entry is train_binary.py:546-549).  No GPU here: BMP_BENCH_RANK_CHECK_ONLY=1 makes every rank
report itself right after the rank check, before the first device call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BMP_BENCH_RANK_CHECK_ONLY"] = "1"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _reports(out):
    import re
    return [json.loads(m) for m in re.findall(r'\{"rank_check"[^{}]*\}', out)]       # (ranks share one stdout)


def test_gpus_2_without_torchrun_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _reports(r.stdout)
    assert sorted(x["rank"] for x in recs) == [0, 1], (r.stdout, r.stderr[-2000:])
    assert all(x["world"] == 2 for x in recs)
    assert all(x["master"].startswith("127.0.0.1:") for x in recs)


def test_gpus_1_runs_in_process():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _reports(r.stdout)
    assert len(recs) == 1 and recs[0]["world"] == 1 and recs[0]["rank"] == 0


def test_world_mismatch_is_refused():
    r = _run(["--gpus", "2"], extra_env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_launcher_parent_does_not_import_torch():
    # the parent's decision is taken before numpy / torch are bound: a failing import in the child path cannot be reached
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def _heavy_imports")[0]
    assert "import torch" not in head.replace("import torch as _torch", "")
    code = ("import sys, runpy; sys.argv=['bench.py','--gpus','2'];\n"
            "import os; os.environ['BMP_BENCH_RANK_CHECK_ONLY']='1'\n"
            "import subprocess; subprocess.call=lambda *a, **k: 0\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code\n"
            "assert 'torch' not in sys.modules, 'launcher imported torch'\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
