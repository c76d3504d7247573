"""CPU: `bench.py --gpus N` without a launcher around it starts N rank
processes of its own (fresh children, one per GPU, rendezvous on 127.0.0.1)
and relays rank 0's line.  The probe mode joins the process group and touches
no GPU, so the launch path is covered here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_rank_commands_describe_one_process_per_gpu():
    cmds = bench.rank_commands(4, ["--gpus", "4", "--steps", "2"], 29512, base_env={"PATH": os.environ.get("PATH", "")})
    assert len(cmds) == 4
    for r, (argv, env) in enumerate(cmds):
        assert argv[0] == sys.executable and argv[-4:] == ["--gpus", "4", "--steps", "2"]
        assert os.path.basename(argv[2]) == "bench.py"
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"]) == (str(r), str(r), "4")
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29512"


def test_gpus_flag_spawns_the_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rank-probe"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 3 and line["rank_sum"] == 1 + 2 + 3  # every rank joined


def test_a_failing_rank_fails_the_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong",
                        "--workload", "fbw_rle", "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0  # no GPU here / strong scaling is undefined for this table: either way the ranks exit non-zero
