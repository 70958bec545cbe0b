"""bench.py's own rank launcher (`python bench.py --gpus N` with no rank environment): the decision is a pure function
taken before torch is imported; the ranks run as a CHILD process (a process that may have touched the GPU is never
replaced) and rank 0's JSON line is relayed as the last line."""
import os
import subprocess
import sys
import textwrap

import bench


def test_single_gpu_runs_in_process():
    assert bench.launch_plan([], {}) is None
    assert bench.launch_plan(["--gpus", "1", "--steps", "3"], {}) is None
    assert bench.launch_plan(["--gpus=1"], {}) is None


def test_rank_environment_is_never_relaunched():
    # the driver's form: python -m torch.distributed.run ... bench.py --gpus 8 -> every rank sees WORLD_SIZE / RANK
    for env in ({"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}, {"RANK": "0"}, {"WORLD_SIZE": "1"}):
        assert bench.launch_plan(["--gpus", "8"], env) is None


def test_plain_multi_gpu_command_starts_its_own_ranks():
    for argv in (["--gpus", "4", "--steps", "5", "--warmup", "2"], ["--steps", "5", "--gpus=4", "--warmup", "2"]):
        cmd = bench.launch_plan(argv, {})
        assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
        assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
        i = cmd.index(os.path.abspath(bench.__file__))
        assert cmd[i + 1:] == argv                               # the ranks get the caller's flags unchanged
    assert bench.launch_plan(["--gpus", "2"], {"MASTER_PORT": "29511"})[8] == "29511"


def test_self_launch_relays_json_last_and_exit_code(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent("""
        import sys
        print("noise before")
        print('{"metric": "m", "value": 1.0, "n_gpus": 2}')
        print("noise after")
        sys.exit(int(sys.argv[1]))
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.dirname(os.path.abspath(bench.__file__))!r})
        import bench
        bench.self_launch([sys.executable, {str(child)!r}, sys.argv[1]])
    """))
    for rc in (0, 7):
        r = subprocess.run([sys.executable, str(driver), str(rc)], capture_output=True, text=True, timeout=300)
        assert r.returncode == rc
        assert r.stdout.strip().splitlines() == ['{"metric": "m", "value": 1.0, "n_gpus": 2}']     # the one line on stdout
        assert r.stderr.strip().splitlines()[-2:] == ["noise before", "noise after"]
