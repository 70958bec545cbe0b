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


def test_self_launch_retries_once_with_the_other_ipc_mode(tmp_path):
    """First-contact safety: ranks that die before a JSON line (RCCL init / IPC handles / the collective pre-flight) get ONE
    fresh child with HSA_ENABLE_IPC_MODE_LEGACY flipped and a new rendezvous port; the JSON line of the attempt that worked
    is the one line on stdout; a child that fails both ways passes its exit code on."""
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent("""
        import json, os, sys
        want, port = sys.argv[1], sys.argv[sys.argv.index("--master-port") + 1]
        print("attempt with", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "port", port)
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != want:
            sys.exit(5)                                   # e.g. hipIpcGetMemHandle: invalid argument
        print(json.dumps({"metric": "m", "value": 2.0, "ipc": os.environ["HSA_ENABLE_IPC_MODE_LEGACY"],
                          "retry": os.environ.get("TT_BENCH_IPC_RETRY") == "1", "port": port}))
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {os.path.dirname(os.path.abspath(bench.__file__))!r})
        import bench
        env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"; env.pop("TT_BENCH_IPC_RETRY", None)
        bench.self_launch([sys.executable, {str(child)!r}, sys.argv[1], "--master-port", "29999"], env)
    """))
    import json

    r = subprocess.run([sys.executable, str(driver), "0"], capture_output=True, text=True, timeout=300)      # first setting works
    got = json.loads(r.stdout.strip())
    assert r.returncode == 0 and got["ipc"] == "0" and got["retry"] is False and got["port"] == "29999"
    r = subprocess.run([sys.executable, str(driver), "1"], capture_output=True, text=True, timeout=300)      # only the flipped one does
    assert r.returncode == 0 and len(r.stdout.strip().splitlines()) == 1
    got = json.loads(r.stdout.strip())
    assert got["ipc"] == "1" and got["retry"] is True and got["port"] != "29999"
    assert "one fresh attempt with HSA_ENABLE_IPC_MODE_LEGACY=1" in r.stderr
    r = subprocess.run([sys.executable, str(driver), "neither"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 5 and r.stdout.strip() == ""
    assert r.stderr.count("\nattempt with") + r.stderr.startswith("attempt with") == 2    # exactly one retry


def _preflight_rank(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench.preflight_collectives(dist, torch, torch.device("cpu"), rank, world)
        ret.put((rank, "ok"))
        # a rank that lies about its identity must be caught by its peers' value checks, not pass silently
        try:
            bench.preflight_collectives(dist, torch, torch.device("cpu"), rank if rank == 0 else rank + 5, world)
            ret.put((rank, "undetected"))
        except RuntimeError as e:
            ret.put((rank, "caught: " + str(e)))
    finally:
        dist.destroy_process_group()


def test_preflight_collectives_on_two_gloo_ranks():
    """The collective pre-flight every world > 1 bench run starts with: passes on a healthy group and raises, naming the rank
    and the collective, when a block arrives with another rank's values."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = bench._free_port()
    procs = [ctx.Process(target=_preflight_rank, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = [ret.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g for g in got if g[1] == "ok") == [(0, "ok"), (1, "ok")]
    rest = [g for g in got if g[1] != "ok"]
    assert len(rest) == 2 and all(g[1].startswith("caught: pre-flight:") for g in rest), rest


def _data_plane_rank(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cpu = torch.device("cpu")
        # a backend that cannot come up here (no GPU in this container): every rank must end on the default gloo group, agreed
        g, label, hung = bench.open_data_plane(dist, torch, cpu, rank, world, backend="nccl", deadline_s=60.0)
        bench.preflight_collectives(dist, torch, cpu, rank, world, group=g)
        ret.put((rank, "fallback", g is None, label, hung))
        # a backend that works: the step's collectives get their own group
        g2, label2, hung2 = bench.open_data_plane(dist, torch, cpu, rank, world, backend="gloo", deadline_s=60.0)
        bench.preflight_collectives(dist, torch, cpu, rank, world, group=g2)
        ret.put((rank, "own", g2 is not None, label2, hung2))
    finally:
        dist.destroy_process_group()


def test_data_plane_falls_back_to_gloo_by_agreement():
    """bench.py at world > 1: the step's collectives run on an RCCL group when RCCL passes the pre-flight on EVERY rank, on the
    default gloo group otherwise -- decided by agreement, reported in the JSON line (`collective_backend`)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = bench._free_port()
    procs = [ctx.Process(target=_data_plane_rank, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = [ret.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fb = sorted(g for g in got if g[1] == "fallback")
    assert [g[0] for g in fb] == [0, 1] and all(g[2] and g[3].startswith("gloo (nccl pre-flight failed") and not g[4] for g in fb), fb
    own = sorted(g for g in got if g[1] == "own")
    assert [g[0] for g in own] == [0, 1] and all(g[2] and g[3] == "gloo" and not g[4] for g in own), own


def test_clock_sampler_degrades_to_nulls_without_a_driver():
    """bench.py's clock / power sampler (amdsmi): on a box without the GPU driver every field is null and the error is carried --
    the JSON line is printed either way."""
    got = bench.ClockSampler(0, period_s=0.01).start().stop()
    assert set(got) >= {"sclk_mhz_median", "sclk_mhz_spec", "socket_power_w_mean", "power_cap_w", "samples", "error"}
    if got["samples"] == 0:
        assert got["sclk_mhz_median"] is None and got["socket_power_w_mean"] is None and got["error"]
