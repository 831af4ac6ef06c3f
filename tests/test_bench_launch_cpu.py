"""bench.py's launch logic, without a GPU: `--gpus N` with no WORLD_SIZE starts N ranks as a child
process (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) -- checked through the dry-run flag -- and a
WORLD_SIZE that disagrees with --gpus is refused.  Also the rank -> host cores mapping."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=120)


def test_gpus_n_without_a_launcher_spawns_n_ranks():
    p = _run(["--gpus", "4", "--steps", "3", "--warmup", "1", "--dry-run-spawn"])
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1 :] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--dry-run-spawn"]  # the ranks get the same arguments


def test_world_size_that_disagrees_with_gpus_is_refused():
    p = _run(["--gpus", "8"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert "refusing" in p.stderr and '"metric"' not in p.stdout
    p = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0


def test_rank_cpus_splits_the_cores_evenly():
    sys.path.insert(0, ROOT)
    import bench

    avail = list(range(64))
    seen = []
    for r in range(8):
        cpus, how = bench.rank_cpus(r, 8, None, avail)
        assert len(cpus) == 8 and how == "contiguous share"
        seen += cpus
    assert sorted(seen) == avail
    assert bench.rank_cpus(0, 1, None, avail)[0] == avail
    assert bench.cpu_list("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]


def test_rank_cpus_follows_the_numa_node_of_each_gpu():
    """Ranks whose GPUs sit on the same NUMA node split that node's cores by their order among those GPUs, whatever the
    device numbering (here: interleaved).  Uses this machine's node 0 if sysfs has one."""
    sys.path.insert(0, ROOT)
    import bench

    if not os.path.exists("/sys/devices/system/node/node0/cpulist"):
        import pytest

        pytest.skip("no NUMA information in sysfs")
    node0 = bench.cpu_list(open("/sys/devices/system/node/node0/cpulist").read())
    avail = sorted(os.sched_getaffinity(0))
    usable = [c for c in node0 if c in set(avail)]
    if len(usable) < 4:
        import pytest

        pytest.skip("too few cores on node 0")
    nodes = [0, 7, 0, 7, 0, 7, 0, 7]  # GPUs 0, 2, 4, 6 on node 0; the others on a node this machine may not have
    seen = []
    for r in (0, 2, 4, 6):
        cpus, how = bench.rank_cpus(r, 8, None, avail, nodes=nodes)
        assert how.startswith("numa node 0") and len(cpus) == len(usable) // 4
        seen += cpus
    assert len(set(seen)) == len(seen) and set(seen) <= set(usable)


def test_rehearse_collective_runs_the_single_rank_under_the_launcher():
    """--gpus 1 --rehearse-collective: the one rank is started by torch.distributed.run as well (WORLD_SIZE = 1), so the
    process group and the gather run on the one GPU of a box; a plain --gpus 1 stays a plain process."""
    p = _run(["--gpus", "1", "--rehearse-collective", "--dry-run-spawn"])
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "1" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"


def test_a_single_rank_may_be_bound_to_its_gpus_numa_node_or_to_another():
    sys.path.insert(0, ROOT)
    import bench

    if not os.path.exists("/sys/devices/system/node/node0/cpulist"):
        import pytest

        pytest.skip("no NUMA information in sysfs")
    avail = sorted(os.sched_getaffinity(0))
    node0 = [c for c in bench.cpu_list(open("/sys/devices/system/node/node0/cpulist").read()) if c in set(avail)]
    cpus, how = bench.rank_cpus(0, 1, None, avail, nodes=[0])
    assert cpus == node0 and how.startswith("numa node 0")
    assert bench.rank_cpus(0, 1, None, avail, nodes=[0], numa="none") == (avail, "all")
    assert bench.rank_cpus(0, 1, None, avail, nodes=[-1]) == (avail, "all")
