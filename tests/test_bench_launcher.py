"""bench.py --gpus N starts its own N ranks when no launcher (torchrun) is around it.

CPU part: the launcher + rendezvous + rank bookkeeping through `--workload selftest`
(gloo, no GPU).  GPU part: the real workload with two ranks sharing the one card of the
test box over gloo (RCCL needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks():
    r = run(["--gpus", "2", "--workload", "selftest", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    ranks = d["config"]["ranks"]
    assert sorted(x["rank"] for x in ranks) == [0, 1]
    assert sorted(x["local_rank"] for x in ranks) == [0, 1]
    assert len({x["pid"] for x in ranks}) == 2          # two processes, not one
    assert d["value"] == 1.0                            # MAX over ranks of the rank number


def test_bench_single_rank_needs_no_rendezvous():
    r = run(["--workload", "selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout)["n_gpus"] == 1


def test_bench_fails_when_a_rank_fails():
    r = run(["--gpus", "2", "--workload", "selftest"], {"E2_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not r.stdout.strip()


def test_bench_refuses_a_world_that_is_not_gpus():
    # the round-2 script accepted WORLD_SIZE=1 for any --gpus and printed n_gpus: 1
    r = run(["--gpus", "2", "--workload", "selftest"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_launcher_survives_a_noisy_rank_0():
    # > 64 KiB on rank 0's stdout before the line: a pipe nobody drains would block rank 0 in
    # write() and the other rank in the barrier behind it
    r = run(["--gpus", "2", "--workload", "selftest"], {"E2_SELFTEST_NOISE": "300000"}, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_a_local_rank_without_a_gpu_is_an_error_under_rccl():
    from elektronn2_amd import parallel
    assert parallel.local_device("nccl", local_rank=3, device_count=8) == 3
    with pytest.raises(RuntimeError, match="LOCAL_RANK=5"):
        parallel.local_device("nccl", local_rank=5, device_count=4)
    with pytest.raises(RuntimeError, match="LOCAL_RANK=1"):
        parallel.local_device("nccl", local_rank=1, device_count=1)
    # the gloo rehearsal (several ranks on the one card of a test box) wraps around
    assert parallel.local_device("gloo", local_rank=1, device_count=1) == 0
    assert parallel.local_device("gloo", local_rank=5, device_count=4) == 1


def test_two_ranks_on_one_gpu_fail_the_run_under_rccl():
    from elektronn2_amd import parallel
    ok = [{"rank": r, "host": "n0", "pci": r * 16, "uuid": "u%d" % r} for r in range(8)]
    parallel.check_distinct_devices(ok, "nccl")
    bad = [dict(d) for d in ok]
    bad[5]["pci"] = bad[2]["pci"]
    with pytest.raises(RuntimeError, match="ranks 2 and 5"):
        parallel.check_distinct_devices(bad, "nccl")
    parallel.check_distinct_devices(bad, "gloo")                   # a rehearsal shares the card
    other_host = [dict(d) for d in bad]
    other_host[5]["host"] = "n1"                                   # same bus id on ANOTHER host
    parallel.check_distinct_devices(other_host, "nccl")


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
            {"E2_DIST_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2"
    ranks = d["config"]["ranks"]
    assert sorted(x["rank"] for x in ranks) == [0, 1]
    assert all(x["backend"] == "gloo" for x in ranks)
    assert d["value"] > 0 and d["scaling"] == "weak"
