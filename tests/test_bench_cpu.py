"""CPU-tier checks of bench.py: the index arithmetic that crashed the round-1 driver run (`--steps 20 --warmup 5` sliced past
the action pool), the self-launch line, and a 2-rank gloo dry run of the launcher (no GPU, no physics)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("steps,warmup", [(20, 5), (1, 0), (2000, 50), (7, 300)])
def test_every_index_stays_inside_the_action_pool(steps, warmup):
    n = bench.N_ACT
    for t in range(warmup + 3 * steps + 5):
        assert 0 <= bench.action_index(t) < n
    for rk in (bench.RK, 1, 7, n):
        for i in range(3 * n // rk + 5):
            lo, hi = bench.rollout_slice(i, rk, n)
            assert 0 <= lo < hi <= n and hi - lo == rk
    # the pools of the secondary legs (256 batches, 16-step rollouts)
    for i in range(40):
        lo, hi = bench.rollout_slice(i, 16, 256)
        assert hi - lo == 16 and hi <= 256


def test_repeats_reach_the_minimum_time():
    assert bench.repeats_for(0.001, 0.5) == 500
    assert bench.repeats_for(0.6, 0.5) == 1
    assert bench.repeats_for(0.0, 0.5) == 2000
    assert bench.repeats_for(1e-9, 0.5) == 2000  # capped
    assert bench.repeats_for(0.3, 0.5) == 2


def test_launch_command_is_the_drivers_line():
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")


def test_defaults_are_one_gpu_and_bounded():
    a = bench.parse_args([])
    assert a.gpus == 1 and a.steps == 2000 and a.warmup == 50


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself (round 1 silently ran one)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3.0


def test_without_a_gpu_the_bench_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)


def test_rank_cpu_plan_gives_every_rank_its_own_cores_next_to_its_gpu():
    """N > 1 host placement (bench.rank_cpu_plan): disjoint slices, each inside the NUMA node of the rank's GPU when the topology
    is known, the stepping thread's core inside the slice; even split of the allowed cores otherwise."""
    numa = {0: list(range(0, 64)), 1: list(range(64, 128))}
    gpus = [0, 0, 0, 0, 1, 1, 1, 1]
    seen = set()
    for r in range(8):
        mine, spin = bench.rank_cpu_plan(r, 8, range(128), gpus, numa)
        assert len(mine) == 16 and spin == mine[0] and set(mine) <= set(numa[gpus[r]]) and not (set(mine) & seen)
        seen |= set(mine)
    assert seen == set(range(128))
    # a cpuset that leaves one node almost empty: that node's ranks fall back to the even split of what is allowed
    allowed = list(range(0, 64)) + [64, 65]
    slices = [bench.rank_cpu_plan(r, 8, allowed, gpus, numa)[0] for r in range(8)]
    assert all(slices) and all(set(s) <= set(allowed) for s in slices)
    assert len(slices[0]) == 16 and set(slices[0]) <= set(numa[0])          # node 0 ranks keep their NUMA-local cores
    # no topology: even split; fewer cores than ranks: no plan (the scheduler places the ranks)
    assert bench.rank_cpu_plan(1, 2, range(8)) == ([4, 5, 6, 7], 4)
    assert bench.rank_cpu_plan(0, 8, range(4)) == ([], None)
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_sysfs_probes_never_raise():
    nodes, cpus = bench.gpu_numa_nodes(), bench.numa_cpu_lists()
    assert nodes is None or isinstance(nodes, list)
    assert isinstance(cpus, dict)


def test_compact_line_keeps_the_head_and_fits_the_drivers_tail():
    """The printed line (bench.compact_line): what the driver's record must show comes first and in full, every string is short, and
    secondary detail is dropped -- in a fixed order, listed in the line -- until it fits 6000 bytes."""
    long = "x" * 900
    out = {"metric": bench.METRIC, "value": 1.7e8, "unit": "env-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.024,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w", "host_thread": "pinned to cpu 3", "obs_gather": long},
           "early_mask": {"sent": 123, "mismatches": 0, "workgroup_launches": 456, "note": long},
           "value_median_region": 1.8e8, "hot_path_rate": 2.0e8, "api_over_hot_path": 0.85,
           "roofline": {"bound": "hbm", "frac": 0.0117, "note": long, "trace": long},
           "cpu_baseline": {"value": 1e6, "sample": long},
           "sync_step_floor": {"note": long, "filler": ["y" * 100] * 30}, "repeat_us_per_step": {"min": 1.0, "filler": ["z" * 100] * 30},
           "secondary": {"batch_sweep": {"rows": [[1024, 1.0]]}}, "scripted_grasp": {"env_step_exact_contacts": {"overflow_env_frac": 0.016}}}
    c = bench.compact_line(out)
    line = json.dumps(c)
    assert len(line) <= bench.LINE_LIMIT
    assert list(c)[:8] == ["metric", "value", "unit", "value_median_region", "hot_path_rate", "api_over_hot_path", "early_mask_sent", "early_mask_mismatches"]
    assert c["early_mask_sent"] == 123 and c["early_mask_mismatches"] == 0 and c["config"]["early_mask"]["mismatches"] == 0
    assert c["config"]["hot_path_rate"] == c["hot_path_rate"] and "trace" not in c["roofline"] and len(c["roofline"].get("note", "")) <= bench.NOTE_LIMIT
    assert len(c["cpu_baseline"]["sample"]) <= bench.NOTE_LIMIT and len(c["config"]["obs_gather"]) <= bench.NOTE_LIMIT
    assert c["dropped_for_length"][0] == "sync_step_floor" and "secondary" in c and "scripted_grasp" in c
    assert out["early_mask"]["sent"] == 123 and "early_mask_sent" not in out   # (the full record is left as it was)


def test_preflight_places_ranks_or_refuses():
    """bench.preflight (the launching parent, before any rank exists): one line per rank with device, NUMA node, cores and peer
    access; non-zero when the ranks do not fit the GPUs."""
    import io

    buf = io.StringIO()
    assert bench.preflight(8, False, device_count=8, peer_matrix=[[True] * 8] * 8, out=buf) == 0
    txt = buf.getvalue()
    assert txt.count("rank ") == 8 and "cuda:7" in txt and "can address peers 11111111" in txt
    buf = io.StringIO()
    assert bench.preflight(4, False, device_count=1, peer_matrix=None, out=buf) == 2 and "FAILED" in buf.getvalue()
    buf = io.StringIO()
    assert bench.preflight(2, True, device_count=1, peer_matrix=None, out=buf) == 0 and "shared" in buf.getvalue()
    buf = io.StringIO()
    pm = [[True, False], [False, True]]
    assert bench.preflight(2, False, device_count=2, peer_matrix=pm, out=buf) == 0 and "RCCL collective" in buf.getvalue()


def test_more_ranks_than_gpus_is_refused_before_any_rank_starts():
    """`python bench.py --gpus 3` in a container without GPUs: the parent's preflight prints the placement table and exits non-zero;
    no rank is spawned."""
    import torch

    if torch.cuda.device_count() >= 3:
        pytest.skip("three GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 2 and "[bench preflight] FAILED" in r.stderr and "rank 2: device NONE" in r.stderr


def test_cpu_baseline_runs_in_a_child_with_bound_threads_and_reports_its_spread():
    cb = bench.cpu_baseline(1.0)
    assert cb["kind"] == "port" and cb["min"] <= cb["value"] <= cb["max"] and cb["threads"] == cb["cores"] >= 1 and cb["value"] > 1e4
    assert "best of 5 runs" in cb["sample"] and cb["value"] == cb["max"] >= cb["median"] >= cb["min"]
