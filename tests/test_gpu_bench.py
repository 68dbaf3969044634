"""The driver's own bench command, run as a subprocess on the GPU box: it must exit 0 and its last stdout line must be the
JSON contract line carrying `roofline` and `cpu_baseline` (round 1's driver run crashed before printing anything)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*flags, timeout=900, raw=False):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    last = r.stdout.strip().splitlines()[-1]
    return (json.loads(last), last, r.stderr) if raw else json.loads(last)


def test_driver_command_prints_the_contract_line():
    out, line, _ = _run("--gpus", "1", "--steps", "20", "--warmup", "5", raw=True)
    # the driver keeps 8 KB tails: the line must fit with room to spare, and what its record has to show comes FIRST, in full
    assert len(line) <= 6000, len(line)
    keys = list(out)
    assert keys[:8] == ["metric", "value", "unit", "value_median_region", "hot_path_rate", "api_over_hot_path", "early_mask_sent", "early_mask_mismatches"]
    assert keys.index("config") < keys.index("roofline") < keys.index("cpu_baseline") < 22
    assert out["early_mask_mismatches"] == 0 and out["early_mask_sent"] > 0
    cfg = out["config"]
    assert cfg["early_mask"]["mismatches"] == 0 and cfg["value_median_region"] == out["value_median_region"] and cfg["hot_path_rate"] == out["hot_path_rate"]

    def strings(x):
        if isinstance(x, dict):
            for k, v in x.items():
                if k != "metric":
                    yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(t) for t in strings(out)) <= 120
    assert os.path.exists(os.path.join(ROOT, "gpurun_out", "bench_line_full.json"))   # (the uncut record)
    assert out["metric"].startswith("env-steps/sec") and out["unit"] == "env-steps/s" and out["n_gpus"] == 1
    assert out["steps"] == 20 and out["warmup"] == 5 and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["value"] > 1e6 and out["path"] == "GenesisEnv.step"
    assert abs(out["ms_per_step"] * 1e-3 * out["value"] - 4096) < 1.0          # value and ms_per_step describe the same time
    assert out["timed_seconds_total"] >= 0.4 and out["repeats"] * 20 == out["timed_steps_total"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - 489.0 * 4096 / (rf["kernel_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "env-steps/s"
    assert cb["min"] <= cb["median"] <= cb["value"] == cb["max"] and cb["threads"] == cb["cores"], cb   # (best of five runs of the same steps; the spread is the shared host's)
    sweep = out["secondary"]["batch_sweep"]
    assert [r[0] for r in sweep["rows"]] == [1024, 4096, 16384, 32768, 65536] and sweep["cols"][:3] == ["num_envs", "kernel_us", "bare_launch_rate"]
    assert sweep["cols"][7] == "single_wave_rollout8_rate" and all(r[7] and r[7] > 1e6 for r in sweep["rows"])
    assert all(r[2] > 1e6 and r[4] > 1e6 and 0 < r[6] < 1 for r in sweep["rows"])
    gx = out["scripted_grasp"]["env_step_exact_contacts"]
    assert gx["overflow_env_frac"] > 0.005 and gx["env_steps_per_s"] > 1e7 and abs(gx["lifted_frac"] - out["scripted_grasp"]["env_step_thinned"]["lifted_frac"]) < 0.05
    rx = out["ref_expert"]   # (both figures in the line -- every contact kept, the default, and the thinned speed knob)
    assert rx["exact_contacts"]["env_step_us"] > 0 and rx["thinned"]["env_step_us"] > 0 and rx["exact_contacts"]["overflow_env_frac"] > 0.1
    assert abs(rx["exact_contacts"]["lifted_frac"] - rx["thinned"]["lifted_frac"]) < 0.06 and cfg["exact_contacts"] is True
    assert out["ik"]["roofline_valu"]["frac"] > 0 and out["ik"]["iters_wave_max"] <= 32
    for key in ("secondary", "pixels", "scripted_grasp", "ref_expert", "box_links", "so101_pick", "stack", "ik"):
        assert key in out and "error" not in out[key], (key, out.get(key))
    assert out["hot_path_rate"] >= out["value"] * 0.9
    assert out["value_median_region"] >= out["value"] * 0.8 and out["config"]["host_thread"].startswith("pinned to cpu")


def test_two_ranks_share_the_gpu_over_gloo():
    """N > 1 plumbing on a 1-GPU box: bench.py spawns two ranks itself, both on cuda:0, process group over gloo, the env axis
    sharded, outputs gathered; rank 0 prints n_gpus == 2 with the world size the process group reports."""
    out = _run("--gpus", "2", "--steps", "20", "--warmup", "5", "--dist-backend", "gloo", "--oversubscribe", "--envs-per-gpu", "512",
               "--min-time", "0.1", "--gather-every", "4")
    assert out["n_gpus"] == 2 and out["config"]["world_size_observed"] == 2 and out["config"]["global_num_envs"] == 1024
    assert out["value"] > 0 and out["config"]["gather_path"] == "copy", out["config"]["obs_gather"]
    # every N > 1 line carries the CPU baseline (rank 0 times it while the others wait at the final barrier), the cost of the
    # gather as a number, and where the rank's threads were put
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert isinstance(out["gather_overhead_us"], float) and out["no_gather"]["value"] > 0
    # SURVEY.md cfg 3's "all-gather obs each step": the same loop with one step per gather, over the copy path and over the collective
    g1 = out["gather_every_1"]
    assert g1["copy"]["value"] > 0 and g1["rccl"]["value"] > 0 and isinstance(g1["copy"]["overhead_us_vs_no_gather"], float)
    gs = out["config"]["gather_stats"]   # (two ranks' copies into each other's buffers, checked against the collective at the end)
    assert gs["pushes"] > 0 and gs["checks"] >= 1 and gs["checks_failed"] == 0
    if len(os.sched_getaffinity(0)) >= 4:
        assert "stepping thread on cpu" in out["config"]["host_thread"], out["config"]["host_thread"]


def test_first_rccl_run_one_rank_force_gather():
    """The RCCL path on the MI355X with the one rank a 1-GPU box has: `nccl` process group bound to the device, the observation
    gather (all_gather_into_tensor of 8 steps' outputs), barriers and the max-over-ranks all-reduce all execute through RCCL;
    the line reports the world size the group observed and what the gather costs per step against the same loop without it."""
    out = _run("--gpus", "1", "--force-gather", "--dist-backend", "nccl", "--steps", "20", "--warmup", "5", "--no-pixels", "--no-stack",
               "--no-cpu-baseline")
    cfg = out["config"]
    assert out["n_gpus"] == 1 and cfg["world_size_observed"] == 1 and cfg["dist_backend"] == "nccl"
    assert cfg["gather_path"] == "copy" and out["value"] > 1e6
    assert isinstance(out["gather_overhead_us"], float) and abs(out["gather_overhead_us"]) < 50.0
    assert out["no_gather"]["mean_us_per_step"] > 0
    g1 = out["gather_every_1"]   # (one step per gather, both paths, through RCCL at world 1)
    assert g1["copy"]["value"] > 1e6 and g1["rccl"]["value"] > 1e6
    # every timed region ends with a push (partial chunks included), the last gather of a region was checked against the collective,
    # and the ring of reused output rows is disclosed
    gs = cfg["gather_stats"]
    assert gs["pushes"] > 0 and gs["partial_pushes"] > 0 and gs["checks"] >= 1 and gs["checks_failed"] == 0
    assert "ring" in cfg["output_ring"]



def test_rccl_collective_path_still_runs():
    """`--gather-path rccl`: the all_gather_into_tensor variant (the fallback of the copy path), one rank over nccl and two ranks
    sharing the GPU over gloo."""
    out = _run("--gpus", "1", "--force-gather", "--dist-backend", "nccl", "--gather-path", "rccl", "--steps", "20", "--warmup", "5", "--no-pixels",
               "--no-stack", "--no-cpu-baseline", "--min-time", "0.2")
    assert out["config"]["gather_path"] == "rccl" and "nccl all_gather" in out["config"]["obs_gather"]
    out = _run("--gpus", "2", "--steps", "20", "--warmup", "5", "--dist-backend", "gloo", "--oversubscribe", "--envs-per-gpu", "512", "--min-time", "0.1",
               "--gather-every", "4", "--gather-path", "rccl", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["config"]["gather_path"] == "rccl" and "gloo all_gather" in out["config"]["obs_gather"]


def test_copy_path_that_cannot_map_a_peer_falls_back_to_the_collective_on_every_rank():
    """Dry run of the first multi-GPU launch: `mir_p2p_enable` is made to fail on ONE of two ranks
    (MIR_P2P_FORCE_FAIL=1: rank 1's calls).  The ranks agree -- the failing one must not skip a collective the other is inside of --
    on the RCCL / gloo collective, the run finishes with rc 0, and the line says which path it took and why."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MIR_P2P_FORCE_FAIL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dist-backend", "gloo", "--oversubscribe",
                        "--envs-per-gpu", "512", "--min-time", "0.1", "--gather-every", "4", "--gather-path", "copy", "--no-cpu-baseline", "--no-pixels", "--no-stack"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["world_size_observed"] == 2 and out["value"] > 0
    assert cfg["gather_path"] == "rccl", cfg
    assert cfg["gather_note"] and "mapping the peers' buffers failed on some rank" in cfg["gather_note"], cfg["gather_note"]


def test_a_rank_that_dies_mid_region_takes_the_run_down_promptly():
    """ rank 1 of two exits in the middle of a timed region (MIR_BENCH_KILL_RANK=1:40): the launcher ends the
    other rank, the command returns non-zero well inside the gather's and the process group's time-outs, prints no result line, and
    nothing is restarted."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MIR_BENCH_KILL_RANK"] = "1:40"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dist-backend", "gloo", "--oversubscribe",
                        "--envs-per-gpu", "512", "--min-time", "0.1", "--gather-every", "4", "--no-cpu-baseline", "--no-pixels", "--no-stack"],
                       capture_output=True, text=True, env=env, timeout=600)
    took = time.time() - t0
    assert r.returncode != 0 and took < 240, (r.returncode, took, r.stderr[-2000:])
    assert not any(line.startswith('{"metric"') for line in r.stdout.splitlines())
    assert "exitcode" in r.stderr and "17" in r.stderr, r.stderr[-3000:]   # (torch.distributed.run's failure summary names the rank that died)
    assert r.stderr.count("rank(s) requested") == 1   # (one launch: nothing was started again)
