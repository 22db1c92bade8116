"""bench.py's launch contract on CPU (emulated kernels, tiny circuit): `--gpus N` without a launcher spawns N ranks
that really form an N-rank communicator, and asking for more RCCL ranks than there are devices fails loudly."""
import json
import os
import subprocess
import sys

from conftest import EMU_LIB, REPO


def _run(extra, timeout=600, detail=None):
    env = dict(os.environ, PLONK_HIP_LIB=EMU_LIB, PLONK_MSM_TABLE_GB="0.0001")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--log-n", "4", "--batch", "3", "--batches-per-step", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-microbench", "--no-fallbacks", "--no-lookup", "--no-latency"] + extra
    if detail is not None:
        cmd += ["--detail", str(detail)]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")


def _one_short_line(stdout):
    """The driver's record keeps the tail of stdout: ONE JSON line, under 4 KB, with every contract field (round 4's line had
    grown to 20 KB and the driver could not parse it)."""
    lines = stdout.strip().splitlines()
    assert len(lines) == 1, stdout
    assert len(lines[0]) < 4096, len(lines[0])
    line = json.loads(lines[0])
    assert all(k in line for k in CONTRACT_KEYS), sorted(line)
    assert "workload" in line["config"] and len(line["config"]) <= 13
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())  # scalars only
    return line


def test_gpus_flag_spawns_that_many_ranks(emu_cdll, tmp_path):
    r = _run(["--gpus", "2", "--dist-backend", "sockets", "--no-end-to-end"], detail=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_short_line(r.stdout)  # ONE short JSON line and nothing else
    detail = json.load(open(tmp_path / "d.json"))  # everything else the run measured
    assert detail["line"] == line
    assert line["n_gpus"] == 2 and line["config"]["ranks_in_communicator"] == 2
    assert detail["config"]["results_gathered_per_step"] == 2 * 3 * 2 and detail["config"]["gather_in_timed_region"]
    assert line["scaling"] == "weak" and line["unit"] == "proofs/s" and line["value"] > 0
    assert line["config"]["sampled_proofs_verify"] is True and len(detail["sampled_verify"]["indices"]) == 4
    pr = detail["per_rank"]  # every rank's own figures, so that a scaling record explains itself
    assert len(pr["proofs_per_s"]) == 2 and all(x > 0 for x in pr["proofs_per_s"])
    assert pr["proofs_per_s_min"] <= pr["proofs_per_s_max"] and len(pr["allgather_us_per_step"]) == 2
    assert pr["proofs_per_s_sum"] >= line["value"] * 0.999
    assert len(line["per_rank"]["proofs_per_s"]) == 2  # the short form travels in the line
    # the detail also went to stderr as one line
    assert any(l.startswith("bench_detail: {") for l in r.stderr.splitlines())


def test_force_comm_runs_the_multi_gpu_code_path_with_one_rank(emu_cdll, tmp_path):
    """`--gpus 1 --force-comm`: a one-rank communicator, the device-resident gather, the max over ranks and the barrier inside
    the timed region, and the per-rank block of an N-GPU line (here over the emulation's communicator stub; the same command
    runs against RCCL on the GPU box: tests/test_gpu_multiprocess.py)."""
    r = _run(["--gpus", "1", "--force-comm"], detail=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_short_line(r.stdout)
    detail = json.load(open(tmp_path / "d.json"))
    cfg = detail["config"]
    assert line["config"]["gather_transport"] == "rccl" and line["config"]["ranks_in_communicator"] == 1
    assert cfg["gather_transport"] == "rccl" and cfg["gather_in_timed_region"] and cfg["ranks_in_communicator"] == 1
    assert cfg["gather_path"].startswith("device buffers") and cfg["results_gathered_per_step"] == 6
    assert cfg["torch_imported"] is False and "rccl_path" in cfg and "rccl_version" in cfg
    assert "end_to_end" in detail and detail["end_to_end"]["fraction_of_value"] > 0
    pr = detail["per_rank"]
    assert len(pr["proofs_per_s"]) == 1 and pr["proofs_per_s_min"] == pr["proofs_per_s_max"] > 0
    assert len(pr["msm_table_build_s"]) == 1 and len(pr["allgather_us_per_step"]) == 1
    # the whole-job value is measured over the barrier, a rank's own rate before it: never below it
    assert pr["proofs_per_s_sum"] >= line["value"] * 0.999


def test_more_rccl_ranks_than_devices_is_an_error(emu_cdll):
    r = _run(["--gpus", "3"])  # the emulation reports one device
    assert r.returncode != 0 and "refusing" in r.stderr


def test_world_size_must_match_gpus(emu_cdll):
    env = dict(os.environ, PLONK_HIP_LIB=EMU_LIB, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--log-n", "4"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_launched_by_torch_distributed_run(emu_cdll):
    """The driver's form: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` — RANK / WORLD_SIZE / MASTER_*
    come from the launcher (whose own store owns MASTER_PORT: the rendezvous uses MASTER_PORT + 1), no torch in bench.py itself."""
    env = dict(os.environ, PLONK_HIP_LIB=EMU_LIB, PLONK_MSM_TABLE_GB="0.0001")
    env.pop("WORLD_SIZE", None)
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "sockets", "--log-n", "4",
           "--batch", "3", "--batches-per-step", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-microbench",
           "--no-fallbacks", "--no-lookup", "--no-latency", "--no-end-to-end"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_short_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["ranks_in_communicator"] == 2 and line["config"]["gather_transport"] == "sockets"


def test_clock_sampler_parses_rocm_smi_text():
    """The sampler's regular expressions on rocm-smi's own layout (two GPUs listed; only the asked index is read)."""
    sys.path.insert(0, REPO)
    import bench

    text = """
============================ ROCm System Management Interface ============================
====================================== Current clock frequencies ======================================
GPU[0]		: fclk clock level: 0: (1250Mhz)
GPU[0]		: mclk clock level: 3: (2000Mhz)
GPU[0]		: sclk clock level: 1: (2077Mhz)
GPU[0]		: socclk clock level: 0: (28Mhz)
GPU[1]		: sclk clock level: S: (95Mhz)
=================================== Power Consumption ====================================
GPU[0]		: Current Socket Graphics Package Power (W): 1141.0
GPU[1]		: Current Socket Graphics Package Power (W): 234.0
"""
    f = [(int(m.group(1)), int(m.group(2))) for m in bench.ClockSampler.SCLK.finditer(text)]
    w = [(int(m.group(1)), float(m.group(2))) for m in bench.ClockSampler.POWER.finditer(text)]
    assert f == [(0, 2077), (1, 95)] and w == [(0, 1141.0), (1, 234.0)]
    s = bench.ClockSampler(0)
    s.samples = [(2077, 1141.0), (2050, 1200.0), (2404, None)]
    out = s.summary()
    assert out["sclk_mhz_median"] == 2077 and out["sclk_mhz_min"] == 2050 and out["socket_power_w_median"] == 1200.0
    assert bench.ClockSampler(0).summary() is None  # nothing sampled (no rocm-smi): the bench line carries null


def test_a_rank_that_dies_mid_step_stops_the_job_and_is_named(emu_cdll):
    """rank 1 exits between enqueueing its batches and the gather of timed step 1.  The launcher watches every rank:
    the job ends non-zero within seconds (not in a 30-minute timeout), names rank 1, its exit code and the tail of its stderr, and
    does not blame rank 0, which only lost its peer (PeerLost -> exit code 76)."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--dist-backend", "sockets", "--no-end-to-end", "--steps", "3", "--inject-fault", "1:1:exit"], timeout=120)
    took = time.monotonic() - t0
    assert r.returncode != 0 and not r.stdout.strip(), (r.returncode, r.stdout[-500:])
    assert "rank 1 of 2 exited with code 17 first" in r.stderr, r.stderr[-3000:]
    assert "rank 0 exited with code 76 (it lost a peer)" in r.stderr and "PeerLost: rank 1 closed its connection" in r.stderr
    assert "INJECTED FAULT 'exit' in timed step 1" in r.stderr  # the stderr tail of the rank that failed
    assert took < 60, took  # (start-up and one emulated warm-up step: ~15 s here; the failure itself is seen at once)


def test_a_rank_that_hangs_mid_step_is_timed_out_and_named(emu_cdll):
    """The same with a rank that stops without exiting: the other rank's collective has a deadline (--comm-timeout), it gives up
    naming the silent rank, and the launcher stops the stuck one."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--dist-backend", "sockets", "--no-end-to-end", "--steps", "3", "--inject-fault", "1:1:hang", "--comm-timeout", "3"], timeout=120)
    took = time.monotonic() - t0
    assert r.returncode != 0 and not r.stdout.strip()
    assert "PeerLost: rank 1 sent nothing for 3 s" in r.stderr, r.stderr[-3000:]
    assert "rank 1 of 2 did not exit and had to be stopped (stuck?)" in r.stderr
    assert "rank 1 stopped by the launcher while still running" in r.stderr
    assert took < 60, took


def test_preflight_reports_every_rank(emu_cdll):
    r = _run(["--gpus", "2", "--dist-backend", "sockets", "--preflight"], timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    rep = json.loads(lines[0])
    assert rep["preflight"] is True and rep["n_gpus"] == 2 and rep["ranks_in_communicator"] == 2 and len(rep["ranks"]) == 2
    for k, row in enumerate(rep["ranks"]):
        assert row["rank"] == k and row["transport"] == "sockets" and row["peer_access"] == [1] and row["hbm_total_gb"] > 0
        assert row["comm_init_s"] >= 0 and row["first_step_s"] > 0 and row["allgather_us"] > 0 and "msm_table" in row
