"""`python bench.py --gpus N` must start its own N ranks (the driver's command form), aggregate rank 0's JSON line and
fail loudly when a rank dies.  CPU: --dry-run keeps the launcher, rendezvous (gloo, 127.0.0.1) and MAX-over-ranks
reduction and skips the GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=240)


def test_gpus2_self_launch_prints_one_json_line():
    r = _run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ranks"] == [[0, 0], [1, 1]]          # one process per GPU: LOCAL_RANK r for rank r
    assert out["elapsed_max_s"] >= 0.02              # MAX over ranks (rank 1 sleeps longer)


def test_failed_rank_fails_the_bench():
    r = _run("--gpus", "2", "--dry-run", env={"VTD_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "does not match" in r.stderr


def test_roofline_traffic_lookup_covers_the_shipped_dominant_kernel_and_fails_loudly_otherwise(tmp_path):
    """bench.py cites HBM bytes per launch from the committed rocprofv3 --pmc summary.  The summary must hold the kernel the shipped
    kernel-selection table runs as the dominant launch (the composed head entry on head_entry_halo256 at B = 32), and a dominant
    kernel it does not hold must surface as an error string in the JSON line, never as a silent null."""
    import json
    import bench
    name = "head_entry_halo256 M/img=25600 N=64 K=3904 (lateral+smooth+head conv composed; border classes in the next slot)"
    traffic, detail, err = bench.lookup_traffic(name)
    assert err is None and traffic > 100e6 and detail["kernel_symbol"] == "head_entry_halo256_kernel<false>("
    # the shipped table really selects that kernel for the composed head entry at the bench's batch
    table = open(os.path.join(ROOT, "video-text-detection-system_amd", "vtd_amd", "tuning", "gfx950.txt")).read().splitlines()
    classed = [ln for ln in table if ln.startswith("conv|in160x160x64|") and "|c1|n32 " in ln]
    assert classed and all(ln.rsplit(" ", 1)[1] == "103" for ln in classed), classed
    # stale or missing summaries
    (tmp_path / "r09_pmc_traffic_per_launch.json").write_text(json.dumps({"some_other_kernel(": {"launches": 9, "hbm_read_MB_corrected_x2": 1.0,
                                                                                                  "hbm_write_MB": 1.0}}))
    t, d, err = bench.lookup_traffic(name, profiles_dir=str(tmp_path))
    assert t is None and d is None and "no entry for kernel symbol" in err
    t, d, err = bench.lookup_traffic("a brand-new kernel", profiles_dir=str(tmp_path))
    assert t is None and "no kernel symbol known" in err
    t, d, err = bench.lookup_traffic(name, profiles_dir=str(tmp_path / "empty"))
    assert t is None and "no profiles/" in err


def test_roofline_traffic_lookup_for_the_transformer_line():
    """The Transformer lines name the decoder's cross-attention; its launches shrink with the live rows, so the lookup cites the LARGEST
    measured grid together with its row count -- and that figure must agree with the algorithmic bytes of the same launch: every byte of a
    live row's encoder states (default form) or keys / values (the reference's form) is read exactly once per layer and step."""
    import bench
    for launch, symbol in (("dec_xattn", "dec_xattn_kernel<24>"), ("dec_cross_attn", "dec_attn_kernel<false")):
        traffic, detail, err = bench.lookup_traffic(launch)
        assert err is None and detail["kernel_symbol"].startswith(symbol)
        rows = detail["live_rows_of_that_launch"]
        assert rows >= 256
        assert 0.95 <= traffic / detail["algorithmic_bytes_of_that_launch"] <= 1.10, (launch, rows, traffic, detail)
