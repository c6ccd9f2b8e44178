"""CPU: bench.py's bookkeeping that must not depend on a GPU -- the PMC traffic record is accepted only for the kernel
sources it was measured on, and `python bench.py --gpus N` without a launcher spawns its own ranks (here: they fail
loudly, there is no GPU and no CPU fallback) instead of asking for torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_traffic_record_is_bound_to_the_kernel_sources(tmp_path, monkeypatch):
    import bench
    have = bench.kernel_source_hashes(20)
    assert set(have) == set(bench.KERNEL_SOURCES["lpl"]) and set(bench.kernel_source_hashes(512)) == set(bench.KERNEL_SOURCES["row"])
    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = {"astroph-k20": {"phi_hbm_bytes_per_launch": 1.0e7, "sweep_hbm_bytes": 3.0e7, "commit": "abc", "source": "x", "source_hashes": have}}
    (prof / "traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    for rel in bench.KERNEL_SOURCES["lpl"]:            # the tree the hashes are taken from
        dst = tmp_path / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_bytes(open(os.path.join(ROOT, rel), "rb").read())
    tr, why = bench._traffic("astroph-k20", 20)
    assert tr is not None and "match" in why
    # one byte of one kernel source changes: the record is refused, and the reason names the file
    f = tmp_path / bench.KERNEL_SOURCES["lpl"][0]
    f.write_bytes(f.read_bytes() + b"\n")
    tr, why = bench._traffic("astroph-k20", 20)
    assert tr is None and "svils_lpl.hip" in why and "refused" in why
    # ... and the line then falls back to the pull model, never above what the counter would have allowed to claim
    rec0 = {"launches_timed": 10, "avg_launch_us": 23.0, "links_in_timed_launches": {"dense": 1750000, "sparse": 0, "shortcut": 200000},
            "achieved": 4000.0, "frac": 0.5}
    out = bench._roofline_fields(rec0, 20, 17903, "astroph-k20")
    assert out["frac_basis"] == "pull_model" and out["traffic"] is None and out["frac"] == out["pull_model"]["frac"] <= 1.0
    assert out["frac_survey_model"] == 0.5 and "refused" in out["traffic_source"]["refused"]
    assert bench._traffic("no-such-workload", 20)[0] is None


def test_committed_traffic_record_matches_this_tree():
    """the record bench.py will be asked to use at the end of the round was measured on these kernel sources"""
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for wl, k in (("astroph-k20", 20), ("mmsb:1000000:512:24", 512), ("synthetic:200000:512:24", 512)):
        assert t[wl]["source_hashes"] == bench.kernel_source_hashes(k), "%s: re-run `tools/evidence.sh <tag> pmc` and copy traffic.json" % wl
        assert os.path.exists(os.path.join(ROOT, t[wl]["source"]))


def test_bare_gpus_n_spawns_ranks_and_fails_loudly_without_a_gpu(tmp_path):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: the bare launch is covered by tests/test_gpu_native_ranks.py")
    except ImportError:
        pass
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0
    # both ranks were started and said why they left (the launcher gives the second one a few seconds to do so before it ends it)
    assert 1 <= r.stderr.count("needs a GPU (no CPU fallback)") <= 2
    assert "torch.distributed.run" not in r.stderr


def test_side_records_config5_ksharded_first_and_timeouts_scale():
    """on the first real node the record of the one layout expected to scale (config 5, K-sharded) must not
    sit behind six others under a global cut-off: it runs FIRST, shares its host set-up with the node-block record of the
    same workload, and the cut-off scales with the records requested."""
    import bench
    plan, budget = bench.side_record_plan("")
    names = [r[0] for r in plan]
    assert names[0] == "ksharded_config5_mmsb_n1m_k512" and names[1] == "config5_mmsb_n1m_k512" and len(names) == 7
    assert plan[0][1] == plan[1][1] == bench.CONFIG5_WORKLOAD                      # adjacent: one set-up serves both
    assert budget >= 400          # (the default --extra-timeout caps it at 300 s: the line must not wait for every record)
    one, b1 = bench.side_record_plan("config4_astroph_k200")
    assert [r[0] for r in one] == ["config4_astroph_k200"] and b1 < 120
    # an --extra-list keeps the canonical order whatever order it names them in
    two, _ = bench.side_record_plan("config4_astroph_k200,ksharded_config5_mmsb_n1m_k512")
    assert [r[0] for r in two] == ["ksharded_config5_mmsb_n1m_k512", "config4_astroph_k200"]


def test_model_prediction_rows_cover_every_n_gt_1_record():
    """the cost model's number travels in the same JSON line as the record it will be compared with: profiles/shard_cost_model.json
    (tools/shard_cost.py --json, measured on one GPU) has a row for the headline workload and for every side record at 2, 4, 8"""
    import bench
    m = json.load(open(os.path.join(ROOT, "profiles", "shard_cost_model.json")))
    assert m["link_model"]["eff_GBps"] > 0 and "lat_us" in m["link_model"]
    for world in (2, 4, 8):
        assert bench.model_prediction("astroph-k20", "nodeblock", world)["predicted_ms_per_step"] > 0
        for name, wl, _, layout in bench.SIDE_RECORDS:
            if layout == "steps":
                continue
            mp = bench.model_prediction(wl, layout, world)
            assert mp is not None and mp["predicted_ms_per_step"] > 0 and mp["compute_ms_per_rank"] > 0, (name, world)
            assert mp["source"].startswith("tools/shard_cost.py")
