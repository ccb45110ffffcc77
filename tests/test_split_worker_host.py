"""Plumbing of the per-rank split-search workers (image-processing_amd/crim_split_worker.py) on
CPU: child launch, ready / go / id / result files, time-outs.  --dry-run keeps the GPU and RCCL out
of it; what a real run adds is covered by the GPU test of the RCCL path with one rank."""
import os

import crim_split_worker as sw


def test_two_dry_run_workers_exchange_the_id_and_report(tmp_path):
    rdv = str(tmp_path / "rdv")
    procs = [sw.launch_worker(r, 2, 0, rdv, size=64, iters=5, dry_run=True) for r in range(2)]
    for r in range(2):
        assert sw.wait_for_file(os.path.join(rdv, "ready.%d" % r), 30.0) == b"ready"
    assert not os.path.exists(os.path.join(rdv, "id"))          # nothing starts before the go file
    sw.release(rdv)
    res = sw.collect(rdv, 2, 30.0)
    for p in procs:
        assert p.wait(timeout=30) == 0
    assert all(r is not None and "error" not in r for r in res)
    assert res[0]["id_sha1"] == res[1]["id_sha1"]
    assert [r["rank"] for r in res] == [0, 1] and all(r["world"] == 2 for r in res)
    s = sw.summarise(res)
    assert s["all_ranks_agree_on_the_trace"] and s["rccl_ranks_seen"] == [2, 2]


def test_worker_without_go_file_times_out_and_says_so(tmp_path):
    rdv = str(tmp_path / "rdv")
    os.makedirs(rdv)
    import subprocess
    import sys
    p = subprocess.run([sys.executable, sw.__file__, "--rank", "1", "--world", "2", "--rendezvous", rdv,
                        "--dry-run", "--wait", "0.3"], capture_output=True, timeout=30)
    assert p.returncode == 3
    res = sw.collect(rdv, 2, 0.5)
    assert res[0] is None and "no go file" in res[1]["error"]
    assert "error" in sw.summarise(res)


def test_missing_rank_is_reported(tmp_path):
    rdv = str(tmp_path / "rdv")
    p = sw.launch_worker(0, 1, 0, rdv, dry_run=True)
    sw.release(rdv)
    assert p.wait(timeout=30) == 0
    res = sw.collect(rdv, 2, 0.5)
    assert res[0] is not None and res[1] is None
    assert sw.summarise(res)["error"].startswith("1 of 2")
