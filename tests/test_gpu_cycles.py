"""GPU suite: database cycles (a table larger than all devices together: the reference's swapDbParts loop, CuClarkDB.cu:775-815,
src/CuCLARK_hh.hh:1765-1772) at BASELINE configs[1]'s size -- the cuCLARK-l table (HTSIZE 57777779, k = 27, ~6.3e8 k-mers of 2048
targets) from its .sz/.ky/.lb files, 1 M reads of a FASTQ file through bin/cuCLARK-l: once against the resident table on one
device, once with two members on the card that hold 2 of 6 parts at a time (MC_GROUP_CYCLES=3).  The CSV must be byte-identical
(the toy-size cases, with the oracle as the judge, are in tests/test_host_cli.py and tests/test_ref_host.py)."""
import hashlib
import os
import shutil

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HT, K, T = 57777779, 27, 2048


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def test_six_parts_two_at_a_time_give_the_resident_table_s_csv(tmp_path, monkeypatch):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from jn_cuclark_amd import synth_gpu
    exe = os.path.join(ROOT, "bin", "cuCLARK-l")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    work = "/dev/shm/mc_test_cycles" if os.path.isdir("/dev/shm") else str(tmp_path / "w")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    try:
        dev = torch.device("cuda:0")
        genomes = synth_gpu.make_genomes(T, 14_000, seed=21, device=dev)
        base = os.path.join(work, "db_central_k%d_t%d_s%d_m0_light_4.tsk" % (K, T, HT))
        ranges = [(HT * j // 4, HT * (j + 1) // 4) for j in range(4)]

        def chunks():
            for b0, b1 in ranges:
                d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 21, K, HT, T, 10.4, genomes=genomes, shard=(b0, b1))
                yield d_sz, d_keys, d_labels, b0, b1

        n_keys, _ = synth_gpu.write_db_files(base, chunks())
        assert 5.5e8 < n_keys < 7e8
        n = 1_000_000
        fq = os.path.join(work, "reads.fq")
        truth = synth_gpu.write_fastq(fq, genomes, n, seed=93).numpy()
        del genomes
        torch.cuda.empty_cache()
        digest = {}
        for tag, env in (("resident", {}), ("cycled", {"MC_GROUP_DEVICES": "0,0", "MC_GROUP_CYCLES": "3"})):
            for k_, v in env.items():
                monkeypatch.setenv(k_, v)
            r = synth_gpu.host_driver_run(exe, work, K, T, fq, n, threads=8, batches=11, truth=truth, timeout=900)
            for k_ in env:
                monkeypatch.delenv(k_)
            assert r["csv_lines"] == n and r["assigned_to_their_genome"] > 0.99 * r["checked"], r
            assert any("database cycle 2 of 3" in t for t in r["timing"]) == (tag == "cycled"), r["timing"]
            digest[tag] = _sha(os.path.join(work, "res.csv"))
        assert digest["cycled"] == digest["resident"]
    finally:
        shutil.rmtree(work, ignore_errors=True)
