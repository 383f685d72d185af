"""GPU suite: the FILE loader at the metric's own size.  The bench's 6.45e9-k-mer table (HTSIZE 1610612741, k = 31) is
written in the reference's on-disk format (.sz/.ky/.lb, 41 GB: src/hashTable_hh.hh:473-546) to a directory with room
for it, loaded with mc_load_db (CuClarkDB::read, src/CuClarkDB.cu:463-770: .ky of 25 GB, more than 2^32 k-mers through
the file stream, two passes over the files) and 40 000 reads are checked against the oracle, which reads the same
files on its own.  Skipped, with the reason, on a box without 60 GB of file space (tmpfs or disk)."""
import os
import shutil
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HT, K, T = 1610612741, 31, 4096


def _dir_with(free_bytes):
    for d in ("/dev/shm", os.environ.get("TMPDIR") or "/tmp", "/tmp"):
        try:
            if os.path.isdir(d) and shutil.disk_usage(d).free > free_bytes:
                return d
        except OSError:
            pass
    return None


def test_full_size_table_from_files_equals_the_oracle(oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from jn_cuclark_amd import CuClarkDB, synth_gpu
    d = _dir_with(60 << 30)
    if d is None:
        pytest.skip("no directory with 60 GB free for the 41 GB of database files")
    work = os.path.join(d, "mc_filesize_%d" % os.getpid())
    os.makedirs(work)
    try:
        dev = torch.device("cuda", 0)
        base = os.path.join(work, "db_central_k%d_t%d_s%d_m0.tsk" % (K, T, HT))
        genomes = synth_gpu.make_genomes(T, 100_000, seed=31, device=dev)
        ranges = [(HT * j // 16, HT * (j + 1) // 16) for j in range(16)]

        def chunks():
            for b0, b1 in ranges:
                d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, K, HT, T, 3.75, genomes=genomes, shard=(b0, b1))
                yield d_sz, d_keys, d_labels, b0, b1

        n_keys, nbytes = synth_gpu.write_db_files(base, chunks())
        assert n_keys > 6_000_000_000 and os.path.getsize(base + ".ky") == 4 * n_keys > 2 ** 32
        assert os.path.getsize(base + ".sz") == HT and os.path.getsize(base + ".lb") == 2 * n_keys
        torch.cuda.empty_cache()
        n = 40_000
        rp, con, truth = synth_gpu.make_reads(genomes, n, 150, seed=78, return_truth=True)
        t0 = time.time()
        with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
            assert db.read(base) is True
            load_s = time.time() - t0
            info = db.db_info()
            fin_t = torch.zeros((n, 5), dtype=torch.int16, device=dev)
            db.query_device(rp, con, final_t=fin_t, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        print("mc_load_db of %.1f GB of files: %.1f s" % (nbytes / 1e9, load_s))
        assert info["index_kind"] == 1 and info["index_fallback"] == 0 and info["n_keys"] == n_keys == info["n_keys_owned"]
        assert info["shard_begin"] == 0 and info["shard_end"] == HT
        fin = fin_t.cpu().numpy().view(np.uint16)
        odb = oracle.OracleDB.load(base, HT, 4)
        want, _ = odb.classify(K, rp.cpu().numpy().view(np.uint32), con.cpu().numpy().view(np.uint16), 15)
        odb.close()
        assert np.array_equal(fin, want)
        tr = truth.cpu().numpy()
        assert (fin[: tr.size, 1] == tr + 1).mean() > 0.995
    finally:
        shutil.rmtree(work, ignore_errors=True)
