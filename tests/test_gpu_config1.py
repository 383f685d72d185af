"""GPU suite at BASELINE configs[1]'s own size: the cuCLARK-l table (HTSIZE 57777779, k = 27, MAXHITS 23,
~6.3e8 k-mers of 2048 targets: SURVEY.md 8d config 2) and 1 M x 150 bp reads -- EVERY read's sparse row and final
row against the oracle, on each in-HBM index (at this size the oracle classifies the whole batch in seconds;
bench.py --config 2 checks a sample of the same workload)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HT, K, T, MAXHITS = 57777779, 27, 2048, 23


@pytest.fixture(scope="module")
def world(oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from jn_cuclark_amd import synth_gpu
    dev = torch.device("cuda:0")
    genomes = synth_gpu.make_genomes(T, 14_000, seed=21, device=dev)
    raw = synth_gpu.build_db(dev, 21, K, HT, T, 10.4, genomes=genomes)
    n = 1_000_000
    rp, con = synth_gpu.make_reads(genomes, n, 150, seed=22)
    sz, ky, lb = (t.cpu().numpy() for t in raw)
    odb = oracle.OracleDB.from_arrays(HT, sz, ky.view(np.uint32), lb.view(np.uint16))
    oracle.set_num_threads(oracle.host_cores())
    want_rows, _ = odb.query_rows(K, rp.cpu().numpy().view(np.uint32), con.cpu().numpy().view(np.uint16), MAXHITS)
    odb.close()
    return dev, raw, (rp, con, n), want_rows


@pytest.mark.parametrize("index", ["minimizer", "lines"])
def test_config1_light_table_one_million_reads_every_row_equals_the_oracle(world, oracle, index, monkeypatch):
    import torch
    from jn_cuclark_amd import CuClarkDB
    dev, raw, (rp, con, n), want_rows = world
    monkeypatch.setenv("MC_INDEX", index)
    assert 5.5e8 < raw[1].numel() < 7e8
    with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=MAXHITS) as db:
        db.read_device(*raw)
        info = db.db_info()
        assert info["index_kind"] == (1 if index == "minimizer" else 0) and info["n_keys"] == raw[1].numel()
        st = torch.cuda.current_stream().cuda_stream
        rows = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
        fin = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.query_device(rp, con, final_t=fin, rows_t=rows, stream=st)
        torch.cuda.synchronize()
        over = db.stats()["reads_over_maxhits"]
    assert np.array_equal(rows.cpu().numpy().view(np.uint16), want_rows)
    want = oracle.result_rows(want_rows)
    assert np.array_equal(fin.cpu().numpy().view(np.uint16), want)
    assert over == 0
    # the batch is not trivial: the genome-sampled half is assigned, the random half hits next to nothing
    assert (want[: n // 2, 1] > 0).mean() > 0.95 and int(want[n // 2:, 0].astype(np.int64).sum()) < 20000
