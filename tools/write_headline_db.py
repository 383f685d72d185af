"""tools/write_headline_db.py <dir> -- the bench's headline table (HTSIZE 1610612741, k = 31, 6.45e9 k-mers of 4096 targets) written
as .sz/.ky/.lb files under <dir> (41 GB; run on the GPU box), for tools/load_time.py and friends.  Prints the base path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jn_cuclark_amd import synth_gpu

K, HT, T, LAM, GLEN = 31, 1610612741, 4096, 3.75, 100_000
d = sys.argv[1]
os.makedirs(d, exist_ok=True)
dev = torch.device("cuda", 0)
genomes = synth_gpu.make_genomes(T, GLEN, seed=31, device=dev)
base = os.path.join(d, "db_central_k%d_t%d_s%d_m0.tsk" % (K, T, HT))
ranges = [(HT * j // 16, HT * (j + 1) // 16) for j in range(16)]


def chunks():
    for b0, b1 in ranges:
        d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, K, HT, T, LAM, genomes=genomes, shard=(b0, b1))
        yield d_sz, d_keys, d_labels, b0, b1


n_keys, nbytes = synth_gpu.write_db_files(base, chunks())
print(base, n_keys, nbytes, flush=True)
