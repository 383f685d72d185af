"""Multi-GPU classification: one process per GPU over torch.distributed (backend "nccl"
is RCCL over xGMI on ROCm; "gloo" for the CPU rehearsal in tests/).

What the reference does (src/CuClarkDB.cu:552-559, :842-851, :909-928): the table is
split into contiguous bucket ranges, one per device; EVERY device receives EVERY read
batch and produces a partial sparse row per read; rows are pulled to device 0 with a
blocking cudaMemcpyPeer binary tree and merged there; top-2 runs on device 0.

What this module does instead (option 1): the same bucket-range
shards, but the combine step is a reduce-scatter by read range -- one all_to_all in
which rank j receives every rank's rows for reads [j*per, (j+1)*per), merges them and
runs top-2 for those reads.  xGMI is point-to-point, so all 7 links of a GPU carry 1/8
of the payload once, instead of log2(G) serialized hops into one GPU.  Rows are exact
integers, so the result is bit-identical to the unsharded run for any shard count
(tests/test_distributed.py, tests/test_gpu_parity.py::test_sharded_*).

`Replica` mode (table on every GPU, reads split) needs no exchange at all and is the
throughput-optimal choice whenever the table fits one GPU (288 GB): sharding divides
only the memory probes, every shard still encodes and hashes every k-mer.
"""
import torch
import torch.distributed as dist


def shard_range(htsize, rank, world):
    """contiguous bucket range of `rank` (the reference sizes ranges by free device
    memory, CuClarkDB.cu:552-559; identical GPUs get equal ranges)."""
    return (htsize * rank // world, htsize * (rank + 1) // world)


def read_range(n_reads, rank, world):
    per = (n_reads + world - 1) // world
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per), per


class ShardedClassifier:
    """`backend` provides, on this rank's device:
         row_len
         query_rows(reads_ptr, containers, n_reads) -> int16 tensor [n_reads, row_len]
         merge_rows(a, b, n)      a <- union(a, b)   (first n rows)
         result_rows(rows, n) -> int16 tensor [n, 5]
    jn_cuclark_amd.dist.HipBackend wraps CuClarkDB; tests use an oracle-based double."""

    def __init__(self, backend, group=None):
        self.be = backend
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def exchange(self, rows, n_reads):
        """all_to_all of partial rows: returns [world, per, row_len], slice j = rank j's
        partial rows for MY read range."""
        W, L = self.world, self.be.row_len
        lo, hi, per = read_range(n_reads, self.rank, W)
        send = torch.zeros((W * per, L), dtype=torch.int16, device=rows.device)
        send[:n_reads] = rows[:n_reads]
        # bytes on the wire: neither RCCL nor gloo has a 16-bit integer type
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on a box without RCCL peers: stage through the host
            s8 = send.view(torch.uint8).view(-1).cpu()
            r8 = torch.empty_like(s8)
            dist.all_to_all_single(r8, s8, group=self.group)
            recv = r8.to(send.device).view(torch.int16).view(W * per, L)
        else:
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv.view(torch.uint8).view(-1), send.view(torch.uint8).view(-1), group=self.group)
        return recv.view(W, per, L), (lo, hi, per)

    def classify(self, reads_ptr, containers, n_reads):
        """Every rank passes the SAME batch.  Returns (final int16 [hi-lo, 5], (lo, hi)):
        the final rows of this rank's read range."""
        rows = self.be.query_rows(reads_ptr, containers, n_reads)
        recv, (lo, hi, per) = self.exchange(rows, n_reads)
        mine = hi - lo
        acc = recv[0]
        for j in range(1, self.world):           # fixed order: integer adds, any order is exact
            self.be.merge_rows(acc, recv[j], mine)
        return self.be.result_rows(acc, mine), (lo, hi)

    def classify_gathered(self, reads_ptr, containers, n_reads):
        """As classify(), then all_gather so every rank holds all final rows."""
        fin, (lo, hi) = self.classify(reads_ptr, containers, n_reads)
        _, _, per = read_range(n_reads, self.rank, self.world)
        pad = torch.zeros((per, 5), dtype=torch.int16, device=fin.device)
        pad[: hi - lo] = fin
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather([o.view(torch.uint8) for o in out], pad.view(torch.uint8), group=self.group)
        return torch.cat(out)[:n_reads]


class HipBackend:
    """ShardedClassifier backend over a CuClarkDB holding this rank's bucket range."""

    def __init__(self, db, device):
        self.db = db
        self.device = device
        self.row_len = db.row_len

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def query_rows(self, reads_ptr, containers, n_reads):
        rows = torch.empty((n_reads, self.row_len), dtype=torch.int16, device=self.device)
        self.db.query_device(reads_ptr, containers, rows_t=rows, stream=self._stream())
        return rows

    def merge_rows(self, a, b, n):
        self.db.merge_rows_device(a, b, a, n, stream=self._stream())

    def result_rows(self, rows, n):
        fin = torch.empty((n, 5), dtype=torch.int16, device=self.device)
        self.db.result_rows_device(rows, fin, n, stream=self._stream())
        return fin
