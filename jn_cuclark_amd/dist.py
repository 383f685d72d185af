"""Multi-GPU classification with one process per GPU over torch.distributed (backend "nccl" is RCCL over
xGMI on ROCm; "gloo" for the CPU rehearsal in tests/).  The one-process form of the same protocol, for the
host driver's `-d N`, is mc_group (include/mc_group.h).

What the reference does (src/CuClarkDB.cu:552-559, :842-851, :909-928): the table is split into contiguous
bucket ranges, one per device; EVERY device receives EVERY read batch and produces a partial sparse row per
read; rows are pulled to device 0 with a blocking cudaMemcpyPeer binary tree, merged pairwise there; top-2
runs on device 0.

What this module does instead:
  * parts are LINE ranges of the minimizer index (mc_load_db_part / CuClarkDB.read_part): the k-mers of one
    minimizer run of a read live on ONE GPU, so a part fetches, matches and scores 1/G of the runs.  Bucket
    ranges (the reference's partition, still accepted) scatter a run over all GPUs and every GPU fetches
    nearly every line (tools/shard_rate.py);
  * the combine step is a reduce-scatter by read range (option 1): one all_to_all in
    which rank j receives every rank's rows for reads [j*per, (j+1)*per) -- xGMI is point-to-point, so all 7
    links of a GPU carry 1/G of the rows once, instead of log2(G) serialized hops into one GPU -- then ONE
    k-way merge + top-2 launch (mc_merge_result_device);
  * a batch is cut into chunks: the exchange and merge of chunk i run on a second stream while the query
    kernel works on chunk i+1; send and receive buffers are allocated once.
Rows are exact integers, so the result is bit-identical to the unsharded run for any number of parts
(tests/test_distributed.py, tests/test_gpu_index.py, tests/test_gpu_dist.py).

`Replica` mode (table on every GPU, reads split) needs no exchange at all and is the throughput-optimal
choice whenever the table fits one GPU (288 GB): every part still encodes every read and finds its
minimizers.  In between: a table that needs S cards is cut into S parts -- as few as memory dictates, the
reference's minParts (src/CuClarkDB.cu:529-559) -- and the world forms G = world // S groups that each hold
the whole table and classify their own batches (plan_shards, shard_groups below): the exchange stays inside
a group, and only S, not world, GPUs repeat the front half of the kernel for a read.
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


def plan_shards(n_keys_total, world, hbm_bytes):
    """(S, G): parts the table is cut into and groups of S ranks that each hold all of it.  S = the smallest part
    count whose share fits `hbm_bytes` per GPU at an acceptable fill (mc_index_plan: the arithmetic mc_group_load_db
    uses), spread evenly: G = world // S_min groups of S = world // G ranks; ranks past S * G stay idle.  A table
    that fits no way gets one group of `world` parts (the load will say so)."""
    from . import _lib
    s_min = _lib.index_plan(n_keys_total, 1, hbm_bytes)["min_parts"]
    if s_min < 1 or s_min > world:
        return world, 1
    G = world // s_min
    return world // G, G


def shard_groups(n_shards, world=None, rank=None):
    """Process groups of `n_shards` consecutive ranks.  EVERY rank of the world must call this (new_group is
    collective).  Returns (group, group_index, part, n_groups); group is None for a rank past the last full group."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    G = max(1, world // n_shards)
    mine = None
    for g in range(G):
        h = dist.new_group(ranks=list(range(g * n_shards, (g + 1) * n_shards)))
        if g * n_shards <= rank < (g + 1) * n_shards:
            mine = h
    gi = rank // n_shards
    return (mine, gi, rank % n_shards, G) if gi < G else (None, -1, -1, G)


def chunk_bounds(n_reads, n_chunks):
    n_chunks = max(1, min(n_chunks, n_reads)) if n_reads else 1
    return [(n_reads * c // n_chunks, n_reads * (c + 1) // n_chunks) for c in range(n_chunks)]


class ShardedClassifier:
    """`backend` provides, on this rank's device:
         row_len, device
         query_rows_into(reads_ptr, containers, r0, r1, out)   rows of reads [r0, r1) of the batch -> out[:r1-r0]
         merge_result(srcs, n)  ->  int16 tensor [n, 5]        k-way merge of the row tensors in `srcs` + top-2
         stream handling (GPU only): current_stream / side_stream
    jn_cuclark_amd.dist.HipBackend wraps CuClarkDB; tests use an oracle-based double on the CPU."""

    def __init__(self, backend, group=None, n_chunks=4):
        self.be = backend
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_chunks = n_chunks
        self._bufs = {}          # (slot, per) -> (send, recv)
        self._gloo = dist.get_backend(group) == "gloo"
        self._check_layout()

    def _check_layout(self):
        """The parts of one table must have been built with ONE layout: line_of(K, lines per part) is part of the
        address of every k-mer, so a rank whose part came out at another fill (another MC_MZ_FILL, a card of another
        size, a card that held something else) would answer for lines nobody asks it about -- silently.  Compared
        across the group before the first batch; a mismatch is an error, not a slower run."""
        lay = getattr(self.be, "layout", None)
        if lay is None or self.world == 1:
            return
        mine = lay()
        all_ = [None] * self.world
        dist.all_gather_object(all_, mine, group=self.group)
        # (bucket-range shards -- n_parts 1 -- are indexes of their own, each with its own line count)
        by_lines = all_[0][1] > 1
        same = all(a[:2] == all_[0][:2] and (not by_lines or a[2] == all_[0][2]) for a in all_)
        parts = sorted(a[3] for a in all_)
        if not same or (by_lines and parts != list(range(self.world))):
            raise RuntimeError("the parts of the sharded table disagree on their layout (index kind, parts, lines per part, part): %r "
                               "-- build every part with the same fill (MC_MZ_FILL) on cards of one size" % (all_,))

    def _buffers(self, slot, per):
        key = (slot, per)
        if key not in self._bufs:
            W, L = self.world, self.be.row_len
            send = torch.zeros((W * per, L), dtype=torch.int16, device=self.be.device)
            recv = torch.zeros((W * per, L), dtype=torch.int16, device=self.be.device)
            self._bufs[key] = (send, recv)
        return self._bufs[key]

    def _all_to_all(self, send, recv):
        # bytes on the wire: neither RCCL nor gloo has a 16-bit integer type
        s8, r8 = send.view(torch.uint8).view(-1), recv.view(torch.uint8).view(-1)
        if send.is_cuda and self._gloo:
            # rehearsal on a box without RCCL peers: stage through the host
            hs = s8.cpu()
            hr = torch.empty_like(hs)
            dist.all_to_all_single(hr, hs, group=self.group)
            r8.copy_(hr)
        else:
            dist.all_to_all_single(r8, s8, group=self.group)

    def classify(self, reads_ptr, containers, n_reads):
        """Every rank passes the SAME batch.  Returns (final int16 [mine, 5], ranges): the final rows of the
        reads this rank owns -- for every chunk [c0, c1) of the batch the read range `rank` of it --
        concatenated in read order, and those ranges as (lo, hi) pairs."""
        W = self.world
        gpu = self.be.device.type == "cuda"
        outs, ranges = [], []
        if gpu:
            main = torch.cuda.current_stream(self.be.device)
            side = self.be.side_stream()
            side.wait_stream(main)
        for ci, (c0, c1) in enumerate(chunk_bounds(n_reads, self.n_chunks)):
            m = c1 - c0
            lo, hi, per = read_range(m, self.rank, W)
            send, recv = self._buffers(ci & 1, per)
            if gpu and ci >= 2:
                main.wait_event(self._done[ci & 1])              # the slot's previous exchange has read `send`
            # rows of the chunk straight into the send layout (row i of the chunk = read c0 + i)
            self.be.query_rows_into(reads_ptr, containers, c0, c1, send)
            if gpu:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                ctx = torch.cuda.stream(side)
            else:
                ctx = _Null()
            with ctx:
                self._all_to_all(send, recv)
                mine = hi - lo
                srcs = [recv[j * per: j * per + mine] for j in range(W)]   # rank j's rows for MY reads of the chunk
                outs.append(self.be.merge_result(srcs, mine))
                if gpu:
                    done = torch.cuda.Event()
                    done.record(side)
                    if not hasattr(self, "_done"):
                        self._done = [None, None]
                    self._done[ci & 1] = done
            ranges.append((c0 + lo, c0 + hi))
        if gpu:
            main.wait_stream(side)
        fin = torch.cat(outs) if outs else torch.zeros((0, 5), dtype=torch.int16, device=self.be.device)
        return fin, ranges

    def classify_gathered(self, reads_ptr, containers, n_reads):
        """As classify(), then all_gather so every rank holds all final rows in read order."""
        fin, ranges = self.classify(reads_ptr, containers, n_reads)
        W = self.world
        per_max = sum(read_range(c1 - c0, 0, W)[2] for c0, c1 in chunk_bounds(n_reads, self.n_chunks))
        pad = torch.zeros((per_max, 5), dtype=torch.int16, device=fin.device)
        pad[: fin.shape[0]] = fin
        out = [torch.empty_like(pad) for _ in range(W)]
        if fin.is_cuda and self._gloo:
            host = [o.cpu().view(torch.uint8) for o in out]
            dist.all_gather(host, pad.cpu().view(torch.uint8), group=self.group)
            out = [h.view(torch.int16).to(fin.device) for h in host]
        else:
            dist.all_gather([o.view(torch.uint8) for o in out], pad.view(torch.uint8), group=self.group)
        full = torch.zeros((n_reads, 5), dtype=torch.int16, device=fin.device)
        for j in range(W):
            at = 0
            for c0, c1 in chunk_bounds(n_reads, self.n_chunks):
                lo, hi, _ = read_range(c1 - c0, j, W)
                full[c0 + lo: c0 + hi] = out[j][at: at + hi - lo]
                at += hi - lo
        return full


def dense_allreduce_classify(backend, reads_ptr, containers, n_reads, num_targets, group=None):
    """The combine step as BASELINE words it: per-target hit VECTORS all-reduced (sum) across the parts, top-2 on
    the sum (SURVEY.md 8e option 2).  Every rank ends with all final rows.  n_reads x num_targets int32 per rank on
    the wire instead of 64 bytes per read, so this is for small target sets and serves as a cross-check of the
    sparse reduce-scatter above (tests/test_distributed.py, tests/test_gpu_dist.py); plain torch ops, no kernels
    of its own.  Exact while no read is over MAXHITS distinct targets (a sparse row cannot say more)."""
    L = backend.row_len
    rows = torch.zeros((max(n_reads, 1), L), dtype=torch.int16, device=backend.device)
    backend.query_rows_into(reads_ptr, containers, 0, n_reads, rows)
    rows = rows[:n_reads].to(torch.int32) & 0xFFFF
    dense = torch.zeros((n_reads, num_targets), dtype=torch.int32, device=backend.device)
    cnt = rows[:, 0]
    slots = torch.arange((L - 2) // 2, device=backend.device)[None, :]
    valid = slots < cnt[:, None]
    tgt = rows[:, 1::2][:, : (L - 2) // 2].to(torch.int64)
    hit = rows[:, 2::2][:, : (L - 2) // 2]
    dense.scatter_add_(1, torch.where(valid, tgt, torch.zeros_like(tgt)), torch.where(valid, hit, torch.zeros_like(hit)))
    if dense.is_cuda and dist.get_backend(group) == "gloo":
        h = dense.cpu()
        dist.all_reduce(h, group=group)
        dense = h.to(backend.device)
    else:
        dist.all_reduce(dense, group=group)
    dense.clamp_(max=65535)                                    # counts saturate
    best, ibest = dense.max(dim=1)                             # first maximum = smallest id (ascending scan, strict '>')
    rest = dense.clone()
    rest.scatter_(1, ibest[:, None], -1)
    second, isecond = rest.max(dim=1)
    fin = torch.zeros((n_reads, 5), dtype=torch.int32, device=backend.device)
    fin[:, 0] = dense.sum(dim=1) & 0xFFFF
    fin[:, 1] = torch.where(best > 0, ibest.to(torch.int32) + 1, torch.zeros_like(best))
    fin[:, 2] = best
    fin[:, 3] = torch.where(second > 0, isecond.to(torch.int32) + 1, torch.zeros_like(best))
    fin[:, 4] = second.clamp(min=0)
    return fin.to(torch.int16)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class HipBackend:
    """ShardedClassifier backend over a CuClarkDB holding this rank's part of the table."""

    def __init__(self, db, device):
        self.db = db
        self.device = torch.device(device)
        self.row_len = db.row_len
        self._side = None

    def side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def layout(self):
        """(index kind, parts, lines per part, this part) of the part held here: ShardedClassifier compares them across the group"""
        i = self.db.db_info()
        return (int(i["index_kind"]), int(i["n_parts"]), int(i["line_end"] - i["line_begin"]), int(i["part"]))

    def query_rows_into(self, reads_ptr, containers, r0, r1, out):
        # the offsets of reads [r0, r1] stay absolute: the kernel indexes the whole container array with them
        self.db.query_device(reads_ptr[r0: r1 + 1], containers, rows_t=out, stream=self._stream())

    def merge_result(self, srcs, n):
        fin = torch.empty((n, 5), dtype=torch.int16, device=self.device)
        if n:
            self.db.merge_result_device(srcs, n, final_t=fin, stream=self._stream())
        return fin
