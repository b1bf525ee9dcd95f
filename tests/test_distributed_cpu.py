"""CPU, world_size 2 over gloo: the N>1 path of the hot path is pure sharding -- every
rank aligns its own tiles, no collective carries DP data; only the control plane
(result dictionaries, throughput counters) crosses ranks.  The per-rank compute here is
the CPU checker standing in for the GPU (this box has none); the GPU version of the same
flow is bench.py --gpus N.  What of the PRODUCT can run without a device does run on every rank: the
library's host stage of a shard -- the closed form of every corridor (cvx_corridor_fit_batch over the shard's tile table, what
the binding does before it submits) -- and its output crosses the ranks with the rest."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    from ngmlr_amd import synth
    from ngmlr_amd.shard import gather_results, shard_tiles
    from oracle.pyoracle import Oracle
    tiles = synth.workload_ont(24, seed=42, max_len=1500)       # identical list on every rank
    parts = shard_tiles([t.cells for t in tiles], world)
    orc = Oracle("port")
    local = {i: (orc.align(tiles[i], want_nm=False)["cigar"], orc.align(tiles[i], want_nm=False)["score_bits"]) for i in parts[rank]}
    # the library's host stages on this rank's shard: corridors -> closed forms, in place on the shard's tile table
    import ctypes as C
    from ngmlr_amd import capi
    lib = capi.load()
    mine = synth.tileset_from_tiles([tiles[i] for i in parts[rank]])
    tab = mine.table().copy()
    nfit = C.c_int32(0)
    assert lib.cvx_corridor_fit_batch(len(tab), tab.ctypes.data, 2, C.byref(nfit)) == 0
    forms = {i: (int(tab["corridor_kind"][k]), float(tab["corridor_k"][k]), float(tab["corridor_d"][k]), int(tab["corridor_width"][k]))
             for k, i in enumerate(parts[rank])}
    local = {i: local[i] + (forms[i],) for i in parts[rank]}
    dist.barrier()
    merged = gather_results(local, world, rank, dist)
    bases = torch.tensor([float(sum(tiles[i].H for i in parts[rank]))], dtype=torch.float64)
    dist.all_reduce(bases, op=dist.ReduceOp.SUM)
    tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((sorted(merged.keys()), merged, float(bases.item()), float(tmax.item()), [len(p) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_alignment_equals_single_process(built):
    import torch.multiprocessing as mp
    from ngmlr_amd import synth
    from oracle.pyoracle import Oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, merged, bases, tmax, sizes = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tiles = synth.workload_ont(24, seed=42, max_len=1500)
    assert keys == list(range(24)) and sum(sizes) == 24 and min(sizes) > 0
    orc = Oracle("port")
    from ngmlr_amd import capi
    n_forms = 0
    for i, t in enumerate(tiles):
        a = orc.align(t, want_nm=False)
        assert merged[i][:2] == (a["cigar"], a["score_bits"])
        kind, k, d, w = merged[i][2]                               # what the owning rank's library call made of the tile's rows
        want = capi.corridor_fit(t.row_offset, t.row_length, t.W, t.H)
        assert (kind, k, d, w) == (want[0], want[1], want[2], want[5])
        n_forms += kind != capi.CORRIDOR_ROWS
    assert n_forms >= 20                                           # the ONT mix's corridors are the builders' own
    assert bases == float(sum(t.H for t in tiles))
    assert tmax == 2.0
