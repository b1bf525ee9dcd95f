"""Dev tool (GPU box): what one candidate-search call of the pipeline costs -- one thread, calls of 806 and 8 060 sub-reads of 256
bases over the k-mer table of a 2 Mbp reference with repeat families (a CS thread of ngmlr_hip_all sends 806 per call), through
cvx_search_batch_arena; then the call's trace (CVX_SEARCH_TRACE) as a small call takes it and sorted by map size.
CVX_LIB=<another build of libcvxalign.so> for an A/B on one box."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from ngmlr_amd import capi, synth
from ngmlr_amd.aligner import ConvexAlignHip, KmerIndex
lib = os.environ.get("CVX_LIB")
al = ConvexAlignHip(device=0)
contigs = synth.big_reference(2 << 20, n_contigs=1)
idx5, locs, starts = synth.kmer_table(al.lib, contigs)
for n in (806, 8060):
    reads = synth.sample_subreads(contigs, n)
    ix = KmerIndex(al, 13, idx5.view(np.dtype([("tab", "<u4"), ("rc", "i1")])), locs, 0)
    arena, offsets, pinned = KmerIndex.make_arena(reads, al.lib)
    ncand, begin, cands, mh, ms = ix.search_arena(arena, offsets)
    cbuf = np.zeros(len(cands) * 2 + 64, dtype=cands.dtype)
    ts = []; ks = []
    for _ in range(30):
        t0 = time.perf_counter(); ix.search_arena(arena, offsets, cands=cbuf); ts.append(time.perf_counter() - t0); ks.append(al.stage_kernel_ms(capi.STAGE_SEARCH))
    print("%s: %d reads per call: median call %.3f ms (min %.3f), kernels %.3f ms, candidates %d" % (al.lib._name.split('/')[-1], n, 1e3 * sorted(ts)[15], 1e3 * min(ts), sorted(ks)[15], int(ncand[ncand > 0].sum())))
    ix.free()
os.environ["CVX_SEARCH_TRACE"] = "1"
reads = synth.sample_subreads(contigs, 806)
ix = KmerIndex(al, 13, idx5.view(np.dtype([("tab", "<u4"), ("rc", "i1")])), locs, 0)
arena, offsets, pinned = KmerIndex.make_arena(reads, al.lib)
for _ in range(3): ix.search_arena(arena, offsets, cands=cbuf)
os.environ["CVX_TUNE_SEARCH_CLASSIFY"] = "0"
for _ in range(3): ix.search_arena(arena, offsets, cands=cbuf)
