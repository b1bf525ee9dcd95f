import sys, ctypes, time, numpy as np
sys.path.insert(0, '.')
from ngmlr_amd import capi, synth
from ngmlr_amd.aligner import ConvexAlignHip, KmerIndex
al = ConvexAlignHip(device=0)
contigs = synth.big_reference(512 << 20, n_contigs=8)
idx5, locs, starts = synth.kmer_table(al.lib, contigs)
reads = synth.sample_subreads(contigs, 100000)
ix = KmerIndex(al, 13, idx5.view(np.dtype([("tab", "<u4"), ("rc", "i1")])), locs, 0)
arena, offsets, pinned = KmerIndex.make_arena(reads, al.lib)
ix.search_arena(arena, offsets)
out = (ctypes.c_ulonglong * 16)()
al.lib.cvx_debug_search_prof(out)
names = ["chunks", " lookup", " fetch(binsearch+issue)", " bin wait", " cast_batch", "  probe", "  claim+verify", "  dups", "  score+scan+writes", "batches", "total", "reads", "lookup-issue"]
for rep in range(2):
    t0 = time.perf_counter(); ix.search_arena(arena, offsets); dt = time.perf_counter() - t0
    al.lib.cvx_debug_search_prof(out)
    v = list(out); n = max(v[11], 1)
    print("call %.2f ms kernels %.2f ms; per read:" % (dt * 1e3, al.stage_kernel_ms(capi.STAGE_SEARCH)), ", ".join("%s %.0f" % (names[z], v[z] / n) for z in range(13)))
