"""TEST INFRASTRUCTURE: how much of the banded DP the sparse path (oracle/sparse_chain.hpp, shasta_amd/csrc/align4_sparse.hpp) would take on
bench-like reads, and that it never differs from the dense DP where it answers.  python scripts/sparse_census.py [reads] [candidates]
[anchored]: with a third argument also the anchored form (oracle/anchored_chain.hpp: dense DP only between the anchors where the optimal
chains differ) on every task, under the tie policy in ORACLE_TIE_POLICY."""
import numpy as np, time, sys, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from oracle import bindings
from shasta_amd import abi, synthetic
orc=bindings.OracleLib()
n_reads=int(sys.argv[1]) if len(sys.argv)>1 else 6000
toc,kmer=bench.make_workload(n_reads,12345)
data7=synthetic.pack_markers(toc,kmer)
t=time.time()
lh=orc.lowhash0(toc,data7,None,bench.lowhash_params(),threads=0)
cand=lh.candidates
print('reads',n_reads,'markers',int(toc[-1]),'candidates',len(cand),'%.1fs'%(time.time()-t))
sub=np.ascontiguousarray(cand[::max(1,len(cand)//int(sys.argv[2] if len(sys.argv)>2 else 20000))])
anchored=len(sys.argv)>3
orc.sparse_census(on=True,reset=True,anchored=anchored)
t=time.time()
out=orc.align4_batch(toc,data7,sub,bench.align_options(),want_ordinals=False,threads=0)
c=orc.sparse_census(on=False)
print('%.1fs'%(time.time()-t), json.dumps(c))
print('certified tasks %.1f%%, their share of dense cells %.1f%%; hits per task %.0f, scan steps per hit %.2f; dense cells per hit %.0f; aligned pairs per task %.0f' % (
  100*c['certified']/c['tasks'], 100*c['dense_cells_of_certified']/c['dense_cells'], c['hits']/c['tasks'], c['scan_steps']/max(1,c['hits']), c['dense_cells']/max(1,c['hits']), c['aligned_pairs']/c['tasks']))
if anchored:
    print('anchored form: %d tasks differ from the dense DP (must be 0); dense cells it solved %.1f%% of all; %d windows, %d tasks run whole; anchors per task %.0f; the largest rectangle %d cells, %d tasks with one of more than 16384, %d with an optimal link of a live hit more than 29 hits back, %d with more than 128 windows' % (
      c['anchored_different'], 100*c['anchored_dense_cells']/c['dense_cells'], c['anchored_windows'], c['anchored_whole_tasks'], c['anchors']/c['tasks'], c['anchored_largest_window'], c['anchored_tasks_with_a_window_over_16384'], c['ambiguous_tasks_with_a_live_link_over_29_hits'], c['anchored_tasks_with_over_128_windows']))
