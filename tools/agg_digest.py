"""Digest of the aggregation kernels' outputs on a dense graph (long CSR runs): run once with TMPNN_AGG=0 (round-1
kernels) and once with the pipelined forms -- the digests must be identical (same order of additions)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trackmpnn_amd import _lib, dense_static_graph
dev = 'cuda:0'
hs = hashlib.sha256()
for T, D, H in ((6, 40, 64), (4, 70, 128), (5, 9, 32)):
    g = dense_static_graph(T, D).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    h = torch.randn(g.N, H, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    es = torch.zeros(g.N, H, device=dev)
    _lib.call('tmpnn_segsum_fwd', g.cref(), h.data_ptr(), H, es.data_ptr(), H, H, 0, 0, st)
    out = torch.zeros(g.N, H, device=dev)
    _lib.call('tmpnn_gather_diff_fwd', g.cref(), h.data_ptr(), H, out.data_ptr(), H, H, 0, st)
    adj = torch.ones(g.N, H, device=dev)
    _lib.call('tmpnn_gather_diff_bwd', g.cref(), h.data_ptr(), H, adj.data_ptr(), H, H, 1, st)
    for t in (es, out, adj):
        hs.update(t.cpu().numpy().tobytes())
print('AGG_DIGEST', hs.hexdigest())
