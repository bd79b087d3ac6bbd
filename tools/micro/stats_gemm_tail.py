"""What the in-launch statistics tree costs the two stats-epilogue GEMMs of a block (za, z2) at P30 x 256, d = 384: the launch with the
statistics epilogue against the plain product with and without the addend, isolated, rotating operands (bench.time_kernel).
profiles/r06_stats_tree_tail.txt holds a run of it beside a timing-only build whose epilogue skips the tree."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/micro/ -> repo root
sys.path.insert(0, ROOT)
import bench
from graphgps_amd import norm as _norm
from graphgps_amd.gemm import absmax, gemm_panel, gemm_panel_stats, split_weights, amax_records
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
M, d = 7569, 384
class Owner: pass
sync = _norm.sync_arena(Owner(), dev)
for (K, name) in ((384, "za  = x + drop(o W_O)  K=384"), (768, "z2  = h + drop(t W_2)  K=768")):
    nset = 6
    A = [torch.randn(M, K, device=dev) for _ in range(nset)]
    add = [torch.randn(M, d, device=dev) for _ in range(nset)]
    w = torch.randn(d, K, device=dev) / K ** 0.5
    b = torch.randn(d, device=dev)
    (img, _), = split_weights([w], tn=False, f16=True)
    recs = amax_records(nset, dev)
    for i in range(nset): absmax([A[i]], out=recs[i:i+1])
    bn = torch.nn.BatchNorm1d(d).to(dev)
    mean, rstd = torch.empty(d, device=dev), torch.empty(d, device=dev)
    desc = _norm.bn_desc(bn, mean, rstd)
    site = sync.site(0)
    t_stats = bench.time_kernel(lambda i: gemm_panel_stats(A[i], img, d, b, add[i], 0.1, 1234, desc, site, a_amax=recs[i]), iters=40, nsets=nset)
    outs = [torch.empty(M, d, device=dev) for _ in range(nset)]
    t_plain = bench.time_kernel(lambda i: gemm_panel(A[i], img, d, bias=b, addend=add[i], out=outs[i], a_amax=recs[i]), iters=40, nsets=nset)
    t_nocin = bench.time_kernel(lambda i: gemm_panel(A[i], img, d, bias=b, out=outs[i], a_amax=recs[i]), iters=40, nsets=nset)
    print(f"{name}: stats epilogue {t_stats*1e3:.2f} us | plain + addend {t_plain*1e3:.2f} us | plain {t_nocin*1e3:.2f} us")
