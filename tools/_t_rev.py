import torch, sys, time
sys.path.insert(0, '/root/repo')
import ogc_amd
from ogc_amd.fused import reverse_neighbours, neighbour_consistency
from ogc_amd.train_step import KITTI_LOSS, build_criterion
from ogc_amd.utils.synthetic import make_scene_batch
crit = build_criterion(KITTI_LOSS)
pcs, segms, flows, _ = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device='cuda')
pcs_l = [pcs[:, i].contiguous() for i in range(4)]
geo = crit.plan_geometry(pcs_l, True)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('plan total', timeit(lambda: crit.plan_geometry(pcs_l, True)))
for name in ('knn', 'ball'):
    idx = geo[name]
    print(name, idx.shape, 'reverse', timeit(lambda: reverse_neighbours(idx)))
    rs = geo[name + '_rev'][0]
    deg = (rs[:, 1:] - rs[:, :-1])
    print('  max in-degree', int(deg.max()), 'mean', float(deg.float().mean()))
    m = torch.rand(16, 8192, 10, device='cuda').softmax(-1).requires_grad_(True)
    print('  fwd', timeit(lambda: neighbour_consistency(m, idx, geo[name + '_rev'], 1)))
    out = neighbour_consistency(m, idx, geo[name + '_rev'], 1)
    go = torch.ones_like(out)
    print('  bwd', timeit(lambda: torch.autograd.grad(out, m, go, retain_graph=True)))
