import sys, time, itertools, torch
sys.path.insert(0, '/root/repo')
import bench, models
dev = torch.device('cuda:0')
T,L,C,D,dl,layers,Din,Nq,Hh,B = bench.WORKLOADS['activitynet_t256']
hosts=[]
gh = torch.Generator().manual_seed(7)
for s_ in range(2):
    hb = bench.make_batch(B, T, L, Nq, Din, seed=1000 + 17 * s_, device="cpu")
    dur = torch.rand(B, generator=gh) * 100 + 20
    ts = torch.rand(B, generator=gh) * dur * 0.5
    hosts.append(dict(video_features=hb["video_features"].pin_memory(), query_features=hb["query_features"].pin_memory(), nfeats=hb["video_mask"].sum((1, 2)),
                      qlen=hb["query_mask"].sum((1, 2)), times=torch.stack([ts, ts + 1.0 + torch.rand(B, generator=gh) * (dur - ts - 1.0)], 1), duration=dur))
feeder = models.vml_amd.BatchFeeder(T, L, Nq, dev)
# stage timing in isolation
slot = feeder.slots[0]
for k in range(3):
    t0=time.perf_counter(); feeder._stage(slot, hosts[k%2]); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print('stage host %.2f ms, +sync %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3))
it = feeder.feed(itertools.cycle(hosts))
t0=time.perf_counter()
for k in range(20):
    b = next(it)
torch.cuda.synchronize()
print('feeder alone: %.2f ms/batch' % ((time.perf_counter()-t0)/20*1e3))
