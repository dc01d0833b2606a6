import sys, threading
sys.path.insert(0, '.')
import numpy as np, torch
from alaz_amd import engine, sharded, replay, weights
from tests.helpers import CLOCK, HostShim
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
topo = replay.make_topology(120, 1500, seed=91)
ev, labels = replay.make_events(topo, 60_000, seed=92, mixed=True, with_raw_outbound=True, with_reverse=True, fixed_labels=True)
W = weights.make_weights(layers)
def mk(**kw):
    g = engine.ServiceGraph(max_known_nodes=topo.n_nodes + 8, max_edges=8192, layers=layers, max_labels=128, max_outbound_ips=512, max_window_events=len(ev), **kw)
    g.set_clock(*CLOCK); g.load_weights(W); HostShim().apply(g, topo.k8s_ops()); g.set_label_count(len(labels)); return g
ref = mk(); ref.ingest(ev); want = ref.flush_window()
# staged unsharded for comparison
ref2 = mk(); ref2.ingest(ev); ref2.window_close(); ref2.window_features()
for l in range(layers): ref2.window_layer(l)
ref2.window_score(); want2 = ref2.window_read().copy()
print('fused vs staged unsharded equal:', want.tobytes() == want2.tobytes())
world = 2
shard = ref.route(ev, world)
shared = sharded.ThreadComm.Shared(world); dev = torch.device('cuda', 0)
ncap = topo.n_nodes + 8 + 128 + 512
engs, bes, outs = [], [], [None] * world
for r in range(world):
    g = mk(rank=r, world=world); g.ingest(ev[shard == r]); engs.append(g)
    bes.append(sharded.HipBackend(g, ncap=ncap, layers=layers, world=world, rank=r, device=dev, max_obip=512, stream=torch.cuda.Stream(dev)))
def run(r):
    sharded.run_window(bes[r], sharded.ThreadComm(shared, r)); outs[r] = engs[r].window_read().copy()
ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join() for t in ths]
got = np.concatenate(outs); key = lambda a: np.lexsort((a['to_ref'], a['from_ref']))
got = got[key(got)]; exp = want[key(want)]
print('rows', len(got), len(exp))
for f in got.dtype.names:
    if got[f].dtype.kind == 'f':
        d = np.abs(got[f] - exp[f]); print(f, 'max diff', d.max(), 'n diff', int((got[f] != exp[f]).sum()))
    else:
        print(f, 'n diff', int((got[f] != exp[f]).sum()))
bad = np.flatnonzero(got['score'] != exp['score'])[:10]
for i in bad:
    fr, to = int(got['from_ref'][i]), int(got['to_ref'][i])
    print(i, hex(fr), hex(to), got['score'][i], exp['score'][i], 'to type', to >> 30)
# which shard produced the bad rows, and is 'to' owned by it
print('total count unsharded', int(exp['count'].sum()), 'sharded', int(got['count'].sum()), 'per shard rows', [len(o) for o in outs], 'events', [int((shard==r).sum()) for r in range(world)])
for r in range(world):
    st = engs[r].stats(); print('shard', r, 'last_window_events', st.last_window_events, 'dropped_src', st.events_dropped_src, 'cap', st.events_dropped_cap, 'misrouted', st.events_misrouted, 'edges', st.last_window_edges, 'nodes', st.last_window_nodes)
st = ref.stats(); print('ref events', st.last_window_events, 'nodes', st.last_window_nodes)
dg = {(int(a['from_ref']), int(a['to_ref'])): a for a in got}; de = {(int(a['from_ref']), int(a['to_ref'])): a for a in exp}
only_g = [k for k in dg if k not in de]; only_e = [k for k in de if k not in dg]
print('only sharded', len(only_g), [tuple(hex(x) for x in k) for k in only_g[:8]]); print('only unsharded', len(only_e), [tuple(hex(x) for x in k) for k in only_e[:8]])
common = [k for k in dg if k in de]
cd = sum(1 for k in common if dg[k]['count'] != de[k]['count']); print('common', len(common), 'count differs', cd)
sd = [abs(float(dg[k]['score']) - float(de[k]['score'])) for k in common]; print('score max diff on common', max(sd))
ro = ref.outbound_ips(); print('ref obips', len(ro), ro[:24])
for r in range(world):
    so = engs[r].outbound_ips(); print('shard', r, 'obips', len(so), so[:24])
    extra = np.setdiff1d(so, ro); print('  extra vs ref', [replay.ip_str(int(x)) for x in extra[:10]], 'missing', [replay.ip_str(int(x)) for x in np.setdiff1d(ro, so)[:10]])
# emulate K1 ownership on host to find misrouted events
pod = {int(ip): i for i, ip in enumerate(topo.pod_ips)}; svc = {int(ip): topo.n_pods + j for j, ip in enumerate(topo.svc_ips)}
