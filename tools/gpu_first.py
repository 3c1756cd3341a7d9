"""First-light check on the GPU box: parity vs oracle on golden + synthetic, rough timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import verifybamid_amd as vb
from oracle.bridge import oracle_data
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
pts = [(0.5,(0,0),(0,0)),(0.03,(.01,.01),(.01,.01)),(0,(.01,.01),(.01,.01)),(0.2,(0.1,-0.12),(0.03,0.02)),(0.999,(1.01,0.01),(-1,0.5))]
for name in ('expected/result.Pileup','test.LongRead.pileup'):
    d = vb.PileupData.from_files(G+'/hapmap/hapmap_3.3.b37.dat', os.path.join(G,name), 2)
    od = oracle_data(d)
    with vb.LikelihoodContext(d) as ctx:
        print(ctx.info())
        got = ctx.llk([p[1] for p in pts],[p[2] for p in pts],[p[0] for p in pts])
        want = np.array([od.llk(p[1],p[2],p[0]) for p in pts])
        print(name, 'rel err', np.abs(got-want)/np.abs(want))
        for kw in [{}, dict(within_ancestry=True), dict(fix_alpha=0.1), dict(fix_pc=[0.034756,0.0193])]:
            r = ctx.optimize(**kw); o = od.optimize(**kw)
            print(kw, r['alpha'], o['alpha'], r['llk1'], o['llk1'], r['num_eval'], o['num_eval'])
for (M,k,seed) in [(10000,2,1),(100000,4,2)]:
    d = vb.synth.make_pileup(M, 30, k, 0.05, seed)
    od = oracle_data(d)
    rng = np.random.default_rng(5)
    B = 16
    pc1 = rng.normal(0,0.03,size=(B,k)); pc2 = rng.normal(0,0.03,size=(B,k)); al = rng.uniform(0,0.5,size=B)
    t0=time.time(); ctx = vb.LikelihoodContext(d); t1=time.time()
    print('ctx create %.3fs'%(t1-t0), ctx.info())
    got = ctx.llk(pc1,pc2,al)
    t0=time.time(); want = np.array([od.llk(pc1[i],pc2[i],al[i],num_thread=8) for i in range(B)]); t1=time.time()
    print('oracle %.1f ms/eval'%((t1-t0)/B*1e3))
    print('M',M,'max rel err', np.max(np.abs(got-want)/np.abs(want)))
    # determinism
    g2 = ctx.llk(pc1,pc2,al); print('deterministic', np.array_equal(got,g2))
    for B2 in (1,2,4,8,16):
        ctx.llk(pc1[:B2],pc2[:B2],al[:B2])
        t0=time.time()
        for _ in range(50): ctx.llk(pc1[:B2],pc2[:B2],al[:B2])
        dt=(time.time()-t0)/50
        print('B=%d host-sync eval: %.1f us/call, %.2f us/eval'%(B2, dt*1e6, dt*1e6/B2))
    t0=time.time(); r = ctx.optimize(); t1=time.time()
    print('optimize %.1f ms'%((t1-t0)*1e3), r['alpha'], r['num_eval'], r['num_launch_point'])
    t0=time.time(); o = od.optimize(num_thread=8); t1=time.time()
    print('oracle optimize %.1f ms'%((t1-t0)*1e3), o['alpha'], o['num_eval'], 'dalpha', abs(o['alpha']-r['alpha']))
    ctx.close()
