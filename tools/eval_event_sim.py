"""Event process of the fused evaluator's candidate path, simulated: 64 users (one wavefront) x 100 000 iid scores in steps of G
items; a list is compacted to its K best when it holds more than `trigger` entries.  Prints how often a wavefront takes the
candidate path (slow_frac), compactions and events per user, and the fraction of steps with a compaction."""
import numpy as np
rng=np.random.default_rng(1)
def sim(K=10, trigger=60, G=16, N=100000, W=8):
    slow=0; comps=0; events=0; steps=N//G; comp_steps=0
    for w in range(W):
        sc=rng.standard_normal((64,N)).astype(np.float32)
        thr=np.full(64,-np.inf,np.float32); lists=[[] for _ in range(64)]
        for s in range(steps):
            blk=sc[:,s*G:(s+1)*G]
            p=blk>thr[:,None]
            if p.any():
                slow+=1
                us=np.nonzero(p.any(1))[0]
                c=0
                for u in us:
                    lists[u].extend(blk[u][p[u]].tolist()); events+=int(p[u].sum())
                    if len(lists[u])>trigger:
                        l=sorted(lists[u],reverse=True)[:K]; lists[u]=l; thr[u]=l[-1]; comps+=1; c=1
                comp_steps+=c
    return dict(K=K,trigger=trigger,slow_frac=slow/(W*steps), comps_per_user=comps/(64*W), events_per_user=events/(64*W), comp_steps_frac=comp_steps/(W*steps))
for trig in (26,42,60,90):
    print(sim(trigger=trig,W=3))
