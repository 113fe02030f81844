"""GPU diagnostic: per-tensor deviation of the engine's state from the oracle's after k train steps, and the one-step
gradient from the ORACLE's intermediate state (separates an arithmetic difference from chaos amplification)."""
import numpy as np, sys, torch
sys.path.insert(0,'/root/repo')
from oracle import np_oracle as O
from tests.helpers import CONFS, oracle_steps, frac_bad, engine_hyper, etas_for
import mfas_amd as M
dev=torch.device('cuda:0')
def order(tag,E,N):
    rng=np.random.default_rng(1800+sum(map(ord,tag))); return np.stack([rng.permutation(N) for _ in range(E)])
np.set_printoptions(linewidth=200, precision=3)
for tag,(cname,R,bn,drpt,B,N,snr) in {"bench": ("c4", 128, True, 0.5, 16, 64, 0.3), "nodrop": ("c4", 128, True, 0.0, 16, 64, 0.3)}.items():
    conf=np.array(CONFS[cname]); hp=O.Hyper(R=R,B=B,bn=bn,drpt=drpt,epochs=3)
    ttr=O.synth_table(N,21,snr=snr); ta=M.FeatureTable.from_numpy(ttr,dev,torch.float32)
    od=order("bench",3,N)
    # (1) sign flips after step 1
    pop=M.Population(engine_hyper(hp),[conf],dev,drop_seeds=[40]); pop.set_state_dict(0,O.init_params(conf,hp,5))
    pop.train(ta,None,3,etas_for(hp,N),order=torch.from_numpy(od.astype(np.int32)),max_steps=1)
    P1,st,_=oracle_steps(conf,hp,O.init_params(conf,hp,5),ttr,1,seed=40,order=od)
    w=pop.get_state_dict(0,0)
    for key in w:
        if key in st.m:
            d=np.abs(w[key].numpy()-P1[key]); print(tag,"flips after step 1",key,int((d>1e-3).sum()),"of",d.size, " >1e-5:",int((d>1e-5).sum()))
    pop.close()
    # (2) one step from the oracle's state P1, batch = rows of step 2 (order row shifted so that step 0 uses batch 1's rows), mask stream position 0
    od2=od.copy(); od2[0]=np.roll(od[0],-B)
    pop=M.Population(engine_hyper(hp),[conf],dev,drop_seeds=[40]); pop.set_state_dict(0,{k:v.copy() for k,v in P1.items()})
    pop.train(ta,None,3,etas_for(hp,N),order=torch.from_numpy(od2.astype(np.int32)),max_steps=1)
    P2,st2,_=oracle_steps(conf,hp,{k:v.copy() for k,v in P1.items()},ttr,1,seed=40,order=od2)
    m=pop.get_state_dict(0,1)
    for key in st2.m:
        mm=m[key].numpy(); mo=st2.m[key]; rel=np.abs(mm-mo)/(np.abs(mo)+1e-30)
        print(tag,"one step from P1",key,"m relerr p50 %.1e p99 %.1e"%(np.median(rel),np.quantile(rel,0.99)))
    pop.close()
