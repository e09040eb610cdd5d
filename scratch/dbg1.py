import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from magma_amd.engine import Dataset, Engine
from magma_amd.synth import simu_arrays
from oracle import magma_oracle as orc
M,n,N=300,50,2000
a=simu_arrays(M,n,N,seed=1003)
ds=Dataset.from_arrays(a["off"],a["t"],a["y"],grid=a["grid"])
eng=Engine(ds)
th0=np.array([1.,1.]); thi=np.tile([1.,1.,0.2],(M,1))
t=time.time(); info=eng.e_step(th0,thi,0.2,0.0); torch.cuda.synchronize(); print("estep info",info,"flagged",eng.stats, time.time()-t)
mean=eng.be.get_mean(); cov=eng.be.get_cov()
print("mean finite",np.isfinite(mean).all(),"cov finite",np.isfinite(cov).all())
ids=np.repeat(np.arange(M),n); db={"ID":ids,"Timestamp":a["t"],"Output":a["y"]}
hp={"theta_0":th0,"theta_i":{i:thi[i] for i in range(M)}}
par=orc._hyperposterior(db,a["grid"],0.0,orc.kernel,orc.kernel,hp,0.2)
print("cov err",np.abs(cov-par["cov"][1]).max()/np.abs(par["cov"][1]).max(),"mean err",np.abs(mean-par["mean"]["Output"]).max())
f,g=eng.be.indiv_eval(thi,True)
print("f finite",np.isfinite(f).all(),"g finite",np.isfinite(g).all(), f[:3], g[:2])
for i in range(3):
    sel=ids==i; db_i={"Timestamp":a["t"][sel],"Output":a["y"][sel]}
    m_i=orc._mean_at(par["mean"],db_i["Timestamp"]); c_i=orc._block(par["cov"][0],par["cov"][1],db_i["Timestamp"])
    print(orc.logL_GP_mod(thi[i],db_i,m_i,orc.kernel,c_i), orc.gr_GP_mod(thi[i],db_i,m_i,orc.kernel,c_i))
f0,g0,inf0=eng.be.theta0_eval(th0,0.4,True); print("theta0",f0,g0,inf0)
print(orc.logL_GP_mod(th0,par["mean"],0.0,orc.kernel,par["cov"][1],0.4), orc.gr_GP_mod(th0,par["mean"],0.0,orc.kernel,par["cov"][1],0.4))
# m-step
t=time.time(); n0,ni=eng.m_step(th0,thi,False); print("mstep",time.time()-t,eng.stats,n0, np.isnan(ni).sum())
