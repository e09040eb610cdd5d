import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from magma_amd.engine import Dataset, Engine
from magma_amd.synth import simu_arrays
from magma_amd import _lib
M,n,N=10000,50,2000
a=simu_arrays(M,n,N,seed=1003)
ds=Dataset.from_arrays(a["off"],a["t"],a["y"],grid=a["grid"])
eng=Engine(ds); be=eng.be
th0=np.array([1.,1.]); thi=np.tile([1.,1.,0.2],(M,1))
eng.e_step(th0,thi,0.2,0.0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
st=be.stream()
def ev(grad):
    be._set_theta(thi)
    _lib.call("magma_indiv_eval",0,be.M,be.nmax,C.c_void_p(be.off.data_ptr()),C.c_void_p(be.t.data_ptr()),C.c_void_p(be.y.data_ptr()),C.c_void_p(be.idx.data_ptr()),C.c_void_p(be.theta.data_ptr()),3,C.c_void_p(be.mean.data_ptr()),C.c_void_p(be.cpack.data_ptr()),grad,None,0,C.c_void_p(be.f.data_ptr()),C.c_void_p(be.g.data_ptr()),C.c_void_p(be.info.data_ptr()),st)
print("indiv_eval f+g  ms", timeit(lambda: ev(1)))
print("indiv_eval f    ms", timeit(lambda: ev(0)))
P2=torch.zeros_like(be.P); w2=torch.zeros_like(be.w)
def sc():
    _lib.call("magma_estep_scatter",be.M,be.nmax,C.c_void_p(be.off.data_ptr()),C.c_void_p(be.t.data_ptr()),C.c_void_p(be.y.data_ptr()),C.c_void_p(be.idx.data_ptr()),C.c_void_p(be.theta.data_ptr()),3,be.N,C.c_void_p(P2.data_ptr()),C.c_void_p(w2.data_ptr()),C.c_void_p(be.info.data_ptr()),st)
print("estep_scatter   ms", timeit(sc))
A=be.P.clone(); ld=torch.zeros(1,dtype=torch.float64,device='cuda'); info=torch.zeros(1,dtype=torch.int32,device='cuda')
def inv():
    A.copy_(be.P)
    _lib.call("magma_spd_inverse",be.N,C.c_void_p(A.data_ptr()),C.c_void_p(ld.data_ptr()),C.c_void_p(info.data_ptr()),C.c_void_p(be.work.data_ptr()),st)
print("spd_inverse(+copy) ms", timeit(inv))
print("theta0_eval f+g ms", timeit(lambda: be.theta0_eval(th0,0.4,True)))
print("theta0_eval f   ms", timeit(lambda: be.theta0_eval(th0,0.4,False)))
f,g=be.indiv_eval(thi,True); print("check", f[:2], g[0])
