#!/usr/bin/env python3
"""CPU-only: how far fp32 arithmetic itself is from the exact substep -- the f32 oracle against the f64 oracle, teacher-forced, same active
sets only (the yardstick tests/test_gpu_substep.py holds the HIP kernel to) -- and what computing the POSITION gaps of the constraint
rows in double precision would buy (oracle experiment switch orc_set_precise_gaps: bit 0 closure gaps, bit 1 flat-ground contact depth,
bit 2 planar rows; the rest of the substep stays fp32).
  python tools/fp32_yardstick.py [CassieEnv-v0] [bits]
Round 6, units of 1e-5 (1 + |x|), one substep:
  Walker3DCustomEnv-v0  bits 0   median 1.17  p99 16.9
  CassieEnv-v0          bits 0   median 9.29  p99 119     (dt = 0.6 ms: a 1e-7 m rounding in a gap becomes 1e-4 m/s of bias)
  CassieEnv-v0          bits 1   median 7.73  p99 118     closure gaps exact
  CassieEnv-v0          bits 2   median 7.64  p99 54      toe depths exact
  CassieEnv-v0          bits 3/7 median 5.50  p99 52.7    both (planar rows add nothing)
Forming the closure gap in the common ancestor's frame (the thigh) instead of the base frame -- the cheap fp32-only variant -- shrinks the gap's
own rounding error from 4.3e-8 m to 2.8e-8 m (median over the mocap cycle's poses; numpy emulation of the kernel's fp32 walk), i.e. about a
third of what bit 0 buys.  Even EXACT gaps leave 5.5 units: the rest is the fp32 solve of closure rows with zero CFM at dt = 0.6 ms, not
the gaps.  A second, double-precision kinematics walk of the leg chains in the kernel (+8-10 % of a Cassie substep) for 9.3 -> 5.5 was
not built: dead end, recorded in profiles/HISTORY.md."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from mocca_envs_amd import model as M
from mocca_envs_amd.vec_env import compile_model_for
from oracle.oracle import Oracle
env_id = sys.argv[1] if len(sys.argv)>1 else "CassieEnv-v0"
precise = int(sys.argv[2]) if len(sys.argv)>2 else 0
task = M.TASK_CASSIE if "Cassie" in env_id else M.TASK_WALKER3D_CUSTOM
m = compile_model_for(env_id); m.n_substeps=1
if "Cassie" in env_id: m.n_llc=1
blob=m.to_bytes(); n,steps=256,160
orc=Oracle(blob,task,n,"f32"); o64=Oracle(blob,task,n,"f64")
if precise:
    import ctypes
    orc.lib.orc_set_precise_gaps.argtypes=[ctypes.c_void_p, ctypes.c_int]; orc.lib.orc_set_precise_gaps.restype=None
    orc.lib.orc_set_precise_gaps(orc.h, precise)
orc.reset(seed=4); o64.reset(seed=4)
rng=np.random.default_rng(2); nd=13+2*m.n_joints
units=lambda a,b: np.abs(a-b)/(1e-5*(1.0+np.abs(b)))
e=[]; ev=[]; ep=[]
for t in range(steps):
    o64.set_state(orc.get_state()); o64.set_task(orc.get_task())
    scale=1.0 if t%3 else 0.3
    a=(scale*rng.uniform(-1,1,(n,orc.act_dim))).astype(np.float32)
    _,_,dc,_=orc.step(a); o64.step(a)
    sc,s6=orc.get_state(),o64.get_state(); d6,dcb=o64.get_debug(),orc.get_debug()
    ok=np.isfinite(sc).all(1)&np.isfinite(s6).all(1)
    same=(d6[:,:12]==dcb[:,:12]).all(1)&ok
    if same.any():
        u=units(sc[same][:,:nd],s6[same][:,:nd]); e.append(u.max(1))
        nj=m.n_joints
        ev.append(u[:,7:13].max(1)); ep.append(u[:,13+nj:].max(1))
    if t%8==7:
        f=(dc!=0).astype(np.uint8)
        if f.any(): orc.reset(seed=4,mask=f)
e=np.concatenate(e); q=lambda x,p: float(np.percentile(x,p))
print(f"{env_id} precise={precise}: f32 vs f64 oracle, units of 1e-5(1+|x|): median {q(e,50):.3g} p90 {q(e,90):.3g} p99 {q(e,99):.3g} max {e.max():.3g}  (base vel median {q(np.concatenate(ev),50):.3g}, qd median {q(np.concatenate(ep),50):.3g})")
