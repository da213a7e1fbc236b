#!/usr/bin/env python3
"""Accuracy of csrc/sincos.hpp (the encodings' sin / cos: fp64 range reduction + two fp32 minimax polynomials) — a numpy restatement
with emulated FMAs, against a float64 evaluation and against the reference's own arithmetic (CPU torch.sin / torch.cos), over the
argument ranges of the positional encodings: 2^l x, l = 0..9, |x| <= 5 and 2^l d, l = 0..3, |d| <= 1.  CPU only.
The coefficients come from a weighted least-squares (Remez-like) fit of (sin x - x) / x^3 and (cos x - 1) / x^2 in x^2 on
[0, (pi/4)^2 (1 + 1e-4)]."""
import numpy as np, torch
f32=np.float32
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
S=[f32(-0.1666666716337204), f32(0.008333305828273296), f32(-0.00019828711810987443), f32(2.6021755274996394e-06)]
C=[f32(-0.5), f32(0.04166664928197861), f32(-0.0013887588866055012), f32(2.4463804948027246e-05)]
def cn_sincos(a):
    ad=a.astype(np.float64)
    kd=np.rint(ad*0.63661977236758134308)
    r=(kd*(-1.57079632679489661923)+ad)   # fp64 fma: emulate with higher precision
    # exact fma emulation via longdouble
    r=(np.longdouble(kd)*np.longdouble(-1.57079632679489661923)+np.longdouble(ad)).astype(np.float64)
    r=r.astype(f32)
    q=kd.astype(np.int64)
    z=(r*r).astype(f32)
    ps=fma(z, np.full_like(z,S[3]), np.full_like(z,S[2])); ps=fma(z,ps,np.full_like(z,S[1])); ps=fma(z,ps,np.full_like(z,S[0]))
    rz=(r*z).astype(f32)
    s=fma(rz,ps,r)
    pc=fma(z, np.full_like(z,C[3]), np.full_like(z,C[2])); pc=fma(z,pc,np.full_like(z,C[1])); pc=fma(z,pc,np.full_like(z,C[0]))
    c=fma(z,pc,np.ones_like(z))
    swap=(q&1)==1
    sn=np.where(swap,c,s); cs=np.where(swap,s,c)
    sn=np.where((q&2)==2,-sn,sn); cs=np.where(((q+1)&2)==2,-cs,cs)
    return sn.astype(f32),cs.astype(f32)
rs=np.random.RandomState(0)
worst={}
for L,lo in ((10,5.0),(4,1.0)):
    for l in range(L):
        x=(rs.uniform(-lo,lo,2_000_000).astype(f32)*f32(2.0**l)).astype(f32)
        sn,cs=cn_sincos(x)
        ts,tc=np.sin(x.astype(np.float64)),np.cos(x.astype(np.float64))
        e=max(np.abs(sn-ts).max(),np.abs(cs-tc).max())
        tt=torch.from_numpy(x)
        et=max(np.abs(torch.sin(tt).numpy()-ts).max(),np.abs(torch.cos(tt).numpy()-tc).max())
        dd=max(np.abs(torch.sin(tt).numpy()-sn).max(),np.abs(torch.cos(tt).numpy()-cs).max())
        frac_same=((torch.sin(tt).numpy()==sn).mean()+(torch.cos(tt).numpy()==cs).mean())/2
        print(f"range {lo} l={l}: custom max err {e:.3e}  torch max err {et:.3e}  custom vs torch max {dd:.3e}  bit-equal {frac_same:.4f}")
# edge: large args
x=np.array([1e5,-3e5,1048575.0,0.0,-0.0,1e-20,np.pi/4,-np.pi/4, 2391.04],f32)
sn,cs=cn_sincos(x); print(np.abs(sn-np.sin(x.astype(np.float64))).max(), np.abs(cs-np.cos(x.astype(np.float64))).max())
