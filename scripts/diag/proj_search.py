import numpy as np, itertools, sys
z=np.load(sys.argv[1])
vis=(z['ref_radii']>0)
P3=z['means3D'][vis].astype(np.float64)
ref=z['ref_means2D'][vis]; hip=z['hip_means2D'][vis]
pm=z['pm'].astype(np.float64).reshape(-1)
W=H=128
f32=lambda v: np.asarray(v).astype(np.float32).astype(np.float64)
mul=lambda a,b: f32(a*b); fma=lambda a,b,c: f32(a*b+c); add=lambda a,b: f32(a+b); div=lambda a,b: f32(a/b)
x,y,zz=P3[:,0],P3[:,1],P3[:,2]
def lin4(m,i0,pat):
    a,b,c,d=m[i0],m[i0+4],m[i0+8],m[i0+12]
    P=[lambda: add(fma(c,zz, fma(b,y, mul(a,x))), d), lambda: add(fma(c,zz, fma(a,x, mul(b,y))), d), lambda: fma(c,zz, fma(b,y, fma(a,x,d))),
       lambda: add(add(add(mul(a,x),mul(b,y)),mul(c,zz)),d), lambda: add(add(fma(a,x,mul(b,y)),mul(c,zz)),d), lambda: add(add(fma(b,y,mul(a,x)),mul(c,zz)),d),
       lambda: add(fma(c,zz,add(mul(a,x),mul(b,y))),d), lambda: fma(a,x, fma(b,y, fma(c,zz,d))), lambda: fma(a,x, add(fma(c,zz,d), mul(b,y))),
       lambda: add(fma(a,x,mul(b,y)), fma(c,zz,d)), lambda: add(fma(b,y,mul(a,x)), fma(c,zz,d)), lambda: add(add(mul(a,x),mul(b,y)), fma(c,zz,d)),
       lambda: fma(a,x, fma(b,y, add(mul(c,zz),d))), lambda: fma(b,y, fma(a,x, fma(c,zz,d))), lambda: fma(b,y, add(mul(a,x), fma(c,zz,d))),
       lambda: fma(c,zz, add(fma(a,x,mul(b,y)), d)) , lambda: fma(c,zz, add(fma(b,y,mul(a,x)), d)), lambda: add(fma(c,zz,mul(a,x)) , add(mul(b,y), d)) ]
    return P[pat]()
NP=18
def ndc2pix(v,S): return np.float32(((v.astype(np.float64)+1.0)*S-1.0)*0.5)
for comp,(row,S) in enumerate(((0,W),(1,H))):
    best=[]
    for pw in range(NP):
        hw=lin4(pm,3,pw); p_w=div(1.0, add(hw, np.float64(np.float32(0.0000001))))
        for px in range(NP):
            hx=lin4(pm,row,px)
            v=mul(hx,p_w)
            pix=ndc2pix(v.astype(np.float32),S)
            best.append(((pix.view(np.int32)==ref[:,comp].view(np.int32)).mean(), (pix.view(np.int32)==hip[:,comp].view(np.int32)).mean(), pw, px))
    best.sort(key=lambda t:-t[0]); print('comp',comp,'vs ref',best[:3]); best.sort(key=lambda t:-t[1]); print('        vs hip',best[:2])
