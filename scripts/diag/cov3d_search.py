import numpy as np, itertools
z=np.load('/root/repo/gpurun_out/ref_geom.npz')
vis=(z['ref_radii']>0)
sc=z['scales'][vis].astype(np.float64); q=z['rot'][vis].astype(np.float64)
ref=z['ref_cov3D'][vis]; hip=z['hip_cov3D'][vis]
f32=lambda v: v.astype(np.float32).astype(np.float64)
mul=lambda a,b: f32(a*b)
fma=lambda a,b,c: f32(a*b+c)
add=lambda a,b: f32(a+b)
r,x,y,zq=q[:,0],q[:,1],q[:,2],q[:,3]
def inner(a,b,c,d,sign,pat):
    # a*b + sign*c*d
    if pat=='A': return fma(a,b, sign*mul(c,d))
    if pat=='B': return fma(sign*c,d, mul(a,b))
    return add(mul(a,b), sign*mul(c,d))
def diag(a,b,pat):  # 1 - 2(a*a + b*b)
    t=inner(a,a,b,b,1.0,pat)
    return f32(1.0-2.0*t)
def off(a,b,c,d,sign,pat): return f32(2.0*inner(a,b,c,d,sign,pat))
ents={ (0,0):lambda p: diag(y,zq,p), (0,1):lambda p: off(x,y,r,zq,-1.0,p), (0,2):lambda p: off(x,zq,r,y,1.0,p),
       (1,0):lambda p: off(x,y,r,zq,1.0,p), (1,1):lambda p: diag(x,zq,p), (1,2):lambda p: off(y,zq,r,x,-1.0,p),
       (2,0):lambda p: off(x,zq,r,y,-1.0,p), (2,1):lambda p: off(y,zq,r,x,1.0,p), (2,2):lambda p: diag(x,y,p)}
Rv={(e,p):ents[e](p) for e in ents for p in 'ABC'}
def sum3(p0a,p0b,p1a,p1b,p2a,p2b,pat):
    if pat=='S1': return fma(p2a,p2b, fma(p1a,p1b, mul(p0a,p0b)))
    if pat=='S2': return fma(p2a,p2b, fma(p0a,p0b, mul(p1a,p1b)))
    if pat=='S3': return add(add(mul(p0a,p0b),mul(p1a,p1b)), mul(p2a,p2b))
    if pat=='S4': return fma(p0a,p0b, fma(p1a,p1b, mul(p2a,p2b)))
    if pat=='S5': return add(fma(p1a,p1b, mul(p0a,p0b)), mul(p2a,p2b))
    if pat=='S6': return fma(p2a,p2b, add(mul(p0a,p0b), mul(p1a,p1b)))
    if pat=='S7': return fma(p0a,p0b, add(mul(p1a,p1b), mul(p2a,p2b)))
    if pat=='S8': return fma(p1a,p1b, fma(p2a,p2b, mul(p0a,p0b)))
outs=[(0,0),(0,1),(0,2),(1,1),(1,2),(2,2)]
for oi,(rr,cc) in enumerate(outs):
    best=[]
    rows=sorted({rr,cc})
    ent_list=[(i,k) for i in rows for k in range(3)]
    for pats in itertools.product('ABC', repeat=len(ent_list)):
        R={e:Rv[(e,p)] for e,p in zip(ent_list,pats)}
        M={e:mul(sc[:,e[1]],R[e]) for e in ent_list}
        for sp in ('S1','S2','S3','S4','S5','S6','S7','S8'):
            v=sum3(M[(rr,0)],M[(cc,0)],M[(rr,1)],M[(cc,1)],M[(rr,2)],M[(cc,2)],sp).astype(np.float32)
            mr=(v.view(np.int32)==ref[:,oi].view(np.int32)).mean(); mh=(v.view(np.int32)==hip[:,oi].view(np.int32)).mean()
            best.append((mr,mh,pats,sp))
    best.sort(key=lambda t:-t[0])
    print((rr,cc),'best vs ref',[(round(a,4),round(b,4),''.join(p),s) for a,b,p,s in best[:3]])
    best.sort(key=lambda t:-t[1])
    print('      best vs hip',[(round(a,4),round(b,4),''.join(p),s) for a,b,p,s in best[:2]])
