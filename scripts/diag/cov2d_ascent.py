import numpy as np, sys, random
f32=lambda v: np.asarray(v).astype(np.float32).astype(np.float64)
mul=lambda a,b: f32(a*b); fma=lambda a,b,c: f32(a*b+c); add=lambda a,b: f32(a+b); div=lambda a,b: f32(a/b)
class DS:
    def __init__(s, path, W=128, H=128):
        z=np.load(path); vis=(z['ref_radii']>0)
        s.P3=z['means3D'][vis].astype(np.float64); s.V6=z['ref_cov3D'][vis].astype(np.float64)
        s.ref=z['ref_conic_opacity'][vis][:, :3]; s.vm=z['vm'].astype(np.float64).reshape(-1)
        tanx,tany=z['tanfov']
        s.fx=np.float64(np.float32(W/(np.float32(2.0)*np.float32(tanx)))); s.fy=np.float64(np.float32(H/(np.float32(2.0)*np.float32(tany))))
        s.limx=np.float64(np.float32(np.float32(1.3)*np.float32(tanx))); s.limy=np.float64(np.float32(np.float32(1.3)*np.float32(tany)))
        s.dref=z['ref_depths'][vis]
def lin4(m,i0,pat,x,y,zz):
    a,b,c,d=m[i0],m[i0+4],m[i0+8],m[i0+12]
    P=[lambda: add(fma(c,zz, fma(b,y, mul(a,x))), d), lambda: add(fma(c,zz, fma(a,x, mul(b,y))), d), lambda: fma(c,zz, fma(b,y, fma(a,x,d))),
       lambda: add(add(add(mul(a,x),mul(b,y)),mul(c,zz)),d), lambda: add(add(fma(a,x,mul(b,y)),mul(c,zz)),d), lambda: add(add(fma(b,y,mul(a,x)),mul(c,zz)),d),
       lambda: add(fma(c,zz,add(mul(a,x),mul(b,y))),d), lambda: add(fma(a,x,mul(b,y)), fma(c,zz,d)), lambda: add(fma(b,y,mul(a,x)), fma(c,zz,d)),
       lambda: fma(a,x, fma(b,y, fma(c,zz,d))), lambda: fma(b,y, fma(a,x, fma(c,zz,d))), lambda: fma(c,zz, add(fma(a,x,mul(b,y)), d)), lambda: fma(c,zz, add(fma(b,y,mul(a,x)), d))]
    return P[pat]()
NL=13
def s3(p,q,pat):
    P=[lambda: fma(p[2],q[2], fma(p[1],q[1], mul(p[0],q[0]))), lambda: fma(p[2],q[2], fma(p[0],q[0], mul(p[1],q[1]))),
       lambda: add(add(mul(p[0],q[0]),mul(p[1],q[1])),mul(p[2],q[2])), lambda: add(fma(p[1],q[1], mul(p[0],q[0])), mul(p[2],q[2])),
       lambda: add(fma(p[0],q[0], mul(p[1],q[1])), mul(p[2],q[2])), lambda: fma(p[2],q[2], add(mul(p[0],q[0]), mul(p[1],q[1]))),
       lambda: fma(p[0],q[0], fma(p[1],q[1], mul(p[2],q[2]))), lambda: fma(p[1],q[1], fma(p[0],q[0], mul(p[2],q[2]))),
       lambda: fma(p[0],q[0], fma(p[2],q[2], mul(p[1],q[1]))), lambda: fma(p[1],q[1], fma(p[2],q[2], mul(p[0],q[0])))]
    return P[pat]()
NS=10
def t2(p0,q0,p2,q2,pat):  # two-term (zero term dropped): p0*q0 + p2*q2
    return [lambda: fma(p2,q2, mul(p0,q0)), lambda: fma(p0,q0, mul(p2,q2)), lambda: add(mul(p0,q0), mul(p2,q2))][pat]()
def detf(c00,c01,c11,pat):
    return [lambda: fma(c00,c11, -mul(c01,c01)), lambda: fma(-c01,c01, mul(c00,c11)), lambda: add(mul(c00,c11), -mul(c01,c01))][pat]()
VARS=[('tx',NL),('ty',NL),('tz',NL)]+[(f'a{r}',3) for r in range(3)]+[(f'b{r}',3) for r in range(3)]+[(f'Va{i}',NS) for i in range(3)]+[(f'Vb{i}',NS) for i in range(3)]+[('c00',NS),('c01',NS),('c11',NS),('det',3),('j',2)]
def evaluate(ds, c):
    x,y,zz=ds.P3[:,0],ds.P3[:,1],ds.P3[:,2]; vm=ds.vm
    tx=lin4(vm,0,c['tx'],x,y,zz); ty=lin4(vm,1,c['ty'],x,y,zz); tz=lin4(vm,2,c['tz'],x,y,zz)
    txtz=div(tx,tz); tytz=div(ty,tz)
    tx2=mul(np.minimum(ds.limx,np.maximum(-ds.limx,txtz)),tz); ty2=mul(np.minimum(ds.limy,np.maximum(-ds.limy,tytz)),tz)
    if c['j']==0:
        j00=div(ds.fx,tz); j02=div(-mul(ds.fx,tx2), mul(tz,tz)); j11=div(ds.fy,tz); j12=div(-mul(ds.fy,ty2), mul(tz,tz))
    else:
        j00=div(ds.fx,tz); j02=-div(mul(ds.fx,tx2), mul(tz,tz)); j11=div(ds.fy,tz); j12=-div(mul(ds.fy,ty2), mul(tz,tz))
    a=[t2(vm[4*r],j00,vm[4*r+2],j02,c[f'a{r}']) for r in range(3)]
    b=[t2(vm[4*r+1],j11,vm[4*r+2],j12,c[f'b{r}']) for r in range(3)]
    V6=ds.V6; V=[[V6[:,0],V6[:,1],V6[:,2]],[V6[:,1],V6[:,3],V6[:,4]],[V6[:,2],V6[:,4],V6[:,5]]]
    Va=[s3(a,[V[0][i],V[1][i],V[2][i]],c[f'Va{i}']) for i in range(3)]
    Vb=[s3(b,[V[0][i],V[1][i],V[2][i]],c[f'Vb{i}']) for i in range(3)]
    c00=add(s3(Va,a,c['c00']),np.float64(np.float32(0.3))); c01=s3(Vb,a,c['c01']); c11=add(s3(Vb,b,c['c11']),np.float64(np.float32(0.3)))
    det=detf(c00,c01,c11,c['det']); di=div(1.0,det)
    v=np.stack([mul(c11,di), mul(-c01,di), mul(c00,di)],1).astype(np.float32)
    return (v.view(np.int32)==ds.ref.view(np.int32)).all(1).mean()
dss=[DS(p) for p in sys.argv[1:]]
def score(c): return sum(evaluate(d,c) for d in dss)/len(dss)
random.seed(0)
best_overall=None
for restart in range(3):
    c={k:(random.randrange(n) if restart else 0) for k,n in VARS}
    if restart==0: c.update(tz=4,det=1,Va0=1,Va1=1,Va2=1)
    cur=score(c)
    for sweep in range(6):
        improved=False
        for k,n in VARS:
            old=c[k]; bestv,bests=old,cur
            for v in range(n):
                if v==old: continue
                c[k]=v; sc=score(c)
                if sc>bests+1e-9: bestv,bests=v,sc
            c[k]=bestv
            if bests>cur+1e-9: cur=bests; improved=True
        print('restart',restart,'sweep',sweep,'score',round(cur,5),c,flush=True)
        if not improved or cur>0.99999: break
    if best_overall is None or cur>best_overall[0]: best_overall=(cur,dict(c))
print('BEST',best_overall)
