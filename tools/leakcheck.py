import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, ripp_amd as R, orclib as o, helpers as h
R.init(0)
n=1<<12
a,b,r=R.synth_g1(1,n),R.synth_g2(2,n),R.synth_fr(3,n)
value=R.product_of_pairings_with_coeffs(a,b,r)
alpha,beta=o.fr_array([5])[0],o.fr_array([7])[0]
def used(): f,t=torch.cuda.mem_get_info(); return (t-f)/2**20
base=None
for it in range(40):
    p=R.SIPP.prove(a,b,r,value); assert R.SIPP.verify(a,b,r,value,p)
    srs=R.SRS.from_trapdoors(alpha,beta,256); pf,_=R.aggregate_proofs(srs,a[:256],b[:256],a[256:512]); srs.close()
    R.MultiexponentiationInnerProductG2.inner_product(o.to_jac_g2(b),r)
    if it==4: base=used()
    if it in (4,20,39): print(it, "device MB used: %.1f"%used())
print("growth MB: %.1f"%(used()-base))
