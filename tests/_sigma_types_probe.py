import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1,eri=S.synthetic_integrals(30)
for name,gen,n in (('hf',S.hf_centred_strings,317),('hf',S.hf_centred_strings,1000),('uniform',S.uniform_strings,4000)):
    sa,sb=gen(30,8,n,1001),gen(30,8,n,1001+7919)
    ctx=_capi.Context(h1,eri)
    ctx.set_subspace(sa,sb); ctx.davidson(fetch=False, max_cycle=3)
    out={}
    for mask in (7,1,2,4):
        os.environ['SQD_SIGMA_TYPES']=str(mask)
        out[mask]=round(ctx.time_sigma(20)*1e3,1)
    os.environ.pop('SQD_SIGMA_TYPES')
    print(name,n,'links',ctx.link_counts(0),ctx.link_counts(1),'sigma us by type mask',out, flush=True)
    ctx.close()
