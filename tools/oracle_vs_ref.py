"""Developer script (dev container only: needs oracle/_ref/ref_det, i.e. the reference compiled from /root/reference):
the plain-C oracle against the reference itself on fresh seeds, all five seeding policies, bit for bit.
usage: python3 tools/oracle_vs_ref.py SECONDS"""
import sys, subprocess, numpy as np, time, os, json
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import oracle_lib, ref_io
from offline_raytracer_amd import api
R='/root/repo/oracle/_ref/ref_det'; base='/root/repo/data/'
t0=time.time(); bad=0; n=0
budget=float(sys.argv[1])
scenes=["c2_analytic","c4_dwarf_room","letters","glass_room","c3_bunny_room","testscene"]
k=0
while time.time()-t0<budget:
    name=scenes[k%len(scenes)]; seed=90000+k
    W,H,spp=(128,96,6) if name!="testscene" else (64,48,4)
    policy=["pixel","chunk","whole","tile32","sample"][k%5]; chunk=2 if policy=="chunk" else 1
    scn=base+name+'.scn'
    out=os.path.join(__import__('tempfile').gettempdir(), 'ovr.f32')
    js=subprocess.check_output([R,'render',scn,base,str(W),str(H),str(spp),str(seed),policy,out,str(chunk)]).decode().strip().splitlines()[-1]
    js=json.loads(js)
    ref=np.fromfile(out,'<f4').reshape(H,W,3)
    sc=api.Scene.load_scn(scn).commit()
    img,st=oracle_lib.OracleScene(sc.flatten(W,H)).render(W,H,spp,seed,policy,chunk=chunk,threads=8)
    d=int((img.view('<u4')!=ref.view('<u4')).any(2).sum())
    same_count = st['shapes_tested']==js['shapes_tested']
    if d or not same_count: bad+=1; print('DIFF',name,policy,seed,d,same_count,flush=True)
    n+=1; k+=1
print('oracle vs reference (ref_det): %d renders, %d with differences'%(n,bad))
