"""Numpy model of the ray sort of the tail march: (wave, tile) pairs and sub-chunks per tile of the benchmark scan for different sort keys (direction bins, cells row-major / Morton, tile columns).  64 rays x an eighth of a tail per wave."""
import numpy as np, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from warpsense_amd import synthetic as S
pts = S.os1_128_scan().astype(np.float64)
n = len(pts)
res=50.0; size=513
d = np.linalg.norm(pts,axis=1); u = pts/d[:,None]
hv = np.floor(pts/res).astype(int)
def morton(bx,by):
    def sp(v):
        v=(v|(v<<4))&0x0f0f; v=(v|(v<<2))&0x3333; v=(v|(v<<1))&0x5555; return v
    return (sp(bx)<<1)|sp(by)
def pairs(order, label):
    # tail from d-1.2m to d+1.0m, 8 parts; sample every 25mm
    tot=0
    ts = np.linspace(-1200,1000,89)
    nparts=8
    for w0 in range(0,n,64):
        idx = order[w0:w0+64]
        for p in range(nparts):
            tt = ts[p*11:(p+1)*11+1]
            P = pts[idx][:,None,:] + u[idx][:,None,:]*tt[None,:,None]
            v = np.floor(P/res).astype(int)+256
            tile = (v[...,0]>>2)*100000 + (v[...,1]>>2)*100 + (v[...,2]>>6)
            tot += len(np.unique(tile))
    print(label, tot)
# direction sort
az = np.arctan2(pts[:,1],pts[:,0]); b=np.clip(((az+np.pi)*(1024/(2*np.pi))).astype(int),0,1023)
se = pts[:,2]/d; e=np.clip(((se+0.5)*8).astype(int),0,7)
pairs(np.argsort(b*8+e,kind='stable'),'direction')
cw=9
bx=np.clip((hv[:,0]+256)//cw,0,63); by=np.clip((hv[:,1]+256)//cw,0,63); zh=(hv[:,2]>=0).astype(int)
pairs(np.argsort((bx*64+by)*2+zh,kind='stable'),'cells rowmajor')
pairs(np.argsort(zh*4096+morton(bx,by),kind='stable'),'cells morton')
# tile-column morton at 0.2 m: 128x128 -> 16384 bins *2
tx=(hv[:,0]+256)>>2; ty=(hv[:,1]+256)>>2
def sp8(v):
    v=(v|(v<<8))&0x00ff00ff; v=(v|(v<<4))&0x0f0f0f0f; v=(v|(v<<2))&0x33333333; v=(v|(v<<1))&0x55555555; return v
pairs(np.argsort(zh*(1<<20)+((sp8(tx)<<1)|sp8(ty)),kind='stable'),'tilecol morton (ideal fine)')
pairs(np.arange(n),'scan order')

def entries(order,label):
    from collections import defaultdict
    ent = defaultdict(int); recs=defaultdict(int)
    ts = np.linspace(-1200,1000,89)
    for w0 in range(0,n,64):
        idx = order[w0:w0+64]
        for p in range(8):
            tt = ts[p*11:(p+1)*11]
            P = pts[idx][:,None,:] + u[idx][:,None,:]*tt[None,:,None]
            v = np.floor(P/res).astype(int)+256
            tile = ((v[...,0]>>2)*100000 + (v[...,1]>>2)*100 + (v[...,2]>>6)).ravel()
            t,c = np.unique(tile, return_counts=True)
            for a,b in zip(t,c):
                ent[a] += (b+31)//32; recs[a]+=b
    e = np.array(list(ent.values())); r=np.array([recs[k] for k in ent])
    print(label,'tiles',len(e),'entries mean',e.mean(),'>64:',(e>64).sum(),'>128:',(e>128).sum(),'records in >64 tiles',r[e>64].sum(),'of',r.sum(), 'slots',e.sum()*32)
entries(np.argsort(zh*4096+morton(bx,by),kind='stable'),'morton')

# polar cells around the sensor, far rings first (longest work items first): 32 rings x 128 sectors x above / below
rr = np.hypot(pts[:,0], pts[:,1]); ang = np.arctan2(pts[:,1], pts[:,0])
ring = np.clip((rr / 450.0).astype(int), 0, 31); sec = np.clip(((ang + np.pi) * (128 / (2 * np.pi))).astype(int), 0, 127)
pairs(np.argsort(((31 - ring) * 2 + zh) * 128 + sec, kind='stable'), 'polar, far first (ring, z, sector)')
pairs(np.argsort(((31 - ring) * 128 + sec) * 2 + zh, kind='stable'), 'polar, far first (ring, sector, z)')
