import sys, numpy as np, torch
sys.path.insert(0,"tests"); sys.path.insert(0,".")
import oracle_lib as O
import warpsense_amd as W
from warpsense_amd import synthetic as S
tau,res,size,rings,az=1000,50,(128,128,64),32,256
mw=640
lm = W.LocalMap(*size, tau, 0)
on = O.OracleMap(size, tau, 0)
t = W.TSDFCuda(lm.device_map(), tau, mw, res)
he = (size[0]*res*0.4, size[1]*res*0.35, size[2]*res*0.3)
pts = S.os1_128_scan(rings=rings, azimuths=az, half_extents_mm=he, seed=7)
O.update_min(on, pts, (0,0,0), (0,0,32768), tau, res)
t.scatter(torch.from_numpy(pts).cuda(), (0,0,0), (0,0,32768))
host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
t.new_map().to_host(host)
new = host.data_
d = np.nonzero(new != on.data)[0]
print("differ", d.size, "of touched", int((on.data != O.pack(tau,0)).sum()))
for i in d[:10]:
    print(i, O.unpack(new[i:i+1]), O.unpack(on.data[i:i+1]))
print(t.stats())
sz = np.array([129,129,65])
x = d // (129*65); y = (d // 65) % 129; z = d % 65
# storage -> world: offset = size/2, pos=0
wx = x - 64; wy = y - 64; wz = z - 32
r = np.sqrt((wx*50.0)**2 + (wy*50.0)**2 + (wz*50.0)**2)
print("missing voxel range histogram (m):", np.histogram(r/1000.0, bins=[0,0.25,0.5,0.75,1,1.5,2,2.5,3,4,5])[0])
tch = np.nonzero(on.data != O.pack(tau,0))[0]
x = tch // (129*65); y = (tch // 65) % 129; z = tch % 65
r2 = np.sqrt(((x-64)*50.0)**2 + ((y-64)*50.0)**2 + ((z-32)*50.0)**2)
print("all touched histogram        :", np.histogram(r2/1000.0, bins=[0,0.25,0.5,0.75,1,1.5,2,2.5,3,4,5])[0])
