import sys, numpy as np, torch
sys.path.insert(0, '.')
import lmono_amd
from oracle import oracle as O
w = O.S1World()
traj = w.trajectory(64)
xyzi, off = w.scans(traj)
ctx = lmono_amd.Context(0)
d = torch.from_numpy(xyzi).cuda()
b = lmono_amd.ScanBatch(ctx, 64, len(xyzi))
b.scanreg(d.data_ptr(), off)
cnt = b.counts()
bad = np.nonzero(cnt[:,5])[0]
print("flagged scans", bad, cnt[bad][:, 5] if len(bad) else "")
for s in bad[:3]:
    for which,name in ((2,'less_sharp'),(4,'less_flat')):
        c = b.cloud(int(s), which, 200000)
        v = c[:,3].astype(np.int32)
        j = np.nonzero(np.diff(v) < 0)[0]
        print(s, name, len(c), "non-monotone at", j[:10], [ (c[k,3], c[k+1,3]) for k in j[:5]])
    ref = O.scanreg(xyzi[off[s]:off[s+1]])
    v = ref['less_flat'][:,3].astype(np.int32); print(" oracle lf nonmono:", np.nonzero(np.diff(v)<0)[0][:5])
    v = ref['less_sharp'][:,3].astype(np.int32); print(" oracle ls nonmono:", np.nonzero(np.diff(v)<0)[0][:5])
