import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import c_oracle, ref_port
from roboticattack_amd import ops, synthetic
DEV='cuda:0'
d=np.load('tests/golden/resize_base50.npz')
B=int(d['batch']); patch=d['patch']; sizes=d['sizes']
pdesc_n,total=ops.make_pdesc(sizes)
pdesc=torch.from_numpy(pdesc_n).to(DEV)
packed=ops.patch_resize_fwd(torch.from_numpy(patch).to(DEV),pdesc,total).cpu().numpy()
o_packed=c_oracle.patch_resize_fwd(patch,pdesc_n,total)
diff=packed!=o_packed
print('resize mismatches',int(diff.sum()),'of',packed.size,'max abs',np.abs(packed-o_packed).max())
if diff.any():
    idx=np.nonzero(diff)[0][:10]; print(idx, packed[idx], o_packed[idx])
    for (h,w,off,_) in pdesc_n:
        dd=diff[off:off+3*h*w].reshape(3,h,w); print((h,w),'mism',int(dd.sum()), 'rows',np.unique(np.nonzero(dd)[1])[:10],'cols',np.unique(np.nonzero(dd)[2])[:10])
imgs=synthetic.synth_images(int(d['img_seed']),B,str(d['img_kind']))
xy=torch.from_numpy(d['xy']).to(DEV); th=torch.from_numpy(d['theta'].reshape(-1,6)).to(DEV)
mh=(int(pdesc_n[:,0].max()),int(pdesc_n[:,1].max()))
out,keep=ops.patch_apply_fwd_multi(torch.from_numpy(imgs).to(DEV),torch.from_numpy(o_packed).to(DEV),pdesc,mh,xy,th,True,0)
_,ob,ok=c_oracle.patch_apply_fwd_multi(imgs,o_packed,pdesc_n,d['xy'],d['theta'],1,0)
got=out.view(torch.int16).cpu().numpy().view(np.uint16)
print('K1 multi mismatches (oracle packed in)',int((got!=ob).sum()))
k=np.unpackbits(keep.cpu().numpy(),axis=-1,bitorder='little')
print('keep mism',int((k!=ok).sum()))
