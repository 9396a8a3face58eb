import sys, os, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import c_oracle, ref_port
from roboticattack_amd import ops, synthetic
DEV='cuda:0'
B=4
imgs=synthetic.synth_images(505,B,'smooth')
torch.manual_seed(42); patch0=torch.rand(3,100,100)
random.seed(7); np.random.seed(7)
sizes,xy,theta=ref_port.draw_params_resized(B,100,100,True)
print(sizes.tolist(), xy.tolist())
pdesc_n,total=ops.make_pdesc(sizes)
pdesc=torch.from_numpy(pdesc_n).to(DEV)
packed=ops.patch_resize_fwd(patch0.to(DEV),pdesc,total).cpu().numpy()
o_packed=c_oracle.patch_resize_fwd(patch0.numpy(),pdesc_n,total)
print('hip vs c-oracle resize mism',int((packed!=o_packed).sum()))
for (h,w,off,_) in pdesc_n:
    ref=ref_port.resize_patch(patch0,int(h),int(w)).numpy().ravel()
    print((h,w),'torch-cpu vs c-oracle mism',int((ref!=o_packed[off:off+3*h*w]).sum()),'maxabs',np.abs(ref-o_packed[off:off+3*h*w]).max())
pix_c=ref_port.apply_random_patch_batch_resized(imgs,patch0,sizes,xy,theta,True).to(torch.bfloat16)
_,ob,_=c_oracle.patch_apply_fwd_multi(imgs,o_packed,pdesc_n,xy,theta,1,0)
pc=pix_c.view(torch.int16).numpy().view(np.uint16)
print('ref_port vs c-oracle K1 mism',int((pc!=ob).sum()))
mh=(int(pdesc_n[:,0].max()),int(pdesc_n[:,1].max()))
out,_=ops.patch_apply_fwd_multi(torch.from_numpy(imgs).to(DEV),torch.from_numpy(packed).to(DEV),pdesc,mh,torch.from_numpy(xy).to(DEV),torch.from_numpy(theta.reshape(-1,6)).to(DEV),True,0)
got=out.view(torch.int16).cpu().numpy().view(np.uint16)
print('hip vs c-oracle K1 mism',int((got!=ob).sum()))
print(torch.__config__.show()[:600])
