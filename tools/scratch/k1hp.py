import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from roboticattack_amd import benchmarks, ops, synthetic, _lib
L = _lib.lib()
vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
L.vaa_patch_apply_fwd_hp.restype = i32
L.vaa_patch_apply_fwd_hp.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp, vp]
dev = torch.device("cuda:0")
for B in (8, 64, 128):
    img = torch.from_numpy(synthetic.synth_images(1234, min(B,64), "noise")).to(dev)
    if B > 64: img = img.repeat(2,1,1,1)[:B].contiguous()
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    out = torch.empty((B, 6, 224, 224), dtype=torch.bfloat16, device=dev)
    keep = torch.empty((B, 3, 224 * 224 // 8), dtype=torch.uint8, device=dev)
    ref, rk = ops.patch_apply_fwd(img, patch, xy, th, True)
    xyc, thc = np.ascontiguousarray(xy_n), np.ascontiguousarray(th_n)
    def hp():
        rc = L.vaa_patch_apply_fwd_hp(img.data_ptr(), patch.data_ptr(), xy.data_ptr(), th.data_ptr(), xyc.ctypes.data, thc.ctypes.data, B, 50, 50, 1, 0,
                                      ops._MEAN, ops._STD, out.data_ptr(), keep.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    hp(); torch.cuda.synchronize()
    print(B, "out mism", int((out.view(torch.int16) != ref.view(torch.int16)).sum()), "keep mism", int((keep != rk).sum()), _lib.lib().vaa_last_error())
    t_hp = benchmarks._time(hp, 30)
    t_dev = benchmarks._time(lambda: ops.patch_apply_fwd(img, patch, xy, th, True), 30)
    print(B, "kernarg params us", round(t_hp[0]*1e6,2), "device params us", round(t_dev[0]*1e6,2))
