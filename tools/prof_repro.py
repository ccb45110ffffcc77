import os, sys, ctypes
sys.path.insert(0, "image-processing_amd")
import numpy as np, torch
import crim_hip_capi as capi, crim_synth
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
order=sys.argv[2] if len(sys.argv)>2 else "sp"
imgs=np.stack([crim_synth.xray_like(256,256,1000+i%8) for i in range(B)]); masks=np.stack([crim_synth.blob_mask(256,256,1000+i%8,0.10,9) for i in range(B)])
d_img=torch.from_numpy(imgs).cuda(); d_mask=torch.from_numpy(masks).cuda(); torch.cuda.synchronize()
lib=capi.load_library()
lib.crim_debug_state.argtypes=[ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
def state(e,i):
    st=np.zeros(16,np.int32); lib.crim_debug_state(e._h,i,st.ctypes.data); return st[:9].tolist()
F={"s":(capi.FLAG_PROFILE_SOLO|capi.FLAG_NO_TRACE,"solo"),"p":(capi.FLAG_PROFILE|capi.FLAG_NO_TRACE,"profile"),"n":(capi.FLAG_NO_TRACE,"plain"),"e":(capi.FLAG_EAGER|capi.FLAG_NO_TRACE,"eager")}
for c in order:
    fl,name=F[c]
    with capi.Engine(256,256,9,batch=B,flags=fl) as pe:
        pe.upload_device(d_img.data_ptr(), d_mask.data_ptr())
        print(name, "after upload", state(pe,0), state(pe,B-1))
        r=pe.run(30)
        print(name, "run", r, "state", state(pe,0), state(pe,B-1), "profile", pe.profile() if c in "sp" else None)
