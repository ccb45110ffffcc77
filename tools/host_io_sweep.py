import sys, os
sys.path[:0]=['/root/repo','/root/repo/image-processing_amd']
import bench, crim_hip_capi as capi
imgs, masks = bench.make_batch(8192, 256, 256, 2, 0, False)
for sh, en in ((2,2),(2,3),(2,4),(3,4)):
    v,_ = bench.streamed_images_per_s(capi, 0, imgs, masks, 3, shares=sh, engines_n=en)
    print(sh, en, round(v))
