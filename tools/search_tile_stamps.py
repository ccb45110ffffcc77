"""Diagnostic: where one tile of the batch search spends its time (block in the middle of the grid, its fourth
tile), with every image live on one lane.  Needs a -DCRIM_STAMPS build with the per-tile stamps compiled in
(CRIM_HIP_LIB=image-processing_amd/lib_dbg/libcriminisi_hip.so, built with CRIM_LIB_OUT=../lib_dbg EXTRA_HIPCC_FLAGS=-DCRIM_STAMPS bash build.sh).
python3 tools/search_tile_stamps.py [batch]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi, crim_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = [(crim_synth.xray_like(256, 256, 1000 + i), crim_synth.blob_mask(256, 256, 1000 + i, 0.10, 9)) for i in range(64)]
imgs = np.stack([base[i % 64][0] for i in range(B)]); masks = np.stack([base[i % 64][1] for i in range(B)])
lib = capi.load_library()
lib.crim_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
names = ["top", "bar0", "init", "pass0", "bar1", "frag1", "pass1", "bar2", "shift2", "pass2", "max", "bar3", "end"]
with capi.Engine(256, 256, 9, batch=B, flags=capi.FLAG_ONE_LANE | capi.FLAG_EAGER | capi.FLAG_NO_TRACE) as eng:
    eng.upload(imgs, masks)
    eng.run(max_iters=10)
    rows = []
    for it in range(30):
        eng.run(max_iters=1)
        st = np.zeros(64, np.int64); lib.crim_debug_stamps(eng._h, st.ctypes.data)
        rows.append(np.diff(st[40:53]) * 10)
    m = np.median(np.array(rows), axis=0)
    print("ns per phase (median of 30 iterations), tile 3 of the middle block:")
    for n, v in zip(names[1:], m): print("  -> %-7s %6d" % (n, v))
    print("  tile total %d ns" % int(m.sum()))
