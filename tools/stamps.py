"""Diagnostic: per-phase wall-clock stamps of the iteration kernels (needs the -DCRIM_STAMPS
build in image-processing_amd/lib_dbg)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
import crim_hip_capi as capi
capi.LIB_PATH = os.path.join(ROOT, "image-processing_amd", "lib_dbg", "libcriminisi_hip.so")
import crim_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
imgs = np.stack([crim_synth.xray_like(H, W, 2000 + i) for i in range(min(B, 8))])
masks = np.stack([crim_synth.blob_mask(H, W, 2000 + i, 0.10, 9) for i in range(min(B, 8))])
imgs = np.concatenate([imgs] * (B // len(imgs) + 1))[:B]; masks = np.concatenate([masks] * (B // len(masks) + 1))[:B]
lib = capi.load_library()
lib.crim_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
if H > 256:   # one large image with a centred square hole, as BASELINE configs[3] / [4]
    imgs = crim_synth.xray_like(H, W, 4000 + H, colour=True)[None]
    masks = crim_synth.square_hole_mask(H, W, H // 2)[None]
eng = capi.Engine(H, W, 9, batch=B, flags=0 if os.environ.get("STAMPS_GRAPH") else capi.FLAG_EAGER)
eng.upload(imgs, masks)
eng.run(max_iters=40)
acc = {}
for it in range(20):
    eng.run(max_iters=1)
    st = np.zeros(64, np.int64); lib.crim_debug_stamps(eng._h, st.ctypes.data)
    for name, slots in (("priority", (0, 1, 4, 5, 6, 7, 2, 3)), ("select", range(8, 12)), ("update", range(16, 24)), ("search (entry, arg-max, template, first plane, passes, pricing, end)", range(32, 39))):
        v = st[list(slots)]
        d = np.diff(v) * 10  # ns (100 MHz)
        acc.setdefault(name, []).append(d)
    acc.setdefault("step near block 0 (entry, arg-min, source pixels, gradient, commit, front list, evaluation, emit)", []).append(np.diff(st[40:48]) * 10)
    acc.setdefault("step bookkeeping (entry, snapshot, fill, validity, tile counts, refresh+state)", []).append(np.diff(st[48:54]) * 10)
    acc.setdefault("search entry -> near entry -> book end", []).append(np.array([st[40] - st[32], st[53] - st[40], st[47] - st[40]]) * 10)
    acc.setdefault("gaps priority->search, search->update, update->next priority", []).append(np.array([st[32] - st[3], st[16] - st[38], 0]) * 10)
for k, v in acc.items():
    print(k, "mean ns per phase:", np.mean(v, axis=0).astype(int).tolist(), "total", int(np.mean(np.sum(v, axis=1))))
st = np.zeros(64, np.int64); lib.crim_debug_stamps(eng._h, st.ctypes.data)
if st[57] > st[56] and st[38] > st[32]:
    print("search block in the middle of the grid: %d shader cycles in %.2f us -> %.2f GHz" % (st[57] - st[56], (st[38] - st[32]) * 0.01, (st[57] - st[56]) / ((st[38] - st[32]) * 10.0)))
if st[51]:
    print("search: priced windows %d in %d blocks -> %.2f per block" % (st[50], st[51], st[50] / st[51]))
if hasattr(lib, "crim_debug_block_stamps"):
    lib.crim_debug_block_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    bs = np.zeros(8192, np.int64); lib.crim_debug_block_stamps(eng._h, bs.ctypes.data)
    ent, end = bs[0::2], bs[1::2]
    on = (ent > 0) & (end > 0)
    if on.any():
        t0 = ent[on].min()
        print("search blocks recorded: %d; entry after the first block's (us): median %.2f max %.2f; block life (us): median %.2f p90 %.2f max %.2f; last end after first entry %.2f us"
              % (on.sum(), np.median(ent[on] - t0) * 0.01, (ent[on] - t0).max() * 0.01, np.median((end - ent)[on]) * 0.01,
                 np.percentile((end - ent)[on], 90) * 0.01, (end - ent)[on].max() * 0.01, (end[on].max() - t0) * 0.01))
        life = (end - ent)[on] * 0.01
        print("   block lives by decile:", np.percentile(life, [0, 10, 25, 50, 75, 90, 100]).round(2).tolist())
