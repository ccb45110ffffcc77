"""Host logic of the directory driver (file listing, start_from, load failures, batching and
padding, same-name outputs) with a stand-in engine: no GPU, no arithmetic."""
import os

import numpy as np

import crim_batch_driver as drv


class FakeEngine:
    """Records the calls; 'inpaints' by adding 1 to every byte of the real images."""
    log = []

    def __init__(self, H, W, batch):
        self.H, self.W, self.batch = H, W, batch
        self.slots = [None] * batch
        self.auto = [False] * batch
        FakeEngine.log.append(("create", H, W, batch))

    def upload_automask(self, images, first=0, threshold=10, border_width=10, dilation_size=4):
        for i, im in enumerate(images):
            self.slots[first + i] = im
            self.auto[first + i] = True
        FakeEngine.log.append(("automask", first, len(images), dilation_size))

    def upload(self, images, masks, first=0):
        assert not masks.any(), "padding must carry empty masks"
        for i, im in enumerate(images):
            self.slots[first + i] = im
            self.auto[first + i] = False
        FakeEngine.log.append(("pad", first, len(images)))

    def run(self, max_iters=0):
        assert all(s is not None for s in self.slots), "every slot of the batch must be uploaded"
        FakeEngine.log.append(("run",))

    def download(self, first=0, count=None, want_conf=False, want_mask=False):
        out = np.stack([self.slots[first + i] + 1 for i in range(count)])
        return out, None, None

    def close(self):
        FakeEngine.log.append(("close", self.H, self.W))


def test_directory_walk_batches_by_shape_and_keeps_names(tmp_path):
    FakeEngine.log = []
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    names = ["a.jpg", "b.jpg", "c.jpg", "d.jpg", "e.jpg", "notes.txt", "f.JPG", "broken.jpg"]
    for n in names:
        (src / n).write_bytes(b"x")
    shapes = {"a.jpg": (8, 8), "b.jpg": (8, 8), "c.jpg": (6, 10), "d.jpg": (8, 8), "e.jpg": (8, 8)}

    def reader(path):
        n = os.path.basename(path)
        if n == "broken.jpg":
            return None
        h, w = shapes[n]
        return np.full((h, w, 3), ord(n[0]) - ord("a"), np.uint8)

    written = {}
    msgs = []
    st = drv.inpaint_directory(str(src), str(dst), batch=2, io_threads=2, log=msgs.append,
                               make_engine=FakeEngine, reader=reader,
                               writer=lambda path, img: written.__setitem__(os.path.basename(path), img))
    assert st["images"] == 5 and st["failed"] == 1
    assert msgs == ["broken.jpg failed to load."]
    assert sorted(written) == ["a.jpg", "b.jpg", "c.jpg", "d.jpg", "e.jpg"]   # only *.jpg, same names
    for n, img in written.items():
        assert img.shape[:2] == shapes[n] and (img == ord(n[0]) - ord("a") + 1).all()
    assert os.path.isdir(dst)
    # one engine per shape; the odd 6x10 image is flushed with one padding slot
    creates = [e for e in FakeEngine.log if e[0] == "create"]
    assert sorted(creates) == [("create", 6, 10, 2), ("create", 8, 8, 2)]
    assert ("pad", 1, 1) in FakeEngine.log
    assert sum(1 for e in FakeEngine.log if e[0] == "run") == 3


def test_start_from_is_inclusive_and_follows_listing_order(tmp_path):
    src = tmp_path / "in"
    src.mkdir()
    for n in ["p.jpg", "q.jpg", "r.jpg"]:
        (src / n).write_bytes(b"x")
    order = [f for f in os.listdir(src) if f.endswith(".jpg")]
    assert drv.list_tags(str(src)) == order
    assert drv.list_tags(str(src), start_from=order[1]) == order[1:]
    assert drv.list_tags(str(src), start_from="absent.jpg") == []
