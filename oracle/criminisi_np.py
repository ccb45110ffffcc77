"""NumPy restatement of the reference's Criminisi fill loop, one method per reference method.

TEST INFRASTRUCTURE ONLY (see ``oracle/cv2_restated.py`` for the rules): this is the checker
for the HIP engine, never the thing shipped or measured as the product.

PARITY UNPINNED: the reference cannot be imported here (``import cv2`` fails with
ModuleNotFoundError) and ships no fixtures; the OpenCV calls are replaced by the restatements in
``cv2_restated.py``.  NumPy calls (``np.sum``, ``np.gradient``, ``argmax``/``argmin``) are the
very functions the reference calls, so their summation order and tie-breaking are anchored to
the real library.

Follows /root/reference/src/criminisi_inpainter.py: state :18-38, loop :40-88, front :94-104,
priorities :106-112, confidence :114-127, data :129-136, normal :138-164, gradient :166-214,
target :216-218, source search :220-265, update :267-308, patch geometry :310-323.
"""
import numpy as np

from . import cv2_restated as cvr


class CriminisiOracle:
    def __init__(self, image, mask, patch_size=9):
        # criminisi_inpainter.py:18-38
        self.image = image
        self.mask = mask.round()
        self.patch_size = patch_size
        self.iheight, self.iwidth = self.image.shape[:2]
        self.working_image = np.copy(self.image)
        self.working_mask = np.copy(self.mask)
        self.front = np.zeros([self.iheight, self.iwidth])
        self.confidence = (1.0 - self.mask)
        self.data = np.zeros([self.iheight, self.iwidth])
        self.priority = np.zeros([self.iheight, self.iwidth])
        # per-iteration record: (target_r, target_c, src_r0, src_c0, cost, remaining_after)
        self.trace = []

    # -- driver ------------------------------------------------------------------------------
    def inpaint(self, max_iters=None):
        # criminisi_inpainter.py:40-88
        if self.image.shape[:2] != self.mask.shape:
            raise AttributeError('mask and image must be of the same size')
        c = 0
        while True:
            self.step()
            remaining = int(np.sum(self.working_mask))
            if remaining == 0:
                break
            c += 1
            if max_iters is not None and c >= max_iters:
                break
        return self.working_image

    def step(self):
        """One pass of the loop body, criminisi_inpainter.py:53-74."""
        self._find_front()
        self._update_priorities()
        target_pixel = self._find_highest_priority_pixel()
        source_patch, cost = self._find_source_patch(target_pixel)
        self._update_image(target_pixel, source_patch)
        remaining = int(np.sum(self.working_mask))
        self.trace.append((int(target_pixel[0]), int(target_pixel[1]),
                           int(source_patch[0][0]), int(source_patch[1][0]),
                           float(cost), remaining))
        return remaining

    # -- stages ------------------------------------------------------------------------------
    def _find_front(self):
        # criminisi_inpainter.py:103-104 (the "< 0" line is a no-op on uint8)
        self.front = cvr.laplacian_ksize3_u8(self.working_mask.astype(np.uint8))

    def _update_priorities(self):
        # criminisi_inpainter.py:106-112
        self._update_confidence()
        self._update_data()
        self.priority = self.confidence * self.data * self.front

    def _update_confidence(self):
        # criminisi_inpainter.py:114-127
        new_confidence = np.copy(self.confidence)
        for point in np.argwhere(self.front > 0):
            patch = self._get_patch(point)
            new_confidence[point[0], point[1]] = np.sum(np.sum(
                self.confidence[patch[0][0]:patch[0][1] + 1,
                                patch[1][0]:patch[1][1] + 1])) / self.patch_size ** 2
        self.confidence = new_confidence

    def _update_data(self):
        # criminisi_inpainter.py:129-136
        normal = self._calc_normal_matrix()
        gradient = self._calc_gradient_matrix()
        normal_gradient = normal * gradient
        self.data = np.sqrt(normal_gradient[:, :, 0] ** 2 + normal_gradient[:, :, 1] ** 2) + 0.001

    def _calc_normal_matrix(self):
        # criminisi_inpainter.py:138-164
        x_kernel = [[.25, 0, -.25], [.5, 0, -.5], [.25, 0, -.25]]
        y_kernel = [[-.25, -.5, -.25], [0, 0, 0], [.25, .5, .25]]
        x_normal = cvr.filter2d_3x3_f64(self.working_mask.astype(float), x_kernel)
        y_normal = cvr.filter2d_3x3_f64(self.working_mask.astype(float), y_kernel)
        normal = np.dstack((x_normal, y_normal))
        height, width = normal.shape[:2]
        norm = np.sqrt(y_normal ** 2 + x_normal ** 2)
        norm = norm.reshape(height, width, 1).repeat(2, axis=2)
        norm[norm == 0] = 1
        return normal / norm

    def _calc_gradient_matrix(self):
        # criminisi_inpainter.py:166-214
        height, width = self.working_image.shape[:2]
        grey_image = cvr.bgr2gray_u8(self.working_image)
        grey_image[self.working_mask > 0] = 0
        gradient = np.nan_to_num(np.array(np.gradient(grey_image)))
        gradient_val = np.sqrt(gradient[0] ** 2 + gradient[1] ** 2)
        max_gradient = np.zeros([height, width, 2])
        for p in np.argwhere(self.front > 0):
            patch = self._get_patch(p)
            sl = (slice(patch[0][0], patch[0][1] + 1), slice(patch[1][0], patch[1][1] + 1))
            patch_y_gradient = gradient[0][sl]
            patch_x_gradient = gradient[1][sl]
            patch_gradient_val = gradient_val[sl]
            patch_max_pos = np.unravel_index(patch_gradient_val.argmax(), patch_gradient_val.shape)
            max_gradient[p[0], p[1], 0] = patch_y_gradient[patch_max_pos]
            max_gradient[p[0], p[1], 1] = patch_x_gradient[patch_max_pos]
        return max_gradient

    def _find_highest_priority_pixel(self):
        # criminisi_inpainter.py:216-218
        return np.unravel_index(self.priority.argmax(), self.priority.shape)

    def _search_costs(self, target_pixel):
        """Cost surface of criminisi_inpainter.py:220-262 with the exact-integer SSD."""
        target_patch = self._get_patch(target_pixel)
        ph = 1 + target_patch[0][1] - target_patch[0][0]
        pw = 1 + target_patch[1][1] - target_patch[1][0]
        lab_image = cvr.bgr2lab_u8(self.working_image).astype(np.int64)
        sl = (slice(target_patch[0][0], target_patch[0][1] + 1),
              slice(target_patch[1][0], target_patch[1][1] + 1))
        target_mask_2d = (1 - self.working_mask[sl]).astype(np.int64)
        target_data = lab_image[sl]
        total_ssd = np.zeros((self.iheight - ph + 1, self.iwidth - pw + 1), dtype=np.float64)
        for c in range(3):
            I = lab_image[:, :, c]
            T = target_data[:, :, c]
            tm_T = target_mask_2d * T
            term1 = cvr.match_template_ccorr_exact(I ** 2, target_mask_2d)
            term2 = cvr.match_template_ccorr_exact(I, tm_T)
            total_ssd += term1 - 2 * term2
        mask_sum = cvr.window_sum_topleft(self.working_mask.astype(np.uint8), ph, pw)
        total_ssd[mask_sum[:total_ssd.shape[0], :total_ssd.shape[1]] > 0] = np.inf
        ty = (target_patch[0][0] + target_patch[0][1]) / 2.0
        tx = (target_patch[1][0] + target_patch[1][1]) / 2.0
        ys = np.arange(total_ssd.shape[0]) + ph / 2.0
        xs = np.arange(total_ssd.shape[1]) + pw / 2.0
        total_ssd += np.sqrt((ys[:, None] - ty) ** 2 + (xs[None, :] - tx) ** 2)
        return total_ssd, ph, pw

    def _find_source_patch(self, target_pixel):
        # criminisi_inpainter.py:220-265
        total_ssd, ph, pw = self._search_costs(target_pixel)
        loc = np.unravel_index(np.argmin(total_ssd), total_ssd.shape)
        return [[loc[0], loc[0] + ph - 1], [loc[1], loc[1] + pw - 1]], total_ssd[loc]

    def _update_image(self, target_pixel, source_patch):
        # criminisi_inpainter.py:267-308
        target_patch = self._get_patch(target_pixel)
        tsl = (slice(target_patch[0][0], target_patch[0][1] + 1),
               slice(target_patch[1][0], target_patch[1][1] + 1))
        ssl = (slice(source_patch[0][0], source_patch[0][1] + 1),
               slice(source_patch[1][0], source_patch[1][1] + 1))
        pixels_positions = np.argwhere(self.working_mask[tsl] > 0) + \
            [target_patch[0][0], target_patch[1][0]]
        patch_confidence = self.confidence[target_pixel[0], target_pixel[1]]
        for point in pixels_positions:
            self.confidence[point[0], point[1]] = patch_confidence
        mask = self.working_mask[tsl].astype(np.uint8)
        bgr_mask = np.repeat(mask[:, :, None], 3, axis=2)      # cv2.COLOR_GRAY2BGR
        source_data = self.working_image[ssl]
        target_data = self.working_image[tsl]
        new_data = source_data * bgr_mask + target_data * (1 - bgr_mask)
        self.working_image[tsl] = new_data
        self.working_mask[tsl] = 0

    def _get_patch(self, point):
        # criminisi_inpainter.py:310-323
        half_patch_size = (self.patch_size - 1) // 2
        height, width = self.working_image.shape[:2]
        return [[max(0, point[0] - half_patch_size), min(point[0] + half_patch_size, height - 1)],
                [max(0, point[1] - half_patch_size), min(point[1] + half_patch_size, width - 1)]]


def brute_force_costs(oracle, target_pixel):
    """Independent form of the cost surface: sum k*(L-T)^2 - sum k*T^2, window by window.
    O(N*p*p); used to cross-check ``_search_costs`` on small images."""
    tp = oracle._get_patch(target_pixel)
    ph = 1 + tp[0][1] - tp[0][0]
    pw = 1 + tp[1][1] - tp[1][0]
    lab = cvr.bgr2lab_u8(oracle.working_image).astype(np.int64)
    m = oracle.working_mask.astype(np.int64)
    k = 1 - m[tp[0][0]:tp[0][1] + 1, tp[1][0]:tp[1][1] + 1]
    T = lab[tp[0][0]:tp[0][1] + 1, tp[1][0]:tp[1][1] + 1]
    const = int(np.sum(k[:, :, None] * T * T))
    oh, ow = oracle.iheight - ph + 1, oracle.iwidth - pw + 1
    ty = (tp[0][0] + tp[0][1]) / 2.0
    tx = (tp[1][0] + tp[1][1]) / 2.0
    out = np.empty((oh, ow), dtype=np.float64)
    for y in range(oh):
        for x in range(ow):
            if m[y:y + ph, x:x + pw].any():
                out[y, x] = np.inf
                continue
            d = lab[y:y + ph, x:x + pw] - T
            s = int(np.sum(k[:, :, None] * d * d)) - const
            out[y, x] = float(s) + np.sqrt((y + ph / 2.0 - ty) ** 2 + (x + pw / 2.0 - tx) ** 2)
    return out
