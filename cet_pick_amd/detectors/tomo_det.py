"""Mirror of `cet_pick/detectors/tomo_det.py` (`TomodetDetector.process` :23-41, `post_process` :43-52,
`save_detection` :54-95): network -> `_sigmoid` -> `tomo_decode` fused into one pass over the logits, then the
reference's host-side filters and output files (`{name}.txt` as x<TAB>z<TAB>y[<TAB>score], `{name}_hm.mrc`)."""
import os
import time

import numpy as np
import torch

from ..models.decode import sigmoid_tomo_decode
from ..utils import mrc as mrcio
from ..utils.post_process import tomo_fiber_postprocess, tomo_post_process
from .base_detector import BaseDetector


class TomodetDetector(BaseDetector):
    def process(self, images, return_time=False):
        with torch.no_grad():
            output = self.model(images)[-1]
            logits = output["hm"]
            torch.cuda.synchronize()
            forward_time = time.time()
            # `_sigmoid` (in place in the reference) and `tomo_decode` in one pass; output['hm'] becomes the
            # clamped heat-map exactly as the in-place sigmoid leaves it (tomo_det.py:33-37)
            hm, dets = sigmoid_tomo_decode(logits.contiguous(), kernel=self.opt.nms, K=self.opt.K,
                                           if_fiber=self.opt.fiber)
            output["hm"] = hm
        if return_time:
            return output, dets, hm, forward_time
        return output, dets, hm

    def post_process(self, dets, meta, scale=1, z_dim_tot=128):
        dets = dets.detach().cpu().numpy().reshape(1, -1, dets.shape[2])
        dets[:, :, :2] *= self.opt.down_ratio
        preds = tomo_post_process(dets, z_dim_tot=z_dim_tot)[0]
        return preds, meta["name"][0]

    def begin_hm_download(self, hm):
        """The heat-map's way to the host, started right behind the decode (BaseDetector.run) so that it overlaps the detections'
        post-processing: the (D, H', W') -> (H', D, W') swap of tomo_det.py:60 and the NaN test of :64 run on the device, the payload
        goes into a pinned buffer with one non-blocking copy, and `_hm.mrc` is written from that buffer (round 6; before: a pageable
        67 MB copy, a host transpose, a host NaN scan, an astype copy and a tobytes copy - 84 of the 137 ms a 256 x 512 x 512 tomogram
        took)."""
        vol = hm.detach()[0][0]
        swapped = vol.permute(1, 0, 2).contiguous()                       # (H', D, W'), as np.swapaxes(hm, 1, 0) leaves it
        nan_flag = torch.isnan(swapped).any().reshape(1).to(torch.uint8)
        n = swapped.numel()
        pin = getattr(self, "_hm_pin", None)
        if pin is None or pin.numel() < n:
            pin = self._hm_pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
            self._flag_pin = torch.empty(1, dtype=torch.uint8, pin_memory=True)
        pin[:n].copy_(swapped.reshape(-1), non_blocking=True)
        self._flag_pin.copy_(nan_flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hm_pending = (hm.data_ptr(), tuple(swapped.shape), ev, swapped)      # (`swapped` stays alive until the copy has run)

    def _hm_on_host(self, hm):
        pend = getattr(self, "_hm_pending", None)
        if pend is None or pend[0] != hm.data_ptr():
            self.begin_hm_download(hm)
            pend = self._hm_pending
        self._hm_pending = None
        _, shape, ev, _keep = pend
        ev.synchronize()
        if int(self._flag_pin[0]):
            raise ValueError("Output contains NaN values")
        n = int(np.prod(shape))
        return self._hm_pin[:n].numpy().reshape(shape)                    # a view of the pinned buffer: no copy

    def save_detection(self, hm, dets, path, meta, prefix="", name=""):
        if not os.path.exists(path):
            os.mkdir(path)
        max_z, max_y, max_x = hm.shape[2], hm.shape[3], hm.shape[4]
        max_x, max_y = max_x * 2, max_y * 2
        mrcio.write(os.path.join(path, "{}_hm.mrc".format(name)), self._hm_on_host(hm), is_vol=True)
        opt = self.opt
        pre_coords = []
        with open(os.path.join(path, "{}.txt".format(name)), "w+") as out_detect:
            for _, v in dets.items():
                for c in v:
                    x, y, z, score = int(np.floor(c[0])), int(np.floor(c[1])), int(np.floor(c[2])), float(c[3])
                    if (score > opt.out_thresh and opt.cutoff_z <= z <= max_z - opt.cutoff_z
                            and 20 < x < max_x - 20 and 20 < y < max_y - 20):
                        if opt.compress:
                            z = int(z) * 2
                        if opt.fiber:
                            pre_coords.append([x, y, z])
                        elif not opt.with_score:
                            print(str(x) + "\t" + str(z) + "\t" + str(y), file=out_detect)
                        else:
                            print(str(x) + "\t" + str(z) + "\t" + str(y) + "\t" + str(score), file=out_detect)
            if opt.fiber:
                for c in tomo_fiber_postprocess(pre_coords, distance_cutoff=opt.distance_cutoff, res_cutoff=opt.r2_cutoff,
                                                curvature_cutoff=opt.curvature_cutoff, scale=opt.distance_scale):
                    print(str(c[0]) + "\t" + str(c[1]) + "\t" + str(c[2]), file=out_detect)
        if getattr(opt, "spike", False):
            # tomo_det.py:89-95 calls tomo_group_postprocess without importing it (NameError in the reference)
            raise NotImplementedError("--spike post-processing is not runnable in the reference (tomo_det.py:89-90)")

    def merge_outputs(self, detections):
        scores = detections[:, -1]
        if len(scores) > self.max_per_image:
            kth = len(scores) - self.max_per_image
            thresh = np.partition(scores, kth)[kth]
            detections = detections[scores >= thresh]
        return detections
