"""Mirror of `cet_pick/detectors/base_detector.py` (`BaseDetector.__init__` :16-34, `run` :62-106)."""
import time

import torch

from ..models.model import create_model, load_model


class BaseDetector(object):
    def __init__(self, opt):
        if opt.gpus[0] < 0:
            raise RuntimeError("cet_pick_amd detectors run on the MI355X only: --gpus must name a device "
                               "(the reference's CPU mode, --gpus -1, is outside this build)")
        opt.device = torch.device("cuda")
        print("Creating model...")
        self.model = create_model(opt.arch, opt.heads, opt.head_conv, last_k=getattr(opt, "last_k", 3))
        if getattr(opt, "load_model", ""):
            self.model = load_model(self.model, opt.load_model)
        if opt.task == "semiclass":
            self.model.fill()
        self.model = self.model.to(opt.device)
        self.model.eval()
        self.max_per_image = 900
        self.opt = opt
        self.pause = True

    def process(self, images, return_time=False):
        raise NotImplementedError

    def post_process(self, dets, meta, scale=1):
        raise NotImplementedError

    def save_detection(self, dets, path, meta, prefix="", name=""):
        raise NotImplementedError

    def run(self, image_or_path_or_tensor, meta=None):
        """base_detector.py:62-106: process -> post_process -> save_detection, with the reference's timing keys."""
        start_time = time.time()
        load_time = time.time() - start_time
        images = image_or_path_or_tensor
        if self.opt.task != "semiclass":
            images = images.to(self.opt.device, non_blocking=True)
        pre_process_time = time.time()
        output, dets, hm, forward_time = self.process(images, return_time=True)
        depth = hm.size(2)
        net_time = forward_time - pre_process_time
        decode_time = time.time()
        dec_time = decode_time - forward_time
        if hasattr(self, "begin_hm_download"):
            self.begin_hm_download(hm)                  # the heat-map's device-to-host copy runs under the post-processing below
        dets, name = self.post_process(dets, meta, z_dim_tot=depth)
        torch.cuda.synchronize()
        post_time = time.time()
        self.save_detection(hm, dets, self.opt.out_path, meta, name=name)
        end_time = time.time()
        # (not reference keys: the two host stages behind the decode, for the bench's entry_point_infer record)
        self.last_stages = {"post": post_time - decode_time, "save": end_time - post_time}
        return {"tot_time": end_time - start_time, "load": load_time, "pre": 0, "net": net_time, "dec": dec_time}
