"""Flag surface of the reference (`cet_pick/opts.py:13-331`), kept as a drop-in: same flag names,
defaults and derived fields (`gpus`, `lr_step`, `head_conv`, `chunk_sizes`, directories, `heads`).

Table-driven restatement; reference defects are fixed, not mirrored: the `--warm --cosine` branch of
the reference reads an undefined `opt.learning_rate` and never imports `math` (opts.py:216-224) -
here it uses `opt.lr`.  `--dist-backend nccl` is RCCL under ROCm.
"""
import argparse
import math
import os


def list_of_floats(arg):
    return list(map(float, arg.split(",")))


# (flag, kwargs) in the reference's order.  "S" = store_true.
_S = {"action": "store_true"}
_FLAGS = [
    ("--dataset", dict(default="semi")), ("--exp_id", dict(default="default")), ("--test", _S),
    ("--debug", dict(type=int, default=4)), ("--load_model", dict(default="")),
    ("--pretrain_model", dict(default="")), ("--resume", _S), ("--fiber", _S), ("--spike", _S),
    # system
    ("--gpus", dict(default="0")), ("--num_workers", dict(type=int, default=4)),
    ("--not_cuda_benchmark", _S), ("--seed", dict(type=int, default=317)),
    # distributed
    ("--world-size", dict(default=-1, type=int)), ("--rank", dict(default=-1, type=int)),
    ("--dist-url", dict(default="env://", type=str)), ("--dist-backend", dict(default="nccl", type=str)),
    ("--local_rank", dict(default=-1, type=int)),
    # log
    ("--print_iter", dict(type=int, default=0)), ("--hide_data_time", _S), ("--save_all", _S),
    ("--metric", dict(default="loss")), ("--vis_thresh", dict(type=float, default=0.3)),
    ("--debugger_theme", dict(default="white", choices=["white", "black"])),
    # model
    ("--arch", dict(default="unet_4")), ("--last_k", dict(type=int, default=3)),
    ("--head_conv", dict(type=int, default=-1)), ("--down_ratio", dict(type=int, default=2)),
    ("--pretrained_model", dict(type=str, default=None)),
    # input
    ("--input_res", dict(type=int, default=-1)), ("--input_h", dict(type=int, default=-1)),
    ("--input_w", dict(type=int, default=-1)),
    # train
    ("--lr", dict(type=float, default=1e-3)), ("--lr_step", dict(type=str, default="200, 400, 600")),
    ("--num_epochs", dict(type=int, default=140)), ("--lr_decay_rate", dict(type=float, default=0.1)),
    ("--cosine", _S), ("--warm", _S), ("--contrastive", _S),
    ("--batch_size", dict(type=int, default=1)), ("--master_batch_size", dict(type=int, default=-1)),
    ("--num_iters", dict(type=int, default=-1)), ("--val_intervals", dict(type=int, default=5)),
    ("--trainval", _S), ("--bbox", dict(type=int, default=32)),
    ("--translation_ratio", dict(type=float, default=0.5)), ("--cr_weight", dict(type=float, default=0.1)),
    ("--thresh", dict(type=float, default=0.5)), ("--temp", dict(type=float, default=0.07)),
    ("--tau", dict(type=float, default=0.1)), ("--nclusters", dict(type=int, default=3)),
    ("--nheads", dict(type=int, default=1)), ("--names", dict(type=str)),
    # test
    ("--nms", dict(type=int, default=3)), ("--cutoff_z", dict(type=int, default=10)),
    ("--K", dict(type=int, default=200)), ("--not_prefetch_test", _S), ("--fix_res", _S), ("--keep_res", _S),
    ("--out_thresh", dict(type=float, default=0.25)), ("--with_score", _S), ("--pn", _S), ("--ge", _S),
    # fiber
    ("--distance_cutoff", dict(type=float, default=15)), ("--r2_cutoff", dict(type=float, default=30)),
    ("--curvature_cutoff", dict(type=float, default=0.003)), ("--distance_scale", dict(type=float, default=2)),
    # data
    ("--train_img_txt", dict(type=str, default="train_images.txt")),
    ("--train_coord_txt", dict(type=str, default="train_coords.txt")),
    ("--val_img_txt", dict(type=str)), ("--val_coord_txt", dict(type=str)),
    ("--test_img_txt", dict(type=str, default="test_images.txt")),
    ("--test_coord_txt", dict(type=str, default="test_coords.txt")),
    ("--compress", _S), ("--gauss", dict(type=float, default=0)), ("--cluster_head", _S),
    ("--out_id", dict(type=str, default="output")), ("--order", dict(type=str, default="xzy")),
    ("--dog", dict(type=list_of_floats, default=[2.5, 5])),
    # MI355X build only (not a reference flag): replay the MoCo / SimSiam step from a hipGraph (default) or launch it eagerly
    ("--hipgraph", dict(dest="hipgraph", action="store_true", default=None)),
    ("--no_hipgraph", dict(dest="hipgraph", action="store_false")),
]

# task -> heads (opts.py:285-304); lambdas see the parsed opt
_HEADS = {
    "tomo": lambda o: {"hm": o.num_classes, "proj": 16},
    "cr": lambda o: {"hm": o.num_classes, "proj": o.head_conv},
    "semi": lambda o: {"hm": o.num_classes, "proj": o.head_conv},
    "semi3d": lambda o: {"hm": o.num_classes, "proj": o.head_conv},
    "semiclass": lambda o: {"hm": o.num_classes, "proj": o.head_conv},
    "fs": lambda o: {"proj": 16},
    "tcla": lambda o: {"class": 1},
    "simsiam": lambda o: {"proj": o.head_conv, "pred": o.head_conv},
    "simsiam2d3d": lambda o: {"proj": o.head_conv, "pred": o.head_conv},
    "simsiam3d": lambda o: {"proj": o.head_conv, "pred": o.head_conv},
    "scan": lambda o: {"proj": o.head_conv, "pred": o.head_conv},
    "scan2d3d": lambda o: {"proj": o.head_conv, "pred": o.head_conv},
    "moco": lambda o: {"proj": 256, "pred": 256},
    "denoise": lambda o: {"proj": 128},
}

# task -> default dataset info (opts.py:310-322)
_DATASET_INFO = {
    "tomo": ([512, 512], 1), "cr": ([64, 64], 1), "semi": ([64, 64], 1), "semiclass": ([64, 64], 1),
    "semi3d": ([64, 64], 1), "fs": ([128, 128], 1), "simsiam": ([24, 24], 256), "scan": ([24, 24], 256),
    "denoise": ([64, 64], 256), "moco": ([32, 32], 256),
}


class opts(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.parser.add_argument("task", default="semi",
                                 help="semi | simsiam | simsiam3d | moco | ... (task names of the reference)")
        for flag, kw in _FLAGS:
            self.parser.add_argument(flag, **kw)

    def parse(self, args=""):
        opt = self.parser.parse_args() if args == "" else self.parser.parse_args(args)
        opt.gpus_str = opt.gpus
        gpus = [int(g) for g in opt.gpus.split(",")]
        opt.gpus = list(range(len(gpus))) if gpus[0] >= 0 else [-1]
        opt.lr_step = [int(i) for i in opt.lr_step.split(",")]
        opt.fix_res = not opt.keep_res
        if opt.head_conv == -1:
            if opt.task in ("simsiam", "simsiam2d3d", "simsiam3d"):
                opt.head_conv = 128
            if opt.task in ("semi", "semiclass"):
                opt.head_conv = 32
        if opt.hipgraph is None:
            # the engine-driven tasks (trains/moco_engine.py, trains/simsiam_engine.py); others run their step eagerly
            opt.hipgraph = opt.task in ("moco", "simsiam", "simsiam3d", "simsiam2d3d")
        opt.pad = 127 if "hourglass" in opt.arch else 31
        opt.num_stacks = 2 if opt.arch == "hourglass" else 1
        if opt.warm:
            opt.warmup_from = 0.01
            opt.warm_epochs = 10
            if opt.cosine:
                eta_min = opt.lr * (opt.lr_decay_rate ** 3)
                opt.warmup_to = eta_min + (opt.lr - eta_min) * (
                    1 + math.cos(math.pi * opt.warm_epochs / opt.num_epochs)) / 2
            else:
                opt.warmup_to = opt.lr
        if opt.val_intervals >= 0 and opt.val_img_txt is None and opt.val_coord_txt is None:
            opt.val_img_txt = opt.train_img_txt
            opt.val_coord_txt = opt.train_coord_txt
        if opt.trainval:
            opt.val_interval = 100000000
        if opt.debug > 0:
            opt.num_workers = 0
            opt.gpus = [opt.gpus[0]]
            opt.master_batch_size = -1
        if opt.master_batch_size == -1:
            opt.master_batch_size = opt.batch_size // len(opt.gpus)
        rest = opt.batch_size - opt.master_batch_size
        opt.chunk_sizes = [opt.master_batch_size]
        for i in range(len(opt.gpus) - 1):
            chunk = rest // (len(opt.gpus) - 1)
            if i < rest % (len(opt.gpus) - 1):
                chunk += 1
            opt.chunk_sizes.append(chunk)
        opt.root_dir = os.getcwd()
        opt.data_dir = os.path.join(opt.root_dir, "data")
        opt.exp_dir = os.path.join(opt.root_dir, "exp", opt.task)
        opt.save_dir = os.path.join(opt.exp_dir, opt.exp_id)
        opt.debug_dir = os.path.join(opt.save_dir, "debug")
        if opt.task == "scan2d3d":
            opt.simsiam_dir = os.path.join(opt.root_dir, "exp", "simsiam2d3d", opt.exp_id)
        elif opt.task == "scan":
            opt.simsiam_dir = os.path.join(opt.root_dir, "exp", "simsiam", opt.exp_id)
        opt.out_path = os.path.join(opt.save_dir, opt.out_id)
        if opt.resume and opt.load_model == "":
            model_path = opt.save_dir[:-4] if opt.save_dir.endswith("TEST") else opt.save_dir
            opt.load_model = os.path.join(model_path, "model_last.pth")
        return opt

    def update_dataset_info_and_set_heads(self, opt, dataset):
        input_h, input_w = dataset.default_resolution
        opt.num_classes = dataset.num_classes
        input_h = opt.input_res if opt.input_res > 0 else input_h
        input_w = opt.input_res if opt.input_res > 0 else input_w
        opt.input_h = opt.input_h if opt.input_h > 0 else input_h
        opt.input_w = opt.input_w if opt.input_w > 0 else input_w
        opt.output_h = opt.input_h // opt.down_ratio
        opt.output_w = opt.input_w // opt.down_ratio
        opt.input_res = max(opt.input_h, opt.input_w)
        opt.output_res = max(opt.output_h, opt.output_w)
        if opt.task not in _HEADS:
            raise AssertionError("task not defined!")
        opt.heads = _HEADS[opt.task](opt)
        return opt

    def init(self, args=""):
        opt = self.parse(args)
        res, ncls = _DATASET_INFO[opt.task]

        class _Info:
            default_resolution = res
            num_classes = ncls
            dataset = opt.task
        opt.dataset = _Info.dataset
        return self.update_dataset_info_and_set_heads(opt, _Info)
