"""Mirror of cet_pick/models/model.py: model factory and checkpoint I/O
(reference models/model.py:33-70, 195-251, 283-296).

Only the architectures on the hot path are built from HIP kernels; the other 20 factory keys of the
reference are out of scope and raise with that explanation.
The checkpoint layout is the reference's: torch.save({'epoch', 'state_dict'[, 'optimizer']}).
"""
import torch

from .networks.moco_encoder_3d import get_moco_net_small_3d
from .networks.simsiam_model_2d import get_simsiam2d_net_small
from .networks.simsiam_model import get_simsiam_net_small
from .networks.simsiam_model_2d3d import get_simsiam2d3d_net_small
from .networks.unet_small import get_tomo_unet_small

_model_factory = {
    "moco3d": get_moco_net_small_3d,
    "simsiam2d": get_simsiam2d_net_small,
    "simsiam": get_simsiam_net_small,
    "simsiam3d": get_simsiam_net_small,          # models/model.py:43: the same factory as 'simsiam'
    "simsiam2d3d": get_simsiam2d3d_net_small,
    "unet": get_tomo_unet_small,
}
_REFERENCE_ARCHS = ("res", "unet", "class", "small", "ressmall", "p3d", "res3d", "unetcla", "resclass", "simsiam",
                    "simsiam3d", "moco3d", "simsiam2d", "simsiamwide3d", "simsiampyr3d", "simsiamsmall3d", "scan2d",
                    "simsiam2d3d", "scan2d3d", "denoise", "moco2d")


def create_model(arch, heads, head_conv, last_k=0, local_path=None):
    """models/model.py:65-70: arch = '<name>_<layers>'."""
    num_layers = int(arch[arch.find("_") + 1:]) if "_" in arch else 0
    arch = arch[:arch.find("_")] if "_" in arch else arch
    if arch not in _model_factory:
        if arch in _REFERENCE_ARCHS:
            raise NotImplementedError(
                "arch '%s' exists in the reference but is outside the MI355X hot path built here "
                "(DESIGN.md §7); available: %s" % (arch, sorted(_model_factory)))
        raise KeyError(arch)
    return _model_factory[arch](num_layers=num_layers, heads=heads, head_conv=head_conv, last_k=last_k,
                                local_path=local_path)


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None, model_only=False):
    """models/model.py:195-251: strips a leading 'module.', skips mis-shaped entries with a printed
    warning, fills missing ones from the model, replays the step decay on resume.  Never raises on
    key mismatch."""
    start_epoch = 0
    checkpoint = torch.load(model_path, map_location=lambda storage, loc: storage)
    print("Loaded {}, epoch {}".format(model_path, checkpoint["epoch"]))
    state_dict = {}
    for k, v in checkpoint["state_dict"].items():
        if k.startswith("module") and not k.startswith("module_list"):
            state_dict[k[7:]] = v
        else:
            state_dict[k] = v
    model_state_dict = model.state_dict()
    msg = ("If you see this, your model does not fully load the pre-trained weight. Please make sure you have "
           "correctly specified --arch xxx or set the correct --num_classes for your own dataset.")
    for k in list(state_dict):
        if k in model_state_dict:
            if state_dict[k].shape != model_state_dict[k].shape:
                print("Skip loading parameter {}, required shape{}, loaded shape{}. {}".format(
                    k, model_state_dict[k].shape, state_dict[k].shape, msg))
                state_dict[k] = model_state_dict[k]
        else:
            print("Drop parameter {}.".format(k) + msg)
    for k in model_state_dict:
        if k not in state_dict:
            print("No param {}.".format(k) + msg)
            state_dict[k] = model_state_dict[k]
    model.load_state_dict(state_dict, strict=False)

    if optimizer is not None and resume:
        if "optimizer" in checkpoint:
            optimizer.load_state_dict(checkpoint["optimizer"])
            start_epoch = checkpoint["epoch"]
        start_lr = lr
        for step in (lr_step or []):
            if start_epoch >= step:
                start_lr *= 0.1
        for param_group in optimizer.param_groups:
            param_group["lr"] = start_lr
        print("Resumed optimizer with start lr", start_lr)
    else:
        print("No optimizer parameters in checkpoint.")
    if optimizer is not None and not model_only:
        return model, optimizer, start_epoch
    return model


def save_model(path, epoch, model, optimizer=None, task=None, **kwargs):
    """models/model.py:283-296.  Tensors are written contiguous in their logical (reference) layout,
    so the file does not depend on the kernel-side weight layout or the parameter arenas."""
    if isinstance(model, torch.nn.DataParallel):
        model = model.module
    state_dict = {k: v.detach().cpu().contiguous().clone() for k, v in model.state_dict().items()}
    data = {"epoch": epoch, "state_dict": state_dict}
    if optimizer is not None:
        data["optimizer"] = optimizer.state_dict()
    if task == "scan":
        data["head"] = kwargs["head"]
    torch.save(data, path)
