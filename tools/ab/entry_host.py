"""Where an iteration of moco_main's epoch loop spends its HOST time (no device sync inside the timed loops) and what the
loader's launches cost on the device."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cet_pick_amd import moco_main
from cet_pick_amd.opts import opts
from cet_pick_amd.synthetic import make_tomo
from cet_pick_amd.utils import mrc
tmp = tempfile.mkdtemp(); os.chdir(tmp); os.makedirs("data")
vol, _ = make_tomo((128, 512, 512), seed=317)
mrc.write("data/c2.rec", vol)
open("data/train_images.txt", "w").write("image_name\trec_path\nc2\tc2.rec\n")
opt = opts().parse(["moco", "--arch", "moco3d_18", "--dataset", "simsiam3d", "--order", "zxy", "--batch_size", "64", "--lr", "0.001",
                    "--exp_id", "h", "--debug", "0", "--dog", "3,5", "--num_epochs", "1"])
opt, model, optimizer, trainer, loader, _, _, _ = moco_main.build(opt)
trainer.train(0, loader)
torch.cuda.synchronize()
def host(fn, n):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); h = time.perf_counter() - t; torch.cuda.synchronize(); w = time.perf_counter() - t
    return h / n * 1e6, w / n * 1e6
n = len(loader)
print("loader only          host %.0f us / batch, wall %.0f us / batch" % host(lambda: [None for _ in loader], n))
batches = [b for b in loader][:50]
eng = trainer.engine
print("engine.step only     host %.0f us / step,  wall %.0f us / step" % host(lambda: [eng.step(b["input"], b["input_aug"]) for b in batches], 50))
print("run_epoch            host %.0f us / iter,  wall %.0f us / iter" % host(lambda: trainer.train(1, loader), n))
os.chdir("/"); shutil.rmtree(tmp, ignore_errors=True)
