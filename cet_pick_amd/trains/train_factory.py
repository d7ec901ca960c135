"""Mirror of cet_pick/trains/train_factory.py:16-30 for the tasks on the hot path."""
from .tomo_moco_trainer import TomoMocoTrainer
from .tomo_simsiam_trainer import TomoSimSiamTrainer
from .tomo_cr_semi_trainer import TomoCRSemiTrainer
from .tomo_moco_small_trainer import MoCoTrainer

train_factory = {
    "moco": TomoMocoTrainer,
    "simsiam": TomoSimSiamTrainer,
    "simsiam3d": TomoSimSiamTrainer,
    "simsiam2d3d": TomoSimSiamTrainer,
    "semi": TomoCRSemiTrainer,
    "semi3d": TomoCRSemiTrainer,
}
# (MoCoTrainer, the symmetric variant, is constructed directly by its script in the reference and here alike)
