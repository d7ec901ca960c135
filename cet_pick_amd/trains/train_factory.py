"""Mirror of cet_pick/trains/train_factory.py:16-30 for the tasks on the hot path."""
from .tomo_moco_trainer import TomoMocoTrainer

train_factory = {
    "moco": TomoMocoTrainer,
}
