from .trainer_ifnet import ImplicitRefinementTrainer, bce_with_logits_sum_mean  # noqa: F401
from .trainer_scene_net import SceneNetTrainer, default_hparams  # noqa: F401
