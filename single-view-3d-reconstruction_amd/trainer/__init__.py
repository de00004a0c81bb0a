from .trainer_ifnet import ImplicitRefinementTrainer, bce_with_logits_sum_mean  # noqa: F401
