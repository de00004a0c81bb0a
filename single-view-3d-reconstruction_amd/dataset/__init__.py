from .implicit_dataset import DeviceSampleLoader, ImplicitDataset  # noqa: F401
