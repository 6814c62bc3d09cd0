from .builder import DATASETS, build_dataset, build_dataloader
from . import synthetic
