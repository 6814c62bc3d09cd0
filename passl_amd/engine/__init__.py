from .trainer import Trainer, IterLoader
