"""Homes of the fused losses (``passl.loss.{moco, nt_xent, mae}``, BASELINE.json north_star / SURVEY
appendix C); the heads and backbones call into these."""
from . import mae, moco, nt_xent
from .moco import MoCoLoss, info_nce
from .nt_xent import nt_xent as nt_xent_loss
from .mae import masked_patch_loss
