"""MI355X-native fused cosine-similarity attention (drop-in for the hot path of
lucidrains/flash-cosine-sim-attention).  Same public names as the reference package
(flash_cosine_sim_attention/__init__.py:1)."""
from .ops import (flash_cosine_sim_attention, plain_cosine_sim_attention, l2norm_tensors,
                  FlashCosineSimAttention)
from .ext import debug

__version__ = '0.3.0'
__all__ = ['flash_cosine_sim_attention', 'plain_cosine_sim_attention', 'l2norm_tensors', 'debug',
           'FlashCosineSimAttention']
