"""flash_attn_mi355 - MI355X (gfx950) fused attention operators.

Counterpart of the reference's `flash_attn_v100` package
(flash_attn_v100/__init__.py:1-18): same public names."""
__version__ = "26.06"

from .flash_attn_interface import (
    flash_attn_func,
    flash_attn_gpu,
    flash_attn_varlen_func,
    flash_attn_varlen_gpu,
    flash_attn_with_kvcache,
    flash_attn_with_kvcache_gpu,
    flash_attn_qkvpacked_func,
    flash_attn_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
    flash_attn_varlen_kvpacked_func,
)

__all__ = [
    "flash_attn_func",
    "flash_attn_gpu",
    "flash_attn_varlen_func",
    "flash_attn_varlen_gpu",
    "flash_attn_with_kvcache",
    "flash_attn_with_kvcache_gpu",
    "flash_attn_qkvpacked_func",
    "flash_attn_kvpacked_func",
    "flash_attn_varlen_qkvpacked_func",
    "flash_attn_varlen_kvpacked_func",
]
