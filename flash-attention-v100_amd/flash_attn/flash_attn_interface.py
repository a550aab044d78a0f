"""Same import path as upstream / the reference (flash_attn/flash_attn_interface.py)."""
from flash_attn_mi355.flash_attn_interface import (  # noqa: F401
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
