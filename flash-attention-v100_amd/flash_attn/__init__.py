"""`flash_attn` compatibility package: lets `from flash_attn import flash_attn_func` resolve to
the MI355X implementation and reports the upstream version the reference spoofs
(flash_attn/__init__.py:1-27 of the reference: __version__ == "2.8.3")."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from flash_attn_mi355 import (
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
    __version__ as _backend_version,
)

__version__ = "2.8.3"

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
    "__version__",
]

__doc__ = f"Flash Attention for AMD Instinct MI355X v{__version__} (backend: v{_backend_version})"
