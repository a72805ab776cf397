"""libyafaray_amd — MI355X-native path-tracing core behind libYafaRay's Interface API.

The package is a thin host-side mirror of the reference's `yafaray4::Interface`
(include/interface/interface.h:48-139) over the C ABI in include/yafaray_c_api.h.  All rendering
happens in hand-written HIP kernels (csrc/yafgpu_device.hip); there is no CPU fallback: loading
fails loudly when the HIP library is missing, rendering fails loudly without a GPU.
"""
from .interface import Interface, YafaRayError, lib_path  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["Interface", "YafaRayError", "lib_path", "scenes"]
