"""jxl_coder_amd — MI355X-native JPEG XL decode path behind awxkee/jxl-coder's decode surface.

Host-side mirror of the reference's Kotlin/JNI API (jxlcoder/src/main/java/com/awxkee/jxlcoder/JxlCoder.kt:50-105)
over the C-ABI in include/jxl_amd.h (libjxlamd.so = host parser + hand-written HIP kernels for gfx950)."""
from .api import (Bitmap, InvalidJXLException, InvalidImageSizeException, JxlAnimatedImage, JxlCoder, JxlDecoder, PreferredColorConfig,
                  ScaleMode, UnsupportedJXLFeature, build, library_path)

__all__ = ["Bitmap", "JxlAnimatedImage", "JxlCoder", "JxlDecoder", "PreferredColorConfig", "ScaleMode", "InvalidJXLException",
           "InvalidImageSizeException", "UnsupportedJXLFeature", "build", "library_path"]
