"""Host-side mirror of the reference's decode API.

Reference surface (Kotlin -> JNI -> C++):
  JxlCoder.decode / decodeSampled      jxlcoder/src/main/java/com/awxkee/jxlcoder/JxlCoder.kt:50-105
  decodeSampledImageImpl               jxlcoder/src/main/cpp/JniDecoding.cpp:45-331
  DecodeJpegXlOneShot / DecodeBasicInfo jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:36-225
Same names, argument meaning and error behaviour; Android Bitmaps become numpy arrays (host) or torch tensors
(device-resident).  All pixel work happens in libjxlamd.so on the GPU; there is NO CPU fallback: importing works
without a GPU (header parsing is host-only) but every decode raises if the HIP device or the library is missing.
"""
import ctypes as C
import enum
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("JXLAMD_LIB") or os.path.join(_HERE, "libjxlamd.so")     # JXLAMD_LIB: an experimental build of the same sources (tools/build_variant.sh)
_lib = None


class InvalidJXLException(Exception):
    """kt/InvalidJXLException — DecodeJpegXlOneShot returned false (JniDecoding.cpp:78)."""


class InvalidImageSizeException(Exception):
    """interop/JxlDecoding.h:38-52 — output >= INT32_MAX bytes."""


class UnsupportedJXLFeature(Exception):
    """Valid JPEG XL that this build does not decode on the GPU yet (never silently routed to a CPU path)."""


class PreferredColorConfig(enum.IntEnum):      # kt/PreferredColorConfig.kt, cpp/Support.h:37-44
    DEFAULT = 1
    RGBA_8888 = 2
    RGBA_F16 = 3
    RGB_565 = 4
    RGBA_1010102 = 5
    HARDWARE = 6


class ScaleMode(enum.IntEnum):                 # kt/ScaleMode.kt, cpp/SizeScaler.h:36-40
    FIT = 1
    FILL = 2
    RESIZE = 3


class Info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("xsize", "ysize", "bits_per_sample", "exponent_bits_per_sample",
                                          "num_color_channels", "num_extra_channels", "alpha_bits", "alpha_premultiplied",
                                          "orientation", "have_animation", "uses_original_profile")] + \
               [("intensity_target", C.c_float)] + \
               [(n, C.c_uint32) for n in ("have_encoded_profile", "color_space", "white_point", "primaries",
                                          "transfer_function", "rendering_intent")] + \
               [("gamma", C.c_double)] + \
               [(n, C.c_uint32) for n in ("out_bits", "prefer_encoding", "has_alpha_in_origin")] + \
               [(n, C.c_double * 2) for n in ("white_point_xy", "primaries_red_xy", "primaries_green_xy", "primaries_blue_xy")] + \
               [("icc_size", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n.endswith("_xy") else getattr(self, n)) for n, _ in self._fields_}


class ReformatInfo(C.Structure):               # jxlamd_reformat_info (include/jxl_amd.h)
    _fields_ = [("stride", C.c_uint32), ("format", C.c_uint32), ("use_floats", C.c_uint32), ("resolved_config", C.c_uint32), ("bytes", C.c_uint64)]


class RescaleInfo(C.Structure):                # jxlamd_rescale_info
    _fields_ = [(n, C.c_uint32) for n in ("scaled_w", "scaled_h", "crop_x", "crop_y", "out_w", "out_h")]


FMT_NAMES = {1: "ARGB_8888", 2: "RGBA_F16", 3: "RGB_565", 4: "RGBA_1010102"}

JXLAMD_ALLOW_16BIT, JXLAMD_OUT_DEVICE, JXLAMD_NO_SIZE_GUARD, JXLAMD_IN_DEVICE, JXLAMD_BAND_SHARED_GPU = 1, 2, 4, 8, 16
_ERR = {-1: InvalidJXLException, -2: UnsupportedJXLFeature, -3: InvalidImageSizeException, -4: RuntimeError, -5: ValueError}

SOURCES = ["kernels_lf.hip", "kernels_lf_general.hip", "kernels_lf_general_b.hip", "kernels_mod.hip", "kernels_pass.hip", "kernels_recon.hip", "kernels_filter.hip", "kernels_compose.hip", "decoder.hip", "band.hip", "post.hip", "resample.hip", "host_parse.cpp", "host_bits.cpp", "host_post.cpp", "host_icc_lut.cpp"]


def library_path():
    return _LIB_PATH


def build(force=False, verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU): one object per source, compiled in
    parallel and only when the source or a header changed, then linked into libjxlamd.so."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(_HERE, "csrc")
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".inc"))] + \
              [os.path.join(os.path.dirname(_HERE), "include", "jxl_amd.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs, objs = [], []
    for sname in SOURCES:
        src = os.path.join(csrc, sname)
        obj = os.path.join(objdir, sname + ".o")
        objs.append(obj)
        dep_time = hdr_time                       # without a dependency file from an earlier compile: any header
        try:
            deps = open(obj + ".d").read().replace("\\\n", " ").split(":", 1)[1].split()
            dep_time = max(os.path.getmtime(d) for d in deps if d.startswith(os.path.dirname(_HERE)))
        except (OSError, IndexError, ValueError):
            pass
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_time):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-MMD", "-MF", obj + ".d",
                         "-c", src, "-o", obj])
    if not jobs and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(o) for o in objs):
        build_compat(verbose=verbose)
        return _LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", _LIB_PATH] + objs)
    build_compat(verbose=verbose)
    return _LIB_PATH


def compat_dir():
    """Directory of the libjxl-named libraries of the secondary boundary (include/jxl_amd_libjxl.h): libjxl.so, libjxl_threads.so."""
    return os.path.join(_HERE, "compat")


def build_compat(verbose=False):
    """compat/libjxl.so + compat/libjxl_threads.so: the libjxl C-API subset the reference's decode path calls, over libjxlamd.so
    (host-only glue, plain g++)."""
    src = os.path.join(_HERE, "csrc", "libjxl_abi.cpp")
    os.makedirs(compat_dir(), exist_ok=True)
    # (libjxlamd.so is linked dynamically: the compat libraries only follow their own source and the two headers — a fresh copy of the tree, whose file times
    # are in no particular order, must not relink them under the feet of a process that is loading them)
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(os.path.dirname(_HERE), "include", "jxl_amd_libjxl.h")),
                 os.path.getmtime(os.path.join(os.path.dirname(_HERE), "include", "jxl_amd.h")))
    for name, macro, extra in (("libjxl.so", "-DJXLC_ONLY_DECODER", [_LIB_PATH, "-Wl,-rpath,$ORIGIN/.."]), ("libjxl_threads.so", "-DJXLC_ONLY_THREADS", [])):
        out = os.path.join(compat_dir(), name)
        if os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        tmp = out + ".%d.tmp" % os.getpid()
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", macro, "-Wl,-soname," + name, "-o", tmp, src] + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp, out)                      # atomic: another process may be loading the library right now
    return compat_dir()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run jxl_coder_amd.build() (python -c 'import __graft_entry__ as g; g.build()'). "
                               "There is no CPU fallback for the decode path.")
        try:
            # torch bundles its own libamdhip64 next to the system one libjxlamd.so links: when both live in one process the runtime
            # that initialises FIRST must be torch's (the other order leaves torch with "No HIP GPUs are available")
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        L.jxlamd_decoder_create.restype = C.c_void_p
        L.jxlamd_decoder_create.argtypes = [C.c_int]
        L.jxlamd_decoder_destroy.argtypes = [C.c_void_p]
        L.jxlamd_last_error.restype = C.c_char_p
        L.jxlamd_last_error.argtypes = [C.c_void_p]
        L.jxlamd_basic_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Info)]
        L.jxlamd_output_size.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t)]
        L.jxlamd_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(Info)]
        L.jxlamd_decoder_set_writer_post.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.jxlamd_decoder_set_epf_reciprocal.argtypes = [C.c_void_p, C.c_int]
        L.jxlamd_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(Info)]
        L.jxlamd_anim_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.jxlamd_decode_resident.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(Info)]
        L.jxlamd_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float * 5)]
        L.jxlamd_decode_batch_resident.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.jxlamd_reformat_query.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ReformatInfo)]
        L.jxlamd_reformat.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_size_t, C.POINTER(ReformatInfo)]
        L.jxlamd_post_fused.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(ReformatInfo)]
        L.jxlamd_color_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_float]
        L.jxlamd_get_icc.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.jxlamd_icc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]
        L.jxlamd_rescale_query.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(RescaleInfo)]
        L.jxlamd_rescale.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t, C.POINTER(RescaleInfo)]
        L.jxlamd_decoder_share_pools.argtypes = [C.c_void_p, C.c_void_p]
        L.jxlamd_band_begin.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(Info)]
        L.jxlamd_band_halo_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.jxlamd_band_export.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.jxlamd_band_import.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.jxlamd_band_reconstruct.argtypes = [C.c_void_p]
        L.jxlamd_band_finish.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _raise(rc, dec):
    msg = lib().jxlamd_last_error(dec).decode(errors="replace")
    raise _ERR.get(rc, RuntimeError)(msg)


class JxlDecoder:
    """One decoder context = one HIP device + stream + reusable HBM work buffers (the reference creates a libjxl
    decoder + thread-pool runner per call, interop/JxlDecoding.cpp:46-48; here the context is reused)."""

    def __init__(self, device=0):
        self._h = lib().jxlamd_decoder_create(device)
        if not self._h:
            raise RuntimeError(lib().jxlamd_last_error(None).decode())
        self.device = device

    def share_pools(self, owner):
        """run the HF phase of this context's batches in `owner`'s coefficient / pixel-plane pools, taking turns (include/jxl_amd.h: jxlamd_decoder_share_pools)"""
        rc = lib().jxlamd_decoder_share_pools(owner._h, self._h)
        if rc:
            _raise(rc, None)

    def close(self):
        if self._h:
            lib().jxlamd_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_one_shot(self, data: bytes, allowed_floats=True, size_guard=True):
        """DecodeJpegXlOneShot (interop/JxlDecoding.cpp:36-176): -> (pixels [h,w,4] u8|u16, info dict)."""
        flags = (JXLAMD_ALLOW_16BIT if allowed_floats else 0) | (0 if size_guard else JXLAMD_NO_SIZE_GUARD)
        n = C.c_size_t()
        rc = lib().jxlamd_output_size(data, len(data), flags, C.byref(n))
        if rc:
            _raise(rc, None)
        out = np.empty(n.value, np.uint8)
        info = Info()
        rc = lib().jxlamd_decode(self._h, data, len(data), flags, out.ctypes.data, out.nbytes, C.byref(info))
        if rc:
            _raise(rc, self._h)
        dt = np.uint16 if info.out_bits == 16 else np.uint8
        return out.view(dt).reshape(info.ysize, info.xsize, 4), info.as_dict()

    def decode_to_device(self, data: bytes, out_ptr: int, out_capacity: int, data_dev_ptr: int = 0, allowed_floats=True,
                         size_guard=True):
        """HBM-resident decode: output (and optionally the compressed bytes) stay in device memory."""
        flags = (JXLAMD_ALLOW_16BIT if allowed_floats else 0) | JXLAMD_OUT_DEVICE | (0 if size_guard else JXLAMD_NO_SIZE_GUARD)
        info = Info()
        if data_dev_ptr:
            rc = lib().jxlamd_decode_resident(self._h, data, len(data), data_dev_ptr, flags, out_ptr, out_capacity, C.byref(info))
        else:
            rc = lib().jxlamd_decode(self._h, data, len(data), flags, out_ptr, out_capacity, C.byref(info))
        if rc:
            _raise(rc, self._h)
        return info.as_dict()

    def set_epf_reciprocal(self, x86_reference_build: bool):
        """jxlamd_decoder_set_epf_reciprocal: False (default) the EPF normalises with the exact quotient; True with the reference x86 build's 12-bit rcpps
        (a table of the golden host's results) — the pixels interop/JxlDecoding.cpp:75 returned there, bit for bit where the rest of the path allows."""
        rc = lib().jxlamd_decoder_set_epf_reciprocal(self._h, int(bool(x86_reference_build)))
        if rc:
            _raise(rc, self._h)

    def set_writer_post(self, enabled: bool, config=PreferredColorConfig.DEFAULT, api_level=34):
        """A10 + A11 with the decode (jxlamd_decoder_set_writer_post): decodes of this context deliver the Bitmap format of
        reformat_query(w, h, 16-bit?, config, has alpha, api_level) — from the writer itself where the frame's last filter stage allows it."""
        rc = lib().jxlamd_decoder_set_writer_post(self._h, int(bool(enabled)), int(config), int(api_level))
        if rc:
            _raise(rc, self._h)

    def decode_frame_to_device(self, data: bytes, frame: int, out_ptr: int, out_capacity: int, allowed_floats=False):
        """Coalesced frame `frame` of an animation into device memory (jxlamd_decode_frame; the reference's JxlAnimatedDecoder::getFrame)."""
        flags = (JXLAMD_ALLOW_16BIT if allowed_floats else 0) | JXLAMD_OUT_DEVICE
        info = Info()
        rc = lib().jxlamd_decode_frame(self._h, data, len(data), int(frame), flags, out_ptr, out_capacity, C.byref(info))
        if rc:
            _raise(rc, self._h)
        return info.as_dict()

    def decode_frame(self, data: bytes, frame: int):
        """-> (h, w, 4) u8: coalesced frame `frame` (always 8-bit, as the reference's animated decoder: JxlAnimatedDecoder.cpp:60)."""
        import numpy as np
        w, h = JxlCoder.getSize(data)
        out = np.empty((h, w, 4), np.uint8)
        info = Info()
        rc = lib().jxlamd_decode_frame(self._h, data, len(data), int(frame), 0, out.ctypes.data, out.nbytes, C.byref(info))
        if rc:
            _raise(rc, self._h)
        return out, info.as_dict()

    def color_matrix_device(self, ptr: int, w: int, h: int, is_u16: bool, depth: int, primaries: int, transfer_function: int,
                            intensity_target: float, xy8=None):
        """A10 on a device buffer, in place (jxlamd_color_matrix; cpp/colorspaces/ColorMatrix.cpp:35-219)."""
        xy = (C.c_double * 8)(*xy8) if xy8 is not None else None
        rc = lib().jxlamd_color_matrix(self._h, ptr, w, h, int(is_u16), depth, primaries, transfer_function, xy, float(intensity_target))
        if rc:
            _raise(rc, self._h)

    def reformat_query(self, w, h, is_u16, config, has_alpha_in_origin, api_level):
        ri = ReformatInfo()
        rc = lib().jxlamd_reformat_query(w, h, int(is_u16), int(config), int(has_alpha_in_origin), int(api_level), C.byref(ri))
        if rc:
            _raise(rc, None)
        return ri

    def reformat_device(self, src_ptr: int, w: int, h: int, is_u16: bool, depth: int, config, alpha_premultiplied: bool,
                        has_alpha_in_origin: bool, api_level: int, dst_ptr: int, dst_capacity: int):
        """A11 on device buffers (jxlamd_reformat; cpp/ReformatBitmap.cpp:46-263).  Premultiplies src in place when the reference does."""
        ri = ReformatInfo()
        rc = lib().jxlamd_reformat(self._h, src_ptr, w, h, int(is_u16), depth, int(config), int(alpha_premultiplied), int(has_alpha_in_origin),
                                   int(api_level), dst_ptr, dst_capacity, C.byref(ri))
        if rc:
            _raise(rc, self._h)
        return ri

    def post_fused_device(self, src_ptr: int, w: int, h: int, is_u16: bool, depth: int, apply_color_matrix: bool, primaries: int, transfer_function: int,
                          intensity_target: float, config, alpha_premultiplied: bool, has_alpha_in_origin: bool, api_level: int, dst_ptr: int, dst_capacity: int,
                          xy8=None):
        """A10 + A11 in one pass over a device buffer (jxlamd_post_fused): the values color_matrix_device + reformat_device give, src untouched."""
        ri = ReformatInfo()
        xy = (C.c_double * 8)(*xy8) if xy8 is not None else None
        rc = lib().jxlamd_post_fused(self._h, src_ptr, w, h, int(is_u16), depth, int(apply_color_matrix), primaries, transfer_function, xy, float(intensity_target),
                                     int(config), int(alpha_premultiplied), int(has_alpha_in_origin), int(api_level), dst_ptr, dst_capacity, C.byref(ri))
        if rc:
            _raise(rc, self._h)
        return ri

    def decode_batch_to_device(self, datas, out_ptrs, out_capacities, data_dev_ptrs=None, allowed_floats=True):
        """jxlamd_decode_batch_resident: n independent frames; the entropy stages of the whole batch share one launch each."""
        n = len(datas)
        flags = (JXLAMD_ALLOW_16BIT if allowed_floats else 0) | JXLAMD_OUT_DEVICE
        bufs = [C.c_char_p(d) for d in datas]
        a_jxl = (C.c_char_p * n)(*bufs)
        a_sz = (C.c_size_t * n)(*[len(d) for d in datas])
        a_dev = (C.c_void_p * n)(*[C.c_void_p(p) for p in (data_dev_ptrs or [0] * n)])
        a_out = (C.c_void_p * n)(*[C.c_void_p(p) for p in out_ptrs])
        a_cap = (C.c_size_t * n)(*out_capacities)
        infos = (Info * n)()
        rc = lib().jxlamd_decode_batch_resident(self._h, n, a_jxl, a_sz, a_dev, flags, a_out, a_cap, infos)
        if rc:
            _raise(rc, self._h)
        return [i.as_dict() for i in infos]

    def icc_transform_device(self, ptr: int, w: int, h: int, is_u16: bool, icc: bytes):
        """A8 on a device buffer, in place (jxlamd_icc_transform; cpp/colorspaces/colorspace.cpp:38-86)."""
        rc = lib().jxlamd_icc_transform(self._h, ptr, w, h, int(is_u16), icc, len(icc))
        if rc:
            _raise(rc, self._h)

    def rescale_query(self, w, h, new_w, new_h, scale_mode):
        ri = RescaleInfo()
        rc = lib().jxlamd_rescale_query(w, h, int(new_w), int(new_h), int(scale_mode), C.byref(ri))
        if rc:
            _raise(rc, None)
        return ri

    def rescale_device(self, src_ptr, w, h, is_u16, depth, new_w, new_h, scale_mode, sampler, premultiply_alpha, dst_ptr, dst_capacity):
        """A9 on device buffers (jxlamd_rescale; cpp/SizeScaler.cpp:38-144)."""
        ri = RescaleInfo()
        rc = lib().jxlamd_rescale(self._h, src_ptr, w, h, int(is_u16), depth, int(new_w), int(new_h), int(scale_mode), int(sampler), int(premultiply_alpha),
                                  dst_ptr, dst_capacity, C.byref(ri))
        if rc:
            _raise(rc, self._h)
        return ri

    # ---- band-sharded decode of one frame (include/jxl_amd.h "Band-sharded decode"; jxl_coder_amd/shard.py drives it)
    def band_begin(self, data: bytes, group_row0: int, group_row1: int, out_ptr: int, out_capacity: int, allowed_floats=True, shared_gpu=False):
        flags = (JXLAMD_ALLOW_16BIT if allowed_floats else 0) | JXLAMD_OUT_DEVICE | (JXLAMD_BAND_SHARED_GPU if shared_gpu else 0)
        info = Info()
        rc = lib().jxlamd_band_begin(self._h, data, len(data), flags, group_row0, group_row1, out_ptr, out_capacity, C.byref(info))
        if rc:
            _raise(rc, self._h)
        return info.as_dict()

    def band_halo_bytes(self, kind: int) -> int:
        n = C.c_size_t()
        rc = lib().jxlamd_band_halo_bytes(self._h, kind, C.byref(n))
        if rc:
            _raise(rc, self._h)
        return n.value

    def band_export(self, kind: int, side: int, ptr: int, capacity: int):
        rc = lib().jxlamd_band_export(self._h, kind, side, ptr, capacity)
        if rc:
            _raise(rc, self._h)

    def band_import(self, kind: int, side: int, ptr: int, size: int):
        rc = lib().jxlamd_band_import(self._h, kind, side, ptr, size)
        if rc:
            _raise(rc, self._h)

    def band_reconstruct(self):
        rc = lib().jxlamd_band_reconstruct(self._h)
        if rc:
            _raise(rc, self._h)

    def band_finish(self):
        rc = lib().jxlamd_band_finish(self._h)
        if rc:
            _raise(rc, self._h)

    def last_timing(self):
        t = (C.c_float * 5)()
        lib().jxlamd_last_timing(self._h, C.byref(t))
        return dict(zip(("lf_groups_ms", "pass_groups_ms", "recon_ms", "filters_write_ms", "device_total_ms"), list(t)))


def _check_preconditions(cfg, scale_mode, sampler=6, os_version=34):
    """checkDecodePreconditions (cpp/Support.cpp:35-92), same order and messages: enum 0 / out-of-range config, the API-level gates of
    RGBA_1010102 (33+), RGBA_F16 (26+), HARDWARE (29+), scale mode, sampler.  The reference throws java.lang.Exception with these
    strings (throwException); the mirror raises ValueError."""
    if int(cfg) < 1 or int(cfg) > 6:
        raise ValueError(f"Invalid Color Config: {int(cfg)} was passed")
    if int(cfg) == PreferredColorConfig.RGBA_1010102 and os_version < 33:
        raise ValueError(f"Color Config RGBA_1010102 supported only 33+ OS version but current is: {os_version}")
    if int(cfg) == PreferredColorConfig.RGBA_F16 and os_version < 26:
        raise ValueError(f"Color Config RGBA_1010102 supported only 26+ OS version but current is: {os_version}")     # the reference's own wording (Support.cpp:57-61)
    if int(cfg) == PreferredColorConfig.HARDWARE and os_version < 29:
        raise ValueError(f"Color Config HARDWARE supported only 29+ OS version but current is: {os_version}")
    if int(scale_mode) < 1 or int(scale_mode) > 3:
        raise ValueError("Invalid Scale Mode was passed")
    if int(sampler) < 1 or int(sampler) > 10:           # XSampler, cpp/SizeScaler.h
        raise ValueError(f"Invalid Sampler: {int(sampler)} was passed")


class JxlCoder:
    """object JxlCoder (kt/JxlCoder.kt:39-268), decode half."""
    _default = None

    @classmethod
    def _decoder(cls):
        if cls._default is None:
            cls._default = JxlDecoder(int(os.environ.get("LOCAL_RANK", "0")))
        return cls._default

    @staticmethod
    def isJXL(data: bytes) -> bool:              # kt/JxlCoder.kt:244-267
        return data[:2] == b"\xff\x0a" or data[:12] == bytes([0, 0, 0, 0xC, 0x4A, 0x58, 0x4C, 0x20, 0xD, 0xA, 0x87, 0xA])

    @staticmethod
    def getSize(data: bytes):                    # kt/JxlCoder.kt:191 -> DecodeBasicInfo
        info = Info()
        rc = lib().jxlamd_basic_info(data, len(data), C.byref(info))
        if rc:
            raise InvalidJXLException(lib().jxlamd_last_error(None).decode())
        return info.xsize, info.ysize

    @classmethod
    def decode(cls, data: bytes, preferredColorConfig=PreferredColorConfig.DEFAULT, scaleMode=ScaleMode.FIT):
        """kt/JxlCoder.kt:50-63: decodeSampledImpl(bytes, -1, -1, cfg, mode, CATMULL_ROM)."""
        return cls.decodeSampled(data, -1, -1, preferredColorConfig, scaleMode)

    @classmethod
    def decodeSampled(cls, data: bytes, width: int, height: int, preferredColorConfig=PreferredColorConfig.DEFAULT,
                      scaleMode=ScaleMode.FIT, jxlResizeFilter=6):
        _check_preconditions(preferredColorConfig, scaleMode, jxlResizeFilter, cls.api_level)
        use_sampler = (width > 0 or height > 0) and (width != 0 and height != 0)      # JniDecoding.cpp:116-117
        return cls._decode_pipeline(data, preferredColorConfig, None, (width, height, scaleMode, jxlResizeFilter) if use_sampler else None).pixels_view()

    # Android API level the mirror emulates: >= 34 tags the Bitmap with a ColorSpace and leaves the pixels alone, below 34 the
    # reference converts to sRGB / Rec.709 with applyColorMatrix (cpp/JniDecoding.cpp:131-228).  ReformatColorConfig's
    # DEFAULT also depends on it (cpp/ReformatBitmap.cpp:52-63).
    api_level = 34

    @classmethod
    def decodeBitmap(cls, data: bytes, preferredColorConfig=PreferredColorConfig.DEFAULT, api_level=None):
        """What decodeSampledImageImpl hands to Bitmap creation (cpp/JniDecoding.cpp:45-331): rows with the reference's stride,
        config name, useFloats — decode (A5) -> colour matrix (A10, API < 34 only) -> reformat (A11), all on the GPU."""
        _check_preconditions(preferredColorConfig, ScaleMode.FIT)
        return cls._decode_pipeline(data, preferredColorConfig, api_level)

    @classmethod
    def _decode_pipeline(cls, data, config, api_level=None, sampling=None, fused_post=True):
        import numpy as np
        import torch
        api = cls.api_level if api_level is None else int(api_level)
        dec = cls._decoder()
        info = Info()
        rc = lib().jxlamd_basic_info(data, len(data), C.byref(info))
        if rc:
            raise InvalidJXLException(lib().jxlamd_last_error(None).decode())
        w, h, is16 = info.xsize, info.ysize, info.out_bits == 16
        dev = f"cuda:{dec.device}"
        raw = torch.empty(w * h * 4 * (2 if is16 else 1), dtype=torch.uint8, device=dev)
        meta = dec.decode_to_device(data, raw.data_ptr(), raw.numel(), allowed_floats=True)
        depth = 16 if is16 else 8                                  # bitDepth as DecodeJpegXlOneShot reports it (JxlDecoding.cpp:92-101)
        if meta["icc_size"] and not meta["prefer_encoding"]:       # A8 convertUseDefinedColorSpace (JniDecoding.cpp:103-114): the ICC vector is non-empty
            buf = (C.c_uint8 * meta["icc_size"])(); n = C.c_size_t()
            if lib().jxlamd_get_icc(data, len(data), buf, meta["icc_size"], C.byref(n)) == 0 and n.value:
                dec.icc_transform_device(raw.data_ptr(), w, h, is16, bytes(buf[:n.value]))
        if sampling is not None:                                   # A9 RescaleImage (JniDecoding.cpp:116-136), before the colour matrix
            sw, sh, mode, sampler = sampling
            q = dec.rescale_query(w, h, sw, sh, mode)
            scaled = torch.empty(q.out_w * q.out_h * 4 * (2 if is16 else 1), dtype=torch.uint8, device=dev)
            dec.rescale_device(raw.data_ptr(), w, h, is16, depth, sw, sh, mode, sampler, bool(meta["has_alpha_in_origin"]), scaled.data_ptr(), scaled.numel())
            raw, w, h = scaled, q.out_w, q.out_h
        tf = meta["transfer_function"]
        matrix = bool(meta["prefer_encoding"] and tf in (16, 18, 17, 1, 65535, 13) and meta["color_space"] == 0 and api < 34)   # JniDecoding.cpp:131-137
        ri = dec.reformat_query(w, h, is16, config, meta["has_alpha_in_origin"], api)
        dst = torch.empty(int(ri.bytes), dtype=torch.uint8, device=dev)
        if fused_post:    # A10 + A11 in one pass; fused_post=False: the reference's two stages as two launches (same pixels)
            ri = dec.post_fused_device(raw.data_ptr(), w, h, is16, depth, matrix, meta["primaries"], tf, meta["intensity_target"], config,
                                       bool(meta["alpha_premultiplied"]), bool(meta["has_alpha_in_origin"]), api, dst.data_ptr(), dst.numel())
        else:
            if matrix:
                dec.color_matrix_device(raw.data_ptr(), w, h, is16, depth, meta["primaries"], tf, meta["intensity_target"])
            ri = dec.reformat_device(raw.data_ptr(), w, h, is16, depth, config, bool(meta["alpha_premultiplied"]), bool(meta["has_alpha_in_origin"]),
                                     api, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        rows = dst.cpu().numpy().reshape(h, ri.stride)
        name = "HARDWARE" if ri.resolved_config == int(PreferredColorConfig.HARDWARE) else FMT_NAMES[ri.format]
        return Bitmap(rows, w, h, int(ri.stride), name, bool(ri.use_floats), meta)


def anim_info(data: bytes):
    """-> (durations_ms, loops): the frame list the reference's JxlAnimatedDecoder constructor collects (interop/JxlAnimatedDecoder.hpp:68-185)."""
    n, loops = C.c_int32(), C.c_int32()
    rc = lib().jxlamd_anim_info(data, len(data), None, 0, C.byref(n), C.byref(loops))
    if rc:
        raise InvalidJXLException(lib().jxlamd_last_error(None).decode())
    d = (C.c_int32 * max(1, n.value))()
    lib().jxlamd_anim_info(data, len(data), d, n.value, C.byref(n), C.byref(loops))
    return [int(d[i]) for i in range(n.value)], int(loops.value)


class JxlAnimatedImage:
    """class JxlAnimatedImage (kt/JxlAnimatedImage.kt:41-199) over jxlamd_anim_info / jxlamd_decode_frame: numberOfFrames, loopsCount, getFrameDuration,
    getFrame(frame, scaleWidth, scaleHeight), getWidth / getHeight, close.  getFrame follows getFrameImpl (cpp/JxlAnimatedDecoderCoordinator.cpp:161-412):
    always 8-bit; colour matrix / tone map (API < 34) -> ICC -> rescale -> reformat — the animated path's order, which differs from the still path's
     — every stage on the HBM-resident frame."""

    def __init__(self, data: bytes, preferredColorConfig=PreferredColorConfig.DEFAULT, scaleMode=ScaleMode.FIT, jxlResizeFilter=3, api_level=None):
        _check_preconditions(preferredColorConfig, scaleMode, jxlResizeFilter, JxlCoder.api_level if api_level is None else api_level)
        if not JxlCoder.isJXL(data):
            raise InvalidJXLException("Not an JXL image")                          # JxlAnimatedDecoder.hpp:71-74
        self._data, self._config, self.scaleMode, self._sampler = bytes(data), preferredColorConfig, scaleMode, jxlResizeFilter
        self._api = JxlCoder.api_level if api_level is None else int(api_level)
        self._durations, self._loops = anim_info(self._data)
        self._w, self._h = JxlCoder.getSize(self._data)
        self._open = True

    def _assert_open(self):
        if not self._open:
            raise RuntimeError("Animated image is already closed, call of it is impossible")     # kt/JxlAnimatedImage.kt:159-165

    @property
    def numberOfFrames(self):
        self._assert_open(); return len(self._durations)

    @property
    def loopsCount(self):
        self._assert_open(); return self._loops

    def getFrameDuration(self, frame: int) -> int:
        self._assert_open()
        if frame < 0 or frame >= len(self._durations):
            raise ValueError("Requested frame index more than frames in the container")          # JxlAnimatedDecoder.cpp:35-38
        return self._durations[frame]

    def getWidth(self):
        self._assert_open(); return self._w

    def getHeight(self):
        self._assert_open(); return self._h

    def getFrame(self, frame: int, scaleWidth: int = 0, scaleHeight: int = 0):
        import torch
        self._assert_open()
        if frame < 0:
            raise ValueError("Frame position must be positive")                                    # JxlAnimatedDecoder.cpp:30-33
        if frame >= len(self._durations):
            raise ValueError("Requested frame index more than frames in the container")
        dec = JxlCoder._decoder()
        dev = f"cuda:{dec.device}"
        w, h = self._w, self._h
        raw = torch.empty(w * h * 4, dtype=torch.uint8, device=dev)
        meta = dec.decode_frame_to_device(self._data, frame, raw.data_ptr(), raw.numel(), allowed_floats=False)
        tf = meta["transfer_function"]
        matrix = bool(meta["prefer_encoding"] and tf in (16, 18, 17, 1, 65535, 13) and meta["color_space"] == 0 and self._api < 34)      # Coordinator.cpp:184-190
        if matrix:
            dec.color_matrix_device(raw.data_ptr(), w, h, False, 8, meta["primaries"], tf, meta["intensity_target"])
        if meta["icc_size"] and not meta["prefer_encoding"]:                                       # Coordinator.cpp:267-274
            buf = (C.c_uint8 * meta["icc_size"])(); n = C.c_size_t()
            if lib().jxlamd_get_icc(self._data, len(self._data), buf, meta["icc_size"], C.byref(n)) == 0 and n.value:
                dec.icc_transform_device(raw.data_ptr(), w, h, False, bytes(buf[:n.value]))
        if (scaleWidth > 0 or scaleHeight > 0) and scaleWidth != 0 and scaleHeight != 0 and scaleWidth > 0 and scaleHeight > 0:      # Coordinator.cpp:275-294
            q = dec.rescale_query(w, h, scaleWidth, scaleHeight, self.scaleMode)
            scaled = torch.empty(q.out_w * q.out_h * 4, dtype=torch.uint8, device=dev)
            dec.rescale_device(raw.data_ptr(), w, h, False, 8, scaleWidth, scaleHeight, self.scaleMode, self._sampler, bool(meta["has_alpha_in_origin"]), scaled.data_ptr(), scaled.numel())
            raw, w, h = scaled, q.out_w, q.out_h
        ri = dec.reformat_query(w, h, False, self._config, meta["has_alpha_in_origin"], self._api)
        dst = torch.empty(int(ri.bytes), dtype=torch.uint8, device=dev)
        ri = dec.reformat_device(raw.data_ptr(), w, h, False, 8, self._config, bool(meta["alpha_premultiplied"]), bool(meta["has_alpha_in_origin"]), self._api, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        rows = dst.cpu().numpy().reshape(h, ri.stride)
        name = "HARDWARE" if ri.resolved_config == int(PreferredColorConfig.HARDWARE) else FMT_NAMES[ri.format]
        return Bitmap(rows, w, h, int(ri.stride), name, bool(ri.use_floats), meta)

    def close(self):
        self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Bitmap:
    """The buffer the reference copies into an android.graphics.Bitmap (cpp/JniDecoding.cpp:266-326): `rows` is (height, stride) u8."""

    def __init__(self, rows, width, height, stride, config, use_floats, info):
        self.rows, self.width, self.height, self.stride, self.config, self.use_floats, self.info = rows, width, height, stride, config, use_floats, info

    def pixels_view(self):
        """(h, w, 4) u8 for ARGB_8888, (h, w, 4) u16 bit patterns (half floats) for RGBA_F16, (h, w) u16 for RGB_565, (h, w) u32 for RGBA_1010102."""
        import numpy as np
        w, h = self.width, self.height
        if self.config in ("ARGB_8888",) or (self.config == "HARDWARE" and not self.use_floats):
            return self.rows[:, :w * 4].reshape(h, w, 4)
        if self.config == "RGBA_F16" or self.config == "HARDWARE":
            return np.ascontiguousarray(self.rows[:, :w * 8]).view(np.uint16).reshape(h, w, 4)
        if self.config == "RGB_565":
            return np.ascontiguousarray(self.rows[:, :w * 2]).view(np.uint16).reshape(h, w)
        return np.ascontiguousarray(self.rows[:, :w * 4]).view(np.uint32).reshape(h, w)
