"""ctypes binding of include/sbbseg.h -- the thin layer between the Python host and the HIP library.

There is deliberately no fallback: if libsbbseg.so is missing or a call fails, a RuntimeError is
raised (the reference's callers expect ordinary Python exceptions, main.py:2061-2157)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsbbseg.so")

PREC_BF16, PREC_F32, PREC_F16, PREC_F16X3 = 0, 1, 2, 3
PRECISIONS = {"bf16": PREC_BF16, "f32": PREC_F32, "f16": PREC_F16, "f16x3": PREC_F16X3}
INPUT_C8, INPUT_PAIRS = 0, 1

EXPORTS = [
    "sbbseg_last_error", "sbbseg_abi_version", "sbbseg_device_count", "sbbseg_create", "sbbseg_destroy",
    "sbbseg_model_load", "sbbseg_model_load_file", "sbbseg_debug_plan_summary",
    "sbbseg_set_stream", "sbbseg_set_lanes", "sbbseg_set_label_channels", "sbbseg_set_dedupe", "sbbseg_set_ksplit", "sbbseg_synchronize", "sbbseg_set_input", "sbbseg_input_form", "sbbseg_add_tensor",
    "sbbseg_add_conv", "sbbseg_add_maxpool", "sbbseg_add_tail", "sbbseg_add_head", "sbbseg_finalize", "sbbseg_model_info",
    "sbbseg_num_ops", "sbbseg_op_info", "sbbseg_op_issued_flops", "sbbseg_device_bytes", "sbbseg_predict", "sbbseg_segment_page",
    "sbbseg_segment_page_dev", "sbbseg_segment_pages_dev", "sbbseg_segment_page_scaled", "sbbseg_segment_page_otsu", "sbbseg_segment_crop", "sbbseg_segment_crop_dev", "sbbseg_debug_largest_contour", "sbbseg_debug_counter", "sbbseg_comm_unique_id", "sbbseg_comm_init", "sbbseg_comm_info", "sbbseg_comm_destroy", "sbbseg_allgather_labels_dev", "sbbseg_otsu_dev",
    "sbbseg_segment_tile_range_bin_dev", "sbbseg_segment_whole", "sbbseg_segment_whole_scaled", "sbbseg_tile_grid", "sbbseg_nearest_map", "sbbseg_segment_tiles_dev",
    "sbbseg_segment_tile_range_dev", "sbbseg_stitch_dev", "sbbseg_debug_ingest", "sbbseg_debug_read_tensor",
    "sbbseg_debug_set_conv_variant", "sbbseg_debug_inject_alloc_failure",
    "sbbseg_morph_dev", "sbbseg_morph", "sbbseg_page_box_dev", "sbbseg_extract_page_box", "sbbseg_extract_page_box_dev",
    "sbbseg_deskew_side", "sbbseg_rotation_matrix", "sbbseg_deskew_profiles_dev", "sbbseg_deskew_profiles",
    "sbbseg_segment_pages",
    "sbbseg_profile_enable", "sbbseg_profile_reset", "sbbseg_profile_get",
    "sbbseg_debug_largest_contour_area2", "sbbseg_text_regions_present_dev", "sbbseg_run_page", "sbbseg_device_alloc", "sbbseg_device_free", "sbbseg_upload", "sbbseg_download", "sbbseg_download_labels",
    "sbbseg_set_owned_regions", "sbbseg_owned_region_info", "sbbseg_op_executed", "sbbseg_debug_owned_range", "sbbseg_debug_region_rows",
    "sbbseg_debug_poison_activations",
]


class ConvSrc(C.Structure):
    _fields_ = [("tensor", C.c_int32), ("channels", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("stride_y", C.c_int32), ("stride_x", C.c_int32), ("pad_top", C.c_int32), ("pad_left", C.c_int32),
                ("up_shift", C.c_int32), ("off_y", C.c_int32), ("off_x", C.c_int32)]


class RunInfo(C.Structure):
    """sbbseg_run_info (include/sbbseg.h)"""
    _fields_ = [("box_xywh", C.c_int32 * 4), ("box_pixels", C.c_int64), ("otsu_threshold", C.c_int32), ("regions_ok", C.c_int32),
                ("text_present", C.c_int32), ("textlines_ok", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("n_src", C.c_int32), ("src", ConvSrc * 2), ("cout", C.c_int32),
                ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("out_stride_y", C.c_int32), ("out_stride_x", C.c_int32), ("out_off_y", C.c_int32), ("out_off_x", C.c_int32),
                ("out_tensor", C.c_int32), ("relu", C.c_int32), ("residual_tensor", C.c_int32),
                ("raw_out_tensor", C.c_int32), ("head_classes", C.c_int32), ("algorithmic_macs", C.c_double)]


_lib = None


def load_library(path: Optional[str] = None):
    """Load libsbbseg.so (no GPU needed for loading).  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -m sbb_textline_detection_amd._build` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(p)
    lib.sbbseg_last_error.restype = C.c_char_p
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    sigs = {
        "sbbseg_abi_version": [],
        "sbbseg_device_count": [C.POINTER(C.c_int)],
        "sbbseg_create": [i32, i32, C.POINTER(vp)],
        "sbbseg_destroy": [vp],
        "sbbseg_model_load": [vp, C.c_size_t, i32, i32, i32, i32, C.POINTER(vp)],
        "sbbseg_model_load_file": [C.c_char_p, i32, i32, i32, i32, C.POINTER(vp)],
        "sbbseg_debug_plan_summary": [vp, C.c_size_t, i32, i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)],
        "sbbseg_set_stream": [vp, vp],
        "sbbseg_synchronize": [vp],
        "sbbseg_set_lanes": [vp, i32],
        "sbbseg_set_label_channels": [vp, i32],
        "sbbseg_set_dedupe": [vp, i32],
        "sbbseg_set_ksplit": [vp, i32],
        "sbbseg_set_input": [vp, i32, i32, i32],
        "sbbseg_input_form": [vp, i32, i32, C.POINTER(C.c_int)],
        "sbbseg_add_tensor": [vp, i32, i32, i32, C.POINTER(C.c_int)],
        "sbbseg_add_conv": [vp, C.POINTER(ConvDesc)] + [vp] * 9,
        "sbbseg_add_maxpool": [vp, i32, i32, i32, i32, vp, vp, i32],
        "sbbseg_add_head": [vp, i32, i32, i32, vp, vp, vp],
        "sbbseg_add_tail": [vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, C.c_double],
        "sbbseg_finalize": [vp, i32],
        "sbbseg_model_info": [vp] + [C.POINTER(C.c_int)] * 4,
        "sbbseg_num_ops": [vp, C.POINTER(C.c_int)],
        "sbbseg_op_info": [vp, i32, C.c_char_p, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "sbbseg_op_issued_flops": [vp, i32, C.POINTER(C.c_double)],
        "sbbseg_device_bytes": [vp, C.POINTER(C.c_size_t)],
        "sbbseg_predict": [vp, vp, i32, vp],
        "sbbseg_segment_page": [vp, vp, i32, i32, vp],
        "sbbseg_segment_page_dev": [vp, vp, i32, i32, vp],
        "sbbseg_segment_pages_dev": [vp, i32, vp, i32, i32, vp],
        "sbbseg_segment_page_scaled": [vp, vp, i32, i32, i32, i32, vp],
        "sbbseg_segment_page_otsu": [vp, vp, i32, i32, i32, i32, vp, C.POINTER(C.c_int)],
        "sbbseg_debug_largest_contour": [vp, i32, i32, vp, C.POINTER(C.c_int64)],
        "sbbseg_debug_counter": [vp, i32, C.POINTER(C.c_int64)],
        "sbbseg_comm_unique_id": [C.c_char_p],
        "sbbseg_comm_init": [vp, i32, i32, C.c_char_p],
        "sbbseg_comm_info": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "sbbseg_comm_destroy": [vp],
        "sbbseg_allgather_labels_dev": [vp, vp, C.c_size_t, vp],
        "sbbseg_segment_crop": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, C.POINTER(C.c_int)],
        "sbbseg_segment_crop_dev": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
        "sbbseg_otsu_dev": [vp, vp, i32, i32, vp],
        "sbbseg_segment_tile_range_bin_dev": [vp, vp, i32, i32, i32, i32, vp, vp],
        "sbbseg_segment_whole": [vp, vp, i32, i32, i32, i32, vp],
        "sbbseg_segment_whole_scaled": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
        "sbbseg_tile_grid": [i32, i32, i32, i32, vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "sbbseg_nearest_map": [i32, i32, vp],
        "sbbseg_segment_tiles_dev": [vp, vp, i32, i32, vp, i32, vp],
        "sbbseg_segment_tile_range_dev": [vp, vp, i32, i32, i32, i32, vp],
        "sbbseg_stitch_dev": [vp, vp, i32, i32, vp],
        "sbbseg_debug_ingest": [vp, vp, i32, i32, vp, i32, i32, vp, C.c_size_t],
        "sbbseg_debug_read_tensor": [vp, i32, i32, vp, C.c_size_t],
        "sbbseg_debug_set_conv_variant": [vp, i32],
        "sbbseg_morph_dev": [vp, vp, i32, i32, i32, i32, i32, vp],
        "sbbseg_morph": [vp, vp, i32, i32, i32, i32, i32, vp],
        "sbbseg_page_box_dev": [vp, vp, i32, i32, vp, C.POINTER(C.c_int64)],
        "sbbseg_extract_page_box": [vp, vp, i32, i32, i32, i32, vp, vp, C.POINTER(C.c_int64)],
        "sbbseg_extract_page_box_dev": [vp, vp, i32, i32, i32, i32, vp, vp, C.POINTER(C.c_int64)],
        "sbbseg_segment_pages": [vp, i32, vp, i32, i32, vp],
        "sbbseg_deskew_side": [i32, i32, C.POINTER(C.c_int)],
        "sbbseg_rotation_matrix": [C.c_double, C.c_double, C.c_double, vp],
        "sbbseg_deskew_profiles_dev": [vp, vp, i32, i32, vp, vp, i32, vp],
        "sbbseg_deskew_profiles": [vp, vp, i32, i32, vp, vp, i32, vp],
        "sbbseg_debug_inject_alloc_failure": [i32],
        "sbbseg_profile_enable": [vp, i32],
        "sbbseg_profile_reset": [vp],
        "sbbseg_profile_get": [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
        "sbbseg_text_regions_present_dev": [vp, vp, i32, i32, i32, C.c_double, C.POINTER(C.c_int)],
        "sbbseg_debug_largest_contour_area2": [vp, i32, i32, C.POINTER(C.c_int64)],
        "sbbseg_run_page": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(RunInfo)],
        "sbbseg_device_alloc": [vp, C.c_size_t, C.POINTER(vp)],
        "sbbseg_device_free": [vp, vp],
        "sbbseg_upload": [vp, vp, vp, C.c_size_t],
        "sbbseg_download": [vp, vp, vp, C.c_size_t],
        "sbbseg_download_labels": [vp, vp, vp, C.c_size_t, i32],
        "sbbseg_set_owned_regions": [vp, i32],
        "sbbseg_owned_region_info": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "sbbseg_op_executed": [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "sbbseg_debug_owned_range": [i32, i32, i32, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "sbbseg_debug_region_rows": [i32, i32, i32, i32, i32, i32, vp, vp],
        "sbbseg_debug_poison_activations": [vp, i32],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    if lib.sbbseg_abi_version() != 5:
        raise RuntimeError("libsbbseg ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = "libsbbseg"):
    if rc != 0:
        msg = load_library().sbbseg_last_error()
        raise RuntimeError(f"{what}: {msg.decode('utf-8', 'replace') if msg else 'unknown error'}")


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """Owns one sbbseg_ctx (one GPU, one plan)."""

    def __init__(self, device: int = 0, precision: int = PREC_BF16, _handle=None):
        self.lib = load_library()
        if _handle is None:
            h = C.c_void_p()
            check(self.lib.sbbseg_create(device, precision, C.byref(h)), "sbbseg_create")
        else:
            h = _handle
        self.h = h
        self.device, self.precision = device, precision
        self.tensor_ids = []

    @classmethod
    def from_sbbw(cls, source, device: int, precision: int, max_batch: int, flags: int = 0) -> "Context":
        """One-call load through the library's own graph reader + planner (sbbseg_model_load[_file]): ``source`` is a path or the
        container's bytes.  The handle comes back finalized."""
        lib = load_library()
        h = C.c_void_p()
        if isinstance(source, (bytes, bytearray, memoryview)):
            buf = bytes(source)
            check(lib.sbbseg_model_load(buf, len(buf), device, precision, int(max_batch), int(flags), C.byref(h)), "sbbseg_model_load")
        else:
            check(lib.sbbseg_model_load_file(os.fsencode(source), device, precision, int(max_batch), int(flags), C.byref(h)), "sbbseg_model_load_file")
        return cls(device, precision, _handle=h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.sbbseg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plan upload ---------------------------------------------------------------------------
    def load_plan(self, plan, max_batch: int):
        lib, h = self.lib, self.h
        check(lib.sbbseg_set_input(h, plan.in_h, plan.in_w, 3))
        ids = []
        for t in plan.tensors:
            tid = C.c_int(-1)
            if t.kind == "input_c8":
                check(lib.sbbseg_input_form(h, INPUT_C8, 0, C.byref(tid)))
            elif t.kind == "input_pairs":
                check(lib.sbbseg_input_form(h, INPUT_PAIRS, t.pad, C.byref(tid)))
            elif t.kind == "unused":
                pass                                   # e.g. the last conv's output when the head is fused
            else:
                check(lib.sbbseg_add_tensor(h, t.H, t.W, t.C, C.byref(tid)))
            ids.append(tid.value)
        self.tensor_ids = ids
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        for s in plan.steps:
            if s.kind == "conv":
                d = ConvDesc()
                d.n_src = len(s.srcs)
                keep = []
                for k, g in enumerate(s.srcs):
                    d.src[k] = ConvSrc(ids[g.tensor], g.channels, g.kh, g.kw, g.stride_y, g.stride_x, g.pad_top,
                                       g.pad_left, g.shift, g.off_y, g.off_x)
                    keep.append(f32(g.w))
                d.cout, d.out_h, d.out_w = s.cout, s.out_h, s.out_w
                d.out_stride_y, d.out_stride_x = s.out_stride
                d.out_off_y, d.out_off_x = s.out_off
                d.out_tensor = ids[s.out] if s.out >= 0 else -1
                d.relu = int(s.relu)
                d.residual_tensor = ids[s.residual] if s.residual >= 0 else -1
                d.raw_out_tensor = ids[s.raw_out] if s.raw_out >= 0 else -1
                d.head_classes = s.head.classes if s.head is not None else 0
                d.algorithmic_macs = float(s.algorithmic_macs)
                sc, sh, rs, rb = f32(s.scale), f32(s.shift), f32(s.raw_scale), f32(s.raw_shift)
                hw = hs = hb = None
                if s.head is not None:
                    hw, hs, hb = f32(s.head.w), f32(s.head.scale), f32(s.head.shift)
                check(lib.sbbseg_add_conv(h, C.byref(d), _ptr(keep[0]), _ptr(keep[1]) if len(keep) > 1 else None,
                                          _ptr(sc), _ptr(sh), _ptr(rs), _ptr(rb), _ptr(hw), _ptr(hs), _ptr(hb)),
                      f"sbbseg_add_conv({s.name})")
            elif s.kind == "tail":
                arrs = [f32(s.w_src0), f32(s.w_img), f32(s.scale), f32(s.shift), f32(s.head.w), f32(s.head.scale), f32(s.head.shift)]
                check(lib.sbbseg_add_tail(h, ids[s.src0], ids[s.img], _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]),
                                          s.head.classes, _ptr(arrs[4]), _ptr(arrs[5]), _ptr(arrs[6]), float(s.algorithmic_macs)),
                      f"sbbseg_add_tail({s.name})")
            elif s.kind == "maxpool":
                ps, pb = f32(s.pre_scale), f32(s.pre_shift)
                check(lib.sbbseg_add_maxpool(h, ids[s.src], ids[s.dst], s.k, s.stride, _ptr(ps), _ptr(pb), int(s.pre_relu)),
                      f"sbbseg_add_maxpool({s.name})")
            elif s.kind == "head":
                w = np.ascontiguousarray(s.w, np.float32)
                sc, sh = np.ascontiguousarray(s.scale, np.float32), np.ascontiguousarray(s.shift, np.float32)
                check(lib.sbbseg_add_head(h, ids[s.src], s.cin, s.classes, _ptr(w), _ptr(sc), _ptr(sh)),
                      f"sbbseg_add_head({s.name})")
            else:
                raise RuntimeError(f"unknown plan step {s.kind}")
        check(lib.sbbseg_finalize(h, int(max_batch)), "sbbseg_finalize")

    # -- queries -------------------------------------------------------------------------------
    def model_info(self):
        v = [C.c_int() for _ in range(4)]
        check(self.lib.sbbseg_model_info(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)          # H, W, classes, max_batch

    def ops(self):
        n = C.c_int()
        check(self.lib.sbbseg_num_ops(self.h, C.byref(n)))
        out = []
        for i in range(n.value):
            buf = C.create_string_buffer(128)
            fl, by = C.c_double(), C.c_double()
            check(self.lib.sbbseg_op_info(self.h, i, buf, 128, C.byref(fl), C.byref(by)))
            iss = C.c_double()
            check(self.lib.sbbseg_op_issued_flops(self.h, i, C.byref(iss)))
            out.append({"index": i, "name": buf.value.decode(), "flops": fl.value, "issued_flops": iss.value, "min_bytes": by.value})
        return out

    def device_bytes(self) -> int:
        b = C.c_size_t()
        check(self.lib.sbbseg_device_bytes(self.h, C.byref(b)))
        return b.value

    def set_stream(self, stream_ptr: int):
        """Run on the caller's HIP stream (0 = the legacy default stream); -1 = the handle's own stream."""
        check(self.lib.sbbseg_set_stream(self.h, C.c_void_p(stream_ptr if stream_ptr >= 0 else 2 ** 64 - 1)))

    def synchronize(self):
        check(self.lib.sbbseg_synchronize(self.h))

    def set_lanes(self, lanes: int):
        """1 = one stream; 2 (default) = chunks of >= 16 tiles run as two concurrent halves (see sbbseg.h)."""
        check(self.lib.sbbseg_set_lanes(self.h, int(lanes)), "sbbseg_set_lanes")

    def set_dedupe(self, on: bool):
        """Fused page paths compute a repeated clamped tile once (default on; same label map).  See sbbseg.h."""
        check(self.lib.sbbseg_set_dedupe(self.h, int(bool(on))), "sbbseg_set_dedupe")

    def set_ksplit(self, on: bool):
        """Whole-image branch: split the K range of one-patch launches over the idle CUs (default on; see sbbseg.h)."""
        check(self.lib.sbbseg_set_ksplit(self.h, int(bool(on))), "sbbseg_set_ksplit")

    def set_owned_regions(self, mode: int):
        """Decoder launches over the region the page stitch keeps of every tile (+ halo) only: 0 = off, 1 = fused page paths
        (default), 2 = the tile-range entry points too (their tile labels are then defined on the owned regions only).  See sbbseg.h."""
        check(self.lib.sbbseg_set_owned_regions(self.h, int(mode)), "sbbseg_set_owned_regions")

    def owned_region_levels(self) -> int:
        """Decoder levels this plan runs as owned-region launches (0: everything is computed whole)."""
        return self.owned_region_info()[1]

    def owned_region_info(self):
        """(mode in force, decoder levels the plan runs as owned-region launches)."""
        mode, levels = C.c_int(0), C.c_int(0)
        check(self.lib.sbbseg_owned_region_info(self.h, C.byref(mode), C.byref(levels)), "sbbseg_owned_region_info")
        return int(mode.value), int(levels.value)

    def poison_activations(self, byte_value: int = 0xFF):
        """Test hook: fill every activation buffer and the tile-label scratch with a byte (0xFF = NaN in every 16-bit format)."""
        check(self.lib.sbbseg_debug_poison_activations(self.h, int(byte_value)), "sbbseg_debug_poison_activations")

    def forwards(self) -> int:
        """Patches run through the plan on this handle so far."""
        v = C.c_int64(0)
        check(self.lib.sbbseg_debug_counter(self.h, 1, C.byref(v)), "sbbseg_debug_counter")
        return int(v.value)

    def _label_out(self, h: int, w: int, channels: int) -> np.ndarray:
        """Output array for a host label map: one plane, or the reference's 3 identical channels (replicated on the
        device: numpy needs ~12 ms to do that for a 3500x2500 page, more than the whole forward pass)."""
        if channels not in (1, 3):
            raise ValueError("channels must be 1 or 3")
        check(self.lib.sbbseg_set_label_channels(self.h, channels), "sbbseg_set_label_channels")
        return np.empty((h, w) if channels == 1 else (h, w, 3), np.uint8)

    # -- execution -----------------------------------------------------------------------------
    def predict(self, x: np.ndarray) -> np.ndarray:
        H, W, classes, _ = self.model_info()
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 4 or x.shape[1:] != (H, W, 3):
            raise ValueError(f"predict expects [n,{H},{W},3], got {x.shape}")
        out = np.empty((x.shape[0], H, W, classes), np.float32)
        check(self.lib.sbbseg_predict(self.h, _ptr(x), x.shape[0], _ptr(out)), "sbbseg_predict")
        return out

    def segment_page(self, page: np.ndarray, channels: int = 1) -> np.ndarray:
        page = np.ascontiguousarray(page, np.uint8)
        if page.ndim != 3 or page.shape[2] != 3:
            raise ValueError(f"page must be uint8 [H,W,3], got {page.shape}")
        out = self._label_out(page.shape[0], page.shape[1], channels)
        check(self.lib.sbbseg_segment_page(self.h, _ptr(page), page.shape[0], page.shape[1], _ptr(out)), "sbbseg_segment_page")
        return out

    def segment_page_scaled(self, page: np.ndarray, out_h: int, out_w: int, channels: int = 1) -> np.ndarray:
        page = np.ascontiguousarray(page, np.uint8)
        out = self._label_out(out_h, out_w, channels)
        check(self.lib.sbbseg_segment_page_scaled(self.h, _ptr(page), page.shape[0], page.shape[1], out_h, out_w, _ptr(out)),
              "sbbseg_segment_page_scaled")
        return out

    def segment_page_otsu(self, page: np.ndarray, out_h: int = 0, out_w: int = 0, channels: int = 1):
        """extract_text_regions (main.py:439-447) in one call: (optional nearest rescale to out_h x out_w) +
        otsu_copy + do_prediction(patches=True).  Returns (labels uint8 [out_h, out_w], Otsu threshold)."""
        page = np.ascontiguousarray(page, np.uint8)
        if page.ndim != 3 or page.shape[2] != 3:
            raise ValueError(f"page must be uint8 [H,W,3], got {page.shape}")
        out_h, out_w = (out_h or page.shape[0]), (out_w or page.shape[1])
        out = self._label_out(out_h, out_w, channels)
        thr = C.c_int(0)
        check(self.lib.sbbseg_segment_page_otsu(self.h, _ptr(page), page.shape[0], page.shape[1], out_h, out_w, _ptr(out),
                                                C.byref(thr)), "sbbseg_segment_page_otsu")
        return out, int(thr.value)

    def segment_crop(self, page: np.ndarray, scaled_h: int, scaled_w: int, box, binarise: bool = False, channels: int = 1):
        """A patch stage on extract_page's cropped page (main.py:2061-2102): ``page`` = the stored image, ``scaled_*`` its size
        after get_image_and_scales, ``box`` = (x, y, w, h) on the upscaled page.  Returns (labels [h][w] (x3), Otsu threshold or
        None); ``binarise`` = the layout stage's otsu_copy on the crop."""
        page = np.ascontiguousarray(page, np.uint8)
        x, y, w, h = (int(v) for v in box)
        out = self._label_out(h, w, channels)
        thr = C.c_int(0)
        check(self.lib.sbbseg_segment_crop(self.h, _ptr(page), page.shape[0], page.shape[1], int(scaled_h), int(scaled_w), x, y, w, h,
                                           1 if binarise else 0, _ptr(out), C.byref(thr)), "sbbseg_segment_crop")
        return out, (int(thr.value) if binarise else None)

    def segment_crop_dev(self, d_page: int, Hs: int, Ws: int, scaled_h: int, scaled_w: int, box, binarise: bool, d_labels: int, d_threshold: int = 0):
        x, y, w, h = (int(v) for v in box)
        check(self.lib.sbbseg_segment_crop_dev(self.h, C.c_void_p(d_page), Hs, Ws, int(scaled_h), int(scaled_w), x, y, w, h,
                                               1 if binarise else 0, C.c_void_p(d_labels), C.c_void_p(d_threshold or None)), "sbbseg_segment_crop_dev")

    # -- the sharded path's collective on RCCL, inside the library (no torch.distributed needed) ------------------
    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """Join the communicator `unique_id` (from :func:`comm_unique_id` on rank 0) as `rank` of `world`; collective."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        check(self.lib.sbbseg_comm_init(self.h, int(rank), int(world), C.create_string_buffer(bytes(unique_id), 128)), "sbbseg_comm_init")

    def comm_info(self):
        r, w = C.c_int(0), C.c_int(0)
        check(self.lib.sbbseg_comm_info(self.h, C.byref(r), C.byref(w)), "sbbseg_comm_info")
        return int(r.value), int(w.value)

    def comm_destroy(self):
        check(self.lib.sbbseg_comm_destroy(self.h), "sbbseg_comm_destroy")

    def allgather_labels_dev(self, d_send: int, bytes_per_rank: int, d_recv: int):
        """All-gather of u8 label maps over RCCL on the handle's stream: d_recv = [world][bytes_per_rank]."""
        check(self.lib.sbbseg_allgather_labels_dev(self.h, C.c_void_p(d_send), int(bytes_per_rank), C.c_void_p(d_recv)), "sbbseg_allgather_labels_dev")

    def otsu_dev(self, d_page: int, Hp: int, Wp: int, d_threshold: int):
        check(self.lib.sbbseg_otsu_dev(self.h, C.c_void_p(d_page), Hp, Wp, C.c_void_p(d_threshold)), "sbbseg_otsu_dev")

    def segment_tile_range_bin_dev(self, d_page: int, Hp: int, Wp: int, first: int, n: int, d_threshold: int, d_tile_labels: int):
        check(self.lib.sbbseg_segment_tile_range_bin_dev(self.h, C.c_void_p(d_page), Hp, Wp, first, n, C.c_void_p(d_threshold),
                                                         C.c_void_p(d_tile_labels)), "sbbseg_segment_tile_range_bin_dev")

    def segment_whole(self, page: np.ndarray, out_h: int, out_w: int, channels: int = 1) -> np.ndarray:
        page = np.ascontiguousarray(page, np.uint8)
        out = self._label_out(out_h, out_w, channels)
        check(self.lib.sbbseg_segment_whole(self.h, _ptr(page), page.shape[0], page.shape[1], out_h, out_w, _ptr(out)),
              "sbbseg_segment_whole")
        return out

    def segment_whole_scaled(self, page: np.ndarray, scaled_h: int, scaled_w: int, out_h: int, out_w: int, channels: int = 1) -> np.ndarray:
        """Whole-image branch on the stored page as if it had first been nearest-upscaled to scaled_h x scaled_w."""
        page = np.ascontiguousarray(page, np.uint8)
        out = self._label_out(out_h, out_w, channels)
        check(self.lib.sbbseg_segment_whole_scaled(self.h, _ptr(page), page.shape[0], page.shape[1], scaled_h, scaled_w, out_h, out_w,
                                                   _ptr(out)), "sbbseg_segment_whole_scaled")
        return out

    def segment_page_dev(self, d_page: int, Hp: int, Wp: int, d_labels: int):
        check(self.lib.sbbseg_segment_page_dev(self.h, C.c_void_p(d_page), Hp, Wp, C.c_void_p(d_labels)), "sbbseg_segment_page_dev")

    def segment_pages_dev(self, d_pages, Hp: int, Wp: int, d_labels):
        """Equally sized pages (device pointers) -> their label maps (device pointers), tiles pooled across pages."""
        n = len(d_pages)
        if n != len(d_labels) or n == 0:
            raise ValueError("need as many label buffers as pages (at least one)")
        pa = (C.c_void_p * n)(*[C.c_void_p(int(v)) for v in d_pages])
        la = (C.c_void_p * n)(*[C.c_void_p(int(v)) for v in d_labels])
        check(self.lib.sbbseg_segment_pages_dev(self.h, n, pa, Hp, Wp, la), "sbbseg_segment_pages_dev")

    def segment_tile_range_dev(self, d_page: int, Hp: int, Wp: int, first: int, n: int, d_tile_labels: int):
        check(self.lib.sbbseg_segment_tile_range_dev(self.h, C.c_void_p(d_page), Hp, Wp, first, n, C.c_void_p(d_tile_labels)),
              "sbbseg_segment_tile_range_dev")

    def segment_tiles_dev(self, d_page: int, Hp: int, Wp: int, tile_xy: np.ndarray, d_tile_labels: int):
        xy = np.ascontiguousarray(tile_xy, np.int32)
        check(self.lib.sbbseg_segment_tiles_dev(self.h, C.c_void_p(d_page), Hp, Wp, _ptr(xy), xy.shape[0], C.c_void_p(d_tile_labels)),
              "sbbseg_segment_tiles_dev")

    def stitch_dev(self, d_tile_labels: int, Hp: int, Wp: int, d_labels: int):
        check(self.lib.sbbseg_stitch_dev(self.h, C.c_void_p(d_tile_labels), Hp, Wp, C.c_void_p(d_labels)), "sbbseg_stitch_dev")

    def debug_ingest(self, page: np.ndarray, tile_xy: np.ndarray, form: int, shape) -> np.ndarray:
        page = np.ascontiguousarray(page, np.uint8)
        xy = np.ascontiguousarray(tile_xy, np.int32)
        out = np.empty((xy.shape[0],) + tuple(shape), np.float32)
        check(self.lib.sbbseg_debug_ingest(self.h, _ptr(page), page.shape[0], page.shape[1], _ptr(xy), xy.shape[0], form,
                                           _ptr(out), out.size), "sbbseg_debug_ingest")
        return out

    def debug_read_tensor(self, plan_tensor: int, n: int, shape) -> np.ndarray:
        if not self.tensor_ids and getattr(self, "ids_provider", None) is not None:
            self.ids_provider()                       # natively planned handle: ids re-derived from the mirror plan (model.py)
        out = np.empty((n,) + tuple(shape), np.float32)
        check(self.lib.sbbseg_debug_read_tensor(self.h, self.tensor_ids[plan_tensor], n, _ptr(out), out.size),
              "sbbseg_debug_read_tensor")
        return out

    # -- stage glue ---------------------------------------------------------------
    def morph(self, plane: np.ndarray, op: int, ksize: int = 5, iterations: int = 1) -> np.ndarray:
        """cv2.erode (op 0) / cv2.dilate (op 1) of a uint8 plane [H,W] with a ksize x ksize kernel of ones, on the device."""
        plane = np.ascontiguousarray(plane, np.uint8)
        if plane.ndim != 2:
            raise ValueError("morph expects a uint8 plane [H, W]")
        out = np.empty_like(plane)
        check(self.lib.sbbseg_morph(self.h, _ptr(plane), plane.shape[0], plane.shape[1], int(op), int(ksize), int(iterations), _ptr(out)),
              "sbbseg_morph")
        return out

    def morph_dev(self, d_src: int, H: int, W: int, op: int, ksize: int, iterations: int, d_dst: int):
        check(self.lib.sbbseg_morph_dev(self.h, C.c_void_p(d_src), H, W, int(op), int(ksize), int(iterations), C.c_void_p(d_dst)), "sbbseg_morph_dev")

    # -- device buffers (callers without torch: stages.InferenceStages.run) ------------------------------------------
    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(self.lib.sbbseg_device_alloc(self.h, int(nbytes), C.byref(p)), "sbbseg_device_alloc")
        return int(p.value)

    def device_free(self, d_ptr: int):
        if d_ptr and getattr(self, "h", None):
            check(self.lib.sbbseg_device_free(self.h, C.c_void_p(d_ptr)), "sbbseg_device_free")

    def upload(self, d_dst: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        check(self.lib.sbbseg_upload(self.h, C.c_void_p(d_dst), _ptr(a), a.nbytes), "sbbseg_upload")

    def download(self, d_src: int, shape, dtype=np.uint8) -> np.ndarray:
        out = np.empty(shape, dtype)
        check(self.lib.sbbseg_download(self.h, _ptr(out), C.c_void_p(d_src), out.nbytes), "sbbseg_download")
        return out

    def download_labels(self, d_labels: int, h: int, w: int, channels: int = 1) -> np.ndarray:
        out = np.empty((h, w, 3) if channels == 3 else (h, w), np.uint8)
        check(self.lib.sbbseg_download_labels(self.h, _ptr(out), C.c_void_p(d_labels), h * w, channels), "sbbseg_download_labels")
        return out

    def text_regions_present_dev(self, d_regions: int, H: int, W: int, label: int = 1, min_area: float = 0.00001) -> bool:
        """get_text_region_contours_and_boxes' `len(contours) > 0` (main.py:456-480, 2083-2096) on a device label plane."""
        present = C.c_int(0)
        check(self.lib.sbbseg_text_regions_present_dev(self.h, C.c_void_p(d_regions), H, W, label, float(min_area), C.byref(present)),
              "sbbseg_text_regions_present_dev")
        return bool(present.value)

    def text_regions_present(self, regions: np.ndarray, label: int = 1, min_area: float = 0.00001) -> bool:
        plane = np.ascontiguousarray(regions[:, :, 0] if regions.ndim == 3 else regions, np.uint8)
        # staging buffer cached on the Context (grown on demand, released with the handle): sbbseg_device_free synchronises the device
        cap = getattr(self, "_gate_cap", 0)
        if cap < plane.size:
            if cap:
                self.device_free(self._gate_buf)
            self._gate_buf, self._gate_cap = self.device_alloc(plane.size), plane.size
        self.upload(self._gate_buf, plane)
        return self.text_regions_present_dev(self._gate_buf, plane.shape[0], plane.shape[1], label, min_area)

    def page_box_dev(self, d_mask: int, H: int, W: int):
        """((x, y, w, h), pixels) of the largest component of the dilated mask (main.py:394-404); pixels == 0: empty mask."""
        box = np.zeros(4, np.int32)
        px = C.c_int64(0)
        check(self.lib.sbbseg_page_box_dev(self.h, C.c_void_p(d_mask), H, W, _ptr(box), C.byref(px)), "sbbseg_page_box_dev")
        return tuple(int(v) for v in box), int(px.value)

    def host_contour_calls(self) -> int:
        v = C.c_int64(0)
        check(self.lib.sbbseg_debug_counter(self.h, 0, C.byref(v)), "sbbseg_debug_counter")
        return int(v.value)

    def extract_page_box(self, page: np.ndarray, scaled_h: int, scaled_w: int, channels: int = 1, want_mask: bool = True):
        """Border model on the page as upscaled to scaled_h x scaled_w + the page box, one call: (mask, (x, y, w, h), pixels).
        want_mask=False: the mask (a local of the reference's extract_page) stays on the device; returns (None, box, pixels)."""
        page = np.ascontiguousarray(page, np.uint8)
        mask = self._label_out(scaled_h, scaled_w, channels) if want_mask else None
        box = np.zeros(4, np.int32)
        px = C.c_int64(0)
        check(self.lib.sbbseg_extract_page_box(self.h, _ptr(page), page.shape[0], page.shape[1], scaled_h, scaled_w,
                                               _ptr(mask) if want_mask else None, _ptr(box), C.byref(px)), "sbbseg_extract_page_box")
        return mask, tuple(int(v) for v in box), int(px.value)

    def extract_page_box_dev(self, d_page: int, Hp: int, Wp: int, scaled_h: int, scaled_w: int, d_mask: int = 0):
        """The same on a page that is already on the device (pointer); d_mask = device buffer of scaled_h x scaled_w labels or 0.
        Returns ((x, y, w, h), pixels)."""
        box = np.zeros(4, np.int32)
        px = C.c_int64(0)
        check(self.lib.sbbseg_extract_page_box_dev(self.h, C.c_void_p(d_page), Hp, Wp, scaled_h, scaled_w, C.c_void_p(d_mask) if d_mask else None,
                                                   _ptr(box), C.byref(px)), "sbbseg_extract_page_box_dev")
        return tuple(int(v) for v in box), int(px.value)

    def segment_pages(self, pages, channels: int = 1):
        """do_prediction(patches=True) for a list of HOST pages of one size, upload / compute / download pipelined:
        list of uint8 label maps ([Hp][Wp], or [Hp][Wp][3] with channels=3)."""
        pages = [np.ascontiguousarray(p_, np.uint8) for p_ in pages]
        Hp, Wp = pages[0].shape[:2]
        if any(p_.shape != (Hp, Wp, 3) for p_ in pages):
            raise ValueError("segment_pages: all pages must be uint8 [Hp][Wp][3] of one size")
        outs = [self._label_out(Hp, Wp, channels) for _ in pages]
        n = len(pages)
        pin = (C.c_void_p * n)(*[p_.ctypes.data for p_ in pages])
        pout = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        check(self.lib.sbbseg_segment_pages(self.h, n, pin, Hp, Wp, pout), "sbbseg_segment_pages")
        return outs

    def deskew_profiles(self, mask: np.ndarray, angles_deg=None, matrices=None) -> np.ndarray:
        """Rotate-and-project of the deskew search: int32 [n_angles][side] row counts of the rotated, binarised region mask."""
        mask = np.ascontiguousarray(mask, np.uint8)
        H, W = mask.shape
        side = deskew_side(H, W)
        if matrices is not None:
            matrices = np.ascontiguousarray(matrices, np.float64).reshape(-1, 6)
            n = matrices.shape[0]
        else:
            angles_deg = np.ascontiguousarray(angles_deg, np.float64).reshape(-1)
            n = angles_deg.shape[0]
        counts = np.zeros((n, side), np.int32)
        check(self.lib.sbbseg_deskew_profiles(self.h, _ptr(mask), H, W, _ptr(matrices) if matrices is not None else None,
                                              _ptr(angles_deg) if matrices is None else None, n, _ptr(counts)), "sbbseg_deskew_profiles")
        return counts

    def set_conv_variant(self, variant: int):
        check(self.lib.sbbseg_debug_set_conv_variant(self.h, int(variant)))

    # -- profiling -----------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        check(self.lib.sbbseg_profile_enable(self.h, int(on)))

    def profile_reset(self):
        check(self.lib.sbbseg_profile_reset(self.h))

    def profile(self):
        out = []
        for i, op in enumerate(self.ops()):
            ms, ln, pt = C.c_double(), C.c_int64(), C.c_int64()
            check(self.lib.sbbseg_profile_get(self.h, i, C.byref(ms), C.byref(ln), C.byref(pt)))
            ex, tex = C.c_double(), C.c_double()
            check(self.lib.sbbseg_op_executed(self.h, i, C.byref(ex), C.byref(tex)))
            # exec_patches: whole-patch equivalents of the work launched since profile_reset (owned-region launches count the share of
            # the output grid they walk); timed_exec_patches: the same over the launches total_ms covers
            op.update(total_ms=ms.value, launches=ln.value, patches=pt.value, exec_patches=ex.value, timed_exec_patches=tex.value)
            out.append(op)
        return out


def host_largest_contour(mask: np.ndarray):
    """((x, y, w, h), pixels) by the library's exact HOST ranking (contour tracing) on an already dilated u8 mask; no GPU."""
    mask = np.ascontiguousarray(mask, np.uint8)
    box = np.zeros(4, np.int32)
    px = C.c_int64(0)
    check(load_library().sbbseg_debug_largest_contour(_ptr(mask), mask.shape[0], mask.shape[1], _ptr(box), C.byref(px)), "sbbseg_debug_largest_contour")
    return tuple(int(v) for v in box), int(px.value)


def run_page(border: "Context", layout: "Context", textline: "Context", page: np.ndarray, scaled_h: int, scaled_w: int,
             channels: int = 3, want_mask: bool = True):
    """The model-running part of run() (main.py:2056-2107) in one library call (``sbbseg_run_page``): the page is uploaded once and
    stays on the device for the border, layout and textline stages.  Returns (page mask or None, regions or None, textlines or None,
    RunInfo); regions / textlines are cut to the page box (h x w [x channels]).  An empty border mask raises like main.py:401."""
    page = np.ascontiguousarray(page, np.uint8)
    pix = scaled_h * scaled_w
    mask = np.empty((scaled_h, scaled_w, 3) if channels == 3 else (scaled_h, scaled_w), np.uint8) if want_mask else None
    regions = np.empty(pix * channels, np.uint8)
    lines = np.empty(pix, np.uint8)
    info = RunInfo()
    rc = border.lib.sbbseg_run_page(border.h, layout.h, textline.h, _ptr(page), page.shape[0], page.shape[1], scaled_h, scaled_w, channels,
                                    _ptr(mask), _ptr(regions), _ptr(lines), C.byref(info))
    if rc != 0:
        msg = (border.lib.sbbseg_last_error() or b"").decode("utf-8", "replace")
        if "empty sequence" in msg:
            raise ValueError("attempt to get argmax of an empty sequence")          # what main.py:401 raises
        raise RuntimeError("sbbseg_run_page: " + msg)
    w, h = int(info.box_xywh[2]), int(info.box_xywh[3])
    r = regions[:h * w * channels].reshape((h, w, 3) if channels == 3 else (h, w)).copy() if info.regions_ok else None
    t = lines[:h * w].reshape(h, w).copy() if info.textlines_ok else None
    return mask, r, t, info


def owned_range(extent: int, tile: int, margin: int, n_tiles: int, t: int):
    """[lo, hi) in tile coordinates of what the page stitch keeps of tile t of an axis (closed form of the owned-region launches; no GPU)."""
    lib = load_library()
    lo, hi = C.c_int(0), C.c_int(0)
    check(lib.sbbseg_debug_owned_range(int(extent), int(tile), int(margin), int(n_tiles), int(t), C.byref(lo), C.byref(hi)), "sbbseg_debug_owned_range")
    return int(lo.value), int(hi.value)


def region_rows(extent: int, tile: int, margin: int, n_tiles: int, t: int, level_sizes):
    """Rows [lo, hi) every decoder level must produce for tile t of an axis: one pair per level, level 0 = the network output (no GPU)."""
    lib = load_library()
    sizes = np.ascontiguousarray(level_sizes, dtype=np.int32)
    out = np.zeros((len(sizes), 2), dtype=np.int32)
    check(lib.sbbseg_debug_region_rows(int(extent), int(tile), int(margin), int(n_tiles), int(t), len(sizes), _ptr(sizes), _ptr(out)),
          "sbbseg_debug_region_rows")
    return out


def host_largest_contour_area2(mask: np.ndarray) -> int:
    """TWICE the largest outer-contour area (cv2.contourArea) of a u8 mask by the library's HOST tracer; no GPU."""
    mask = np.ascontiguousarray(mask, np.uint8)
    a2 = C.c_int64(0)
    check(load_library().sbbseg_debug_largest_contour_area2(_ptr(mask), mask.shape[0], mask.shape[1], C.byref(a2)), "sbbseg_debug_largest_contour_area2")
    return int(a2.value)


def comm_unique_id() -> bytes:
    """128 opaque bytes naming a new RCCL communicator (rank 0 calls this and ships them to every rank)."""
    buf = C.create_string_buffer(128)
    check(load_library().sbbseg_comm_unique_id(buf), "sbbseg_comm_unique_id")
    return buf.raw


def tile_grid(Hp: int, Wp: int, H: int, W: int):
    """(tile_xy int32 [n,2], nxf, nyf) from the library's restatement of main.py:246-281."""
    lib = load_library()
    nx, ny = C.c_int(), C.c_int()
    check(lib.sbbseg_tile_grid(Hp, Wp, H, W, None, 0, C.byref(nx), C.byref(ny)), "sbbseg_tile_grid")
    xy = np.empty((nx.value * ny.value, 2), np.int32)
    check(lib.sbbseg_tile_grid(Hp, Wp, H, W, _ptr(xy), xy.shape[0], None, None), "sbbseg_tile_grid")
    return xy, nx.value, ny.value


def deskew_side(H: int, W: int) -> int:
    """Side of the zero square the deskew search centres a region mask on: int(1.4 * max(H, W)) (main.py:1613)."""
    side = C.c_int(0)
    check(load_library().sbbseg_deskew_side(int(H), int(W), C.byref(side)), "sbbseg_deskew_side")
    return int(side.value)


def rotation_matrix(cx: float, cy: float, angle_deg: float) -> np.ndarray:
    """cv2.getRotationMatrix2D((cx, cy), angle, 1.0) as the library computes it: float64 [2][3]."""
    m = np.zeros(6, np.float64)
    check(load_library().sbbseg_rotation_matrix(float(cx), float(cy), float(angle_deg), _ptr(m)), "sbbseg_rotation_matrix")
    return m.reshape(2, 3)


def nearest_map(src_len: int, dst_len: int) -> np.ndarray:
    """int32 [dst_len]: the library's cv2.INTER_NEAREST index rule (sbbseg_nearest_map)."""
    out = np.empty(dst_len, np.int32)
    check(load_library().sbbseg_nearest_map(int(src_len), int(dst_len), _ptr(out)), "sbbseg_nearest_map")
    return out


def native_plan_summary(sbbw_bytes: bytes, precision: int, flags: int = 0) -> str:
    """Text summary of the plan the library's own planner builds from a .sbbw container (no GPU needed; test hook)."""
    lib = load_library()
    need = C.c_size_t(0)
    check(lib.sbbseg_debug_plan_summary(sbbw_bytes, len(sbbw_bytes), precision, flags, None, 0, C.byref(need)), "sbbseg_debug_plan_summary")
    buf = C.create_string_buffer(need.value)
    check(lib.sbbseg_debug_plan_summary(sbbw_bytes, len(sbbw_bytes), precision, flags, buf, need.value, None), "sbbseg_debug_plan_summary")
    return buf.value.decode()
