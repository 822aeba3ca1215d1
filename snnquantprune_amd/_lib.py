"""ctypes binding of libsnnqp.so (include/snnqp.h).

The product path has no CPU fallback: if the HIP library is missing or fails to
load, importing an op raises.  Build it with ``python __graft_entry__.py`` or
``python snnquantprune_amd/csrc/build.py``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int8,
                    c_int32, c_uint32, c_int64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsnnqp.so")

# include/snnqp.h SNNQP_VERSION the prototypes below were written against
ABI_VERSION = 502

# enums of include/snnqp.h
F32, U8, BITS, EV1, EV4 = 0, 1, 2, 3, 4
W_F32, W_I8 = 0, 1
NEURON_NONE, NEURON_MULTI_STEP_LIF, NEURON_PARAMETRIC_LEAKY_IF, NEURON_LIF = 0, 1, 2, 3
Q_DUQ, Q_UNIFORM_STATIC, Q_PARAMETRIC_D, Q_PARAMETRIC_D_XMAX = 0, 1, 2, 3
IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA = 0, 1, 2
FLAG_CODE_OVERFLOW, FLAG_MASK_NOT_BINARY = 1, 2
FLAG_NOT_INTEGER, FLAG_GT_ONE, FLAG_GT_127, FLAG_GT_15 = 4, 8, 16, 32
OK, EINVAL, EUNSUPPORTED, EHIP = 0, -1, -2, -3
STATUS_QUEUE_CORRUPT, STATUS_TICKET, STATUS_BOUND = 1, 2, 4


class WeightT(Structure):
  _fields_ = [("wtype", c_int32), ("w", c_void_p), ("L", c_float), ("m", c_float),
              ("abs_sum_max", c_int32), ("code_max", c_int32), ("col_sum", c_void_p),
              ("wt_fp6", c_void_p), ("min_current_bits", c_uint32), ("ch_stack_max", c_int32),
              ("ch_slots", c_void_p)]


BN_MEAN_ZERO, BN_BIAS_ZERO, BN_MUL_UNIFORM = 1, 2, 4


class BnT(Structure):
  _fields_ = [("mean", c_void_p), ("mul", c_void_p), ("bias", c_void_p), ("flags", c_int32)]


class NeuronT(Structure):
  _fields_ = [("kind", c_int32), ("k", c_float), ("v_threshold", c_float),
              ("v_reset", c_float), ("decay", c_void_p)]


class ConvGeomT(Structure):
  _fields_ = [(n, c_int32) for n in (
      "H", "W", "Cin", "Cout", "KH", "KW", "stride_h", "stride_w",
      "pad_h_lo", "pad_h_hi", "pad_w_lo", "pad_w_hi",
      "in_dil_h", "in_dil_w", "k_dil_h", "k_dil_w", "groups")]


class Conv3dGeomT(Structure):
  _fields_ = ([(n, c_int32) for n in ("D", "H", "W", "Cin", "Cout", "KD", "KH", "KW")]
              + [(n, c_int32 * 3) for n in ("stride", "pad_lo", "pad_hi", "in_dil", "k_dil")]
              + [("groups", c_int32)])


_PROTOTYPES = {
    "snnqp_version": (c_int, []),
    "snnqp_last_error": (c_char_p, []),
    "snnqp_build_flags": (c_char_p, []),
    "snnqp_conv_out_shape": (c_int, [POINTER(ConvGeomT), POINTER(c_int32),
                                     POINTER(c_int32)]),
    "snnqp_quantize": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_float,
                               c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snnqp_quantize_ex": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_float,
                                  c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snnqp_pack_codes_mfma": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                      c_void_p]),
    "snnqp_pack_codes_fp6": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "snnqp_inspect_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "snnqp_inspect_u8": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "snnqp_f32_to_u8": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "snnqp_narrow_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "snnqp_pack_frames": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "snnqp_unpack_frames": (c_int, [c_void_p, c_int, c_int64, c_int32, c_int32, c_void_p,
                                    c_void_p]),
    "snnqp_pack_bits": (c_int, [c_void_p, c_int, c_int64, c_int32, c_void_p,
                                c_void_p]),
    "snnqp_pack_bits_checked": (c_int, [c_void_p, c_int, c_int64, c_int32, c_void_p, c_void_p,
                                        c_void_p]),
    "snnqp_unpack_bits": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "snnqp_conv_forward": (c_int, [c_void_p, c_int, c_int64, POINTER(ConvGeomT),
                                   POINTER(WeightT), c_void_p, c_void_p, c_void_p]),
    "snnqp_conv_forward_if": (c_int, [c_void_p, c_void_p, c_int, c_int64, POINTER(ConvGeomT),
                                      POINTER(WeightT), c_void_p, c_void_p]),
    "snnqp_conv3d_out_shape": (c_int, [POINTER(Conv3dGeomT), POINTER(c_int32), POINTER(c_int32),
                                       POINTER(c_int32)]),
    "snnqp_conv3d_lif_forward": (c_int, [
        c_void_p, c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, POINTER(Conv3dGeomT),
        POINTER(WeightT), POINTER(BnT), POINTER(NeuronT), c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "snnqp_conv_gated_packed_bytes": (c_int64, [c_int32, c_int32]),
    "snnqp_pack_codes_gated": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "snnqp_conv_gated_packed_bytes_ex": (c_int64, [c_int32, c_int32, c_int32]),
    "snnqp_pack_codes_gated_ex": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "snnqp_conv_gated_forward": (c_int, [c_void_p, c_void_p, c_int64, POINTER(ConvGeomT),
                                         POINTER(WeightT), c_void_p, c_void_p, c_void_p]),
    "snnqp_dense_gated_packed_bytes": (c_int64, [c_int32, c_int32]),
    "snnqp_pack_codes_dense_gated": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "snnqp_dense_gated_packed_bytes_ex": (c_int64, [c_int32, c_int32, c_int32]),
    "snnqp_pack_codes_dense_gated_ex": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "snnqp_dense_gated_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                          POINTER(WeightT), c_void_p, c_void_p, c_void_p]),
    "snnqp_conv_lif_forward": (c_int, [
        c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, POINTER(ConvGeomT),
        POINTER(WeightT), c_void_p, POINTER(BnT), POINTER(NeuronT), c_void_p,
        c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "snnqp_conv_lif_forward_pred": (c_int, [
        c_void_p, c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, POINTER(ConvGeomT),
        POINTER(WeightT), c_void_p, POINTER(BnT), POINTER(NeuronT), c_void_p,
        c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "snnqp_pack_frames_checked": (c_int, [c_void_p, c_int, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                          c_void_p]),
    "snnqp_conv_lif_forward_if": (c_int, [
        c_void_p, c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, POINTER(ConvGeomT),
        POINTER(WeightT), POINTER(BnT), POINTER(NeuronT), c_void_p, c_void_p, c_void_p, c_int, c_int,
        c_void_p]),
    "snnqp_dense_lif_forward_if": (c_int, [
        c_void_p, c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32,
        POINTER(WeightT), POINTER(BnT), POINTER(NeuronT), c_void_p, c_void_p, c_void_p, c_int,
        c_void_p]),
    "snnqp_dense_lif_forward": (c_int, [
        c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32,
        POINTER(WeightT), c_void_p, POINTER(BnT), POINTER(NeuronT), c_void_p,
        c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "snnqp_dense_workspace_bytes": (c_int64, [c_int, c_int32, c_int32, c_int32, c_int32, POINTER(WeightT)]),
    "snnqp_dense_lif_forward_ws": (c_int, [
        c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32,
        POINTER(WeightT), c_void_p, POINTER(BnT), POINTER(NeuronT), c_void_p,
        c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "snnqp_dense_head_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "snnqp_dense_head_forward": (c_int, [
        c_void_p, c_int, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, POINTER(WeightT),
        c_void_p, POINTER(NeuronT), c_int32, POINTER(WeightT), c_void_p, POINTER(NeuronT), c_int32,
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "snnqp_device_status": (c_int, [c_int, POINTER(c_uint32), c_int]),
    "snnqp_workqueue_capture_mark": (c_int, [c_int, POINTER(c_int64)]),
    "snnqp_workqueue_capture_release": (c_int, [c_int, c_int64, c_int64]),
    "snnqp_workqueue_stats": (c_int, [POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), c_int]),
    "snnqp_debug_workqueue_poke": (c_int, [c_int, c_int64, c_int, c_uint32]),
    "snnqp_fallback_counts": (c_int, [POINTER(c_int64), POINTER(c_int64), c_char_p, c_int32, c_int]),
    "snnqp_conv_dequant_form": (c_int, [POINTER(WeightT), POINTER(NeuronT)]),
    "snnqp_current_min": (c_int, [POINTER(WeightT), POINTER(BnT), c_int32, c_int32, c_void_p, c_void_p]),
    "snnqp_lif_forward": (c_int, [c_void_p, c_int32, c_int64, c_int32, POINTER(BnT),
                                  POINTER(NeuronT), c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p]),
    "snnqp_batchnorm_forward": (c_int, [c_void_p, c_int64, c_int32, POINTER(BnT),
                                        c_void_p, c_void_p]),
    "snnqp_maxpool2x2": (c_int, [c_void_p, c_int, c_int64, c_int32, c_int32, c_int32,
                                 c_void_p, c_void_p]),
    "snnqp_spatial_mean": (c_int, [c_void_p, c_int, c_int64, c_int32, c_int32, c_void_p,
                                   c_void_p]),
    "snnqp_sigmoid_gate": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "snnqp_apply_gate": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int32, c_int32,
                                 c_void_p, c_void_p]),
    "snnqp_events_to_frames": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                       c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "snnqp_density": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "snnqp_vote": (c_int, [c_void_p, c_int, c_int32, c_int32, c_int32, c_int32,
                           c_void_p, c_void_p]),
    "snnqp_vote_if": (c_int, [c_void_p, c_void_p, c_int, c_int32, c_int32, c_int32, c_int32,
                              c_void_p, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


class SnnqpError(RuntimeError):
  def __init__(self, code, msg):
    super().__init__("libsnnqp error %d: %s" % (code, msg))
    self.code = code


def lib():
  """Loads libsnnqp.so once; raises if it is missing (no CPU fallback).

  SNNQP_DIAG_LIB=<path> loads a diagnostic build (tools/diag_build.py writes them under
  diag_build/, never over the in-tree library) instead, and says so on stderr; such a
  library reports its switches in snnqp_build_flags() and `require_product_build()`
  (tests, bench.py's default run) refuses it."""
  global _lib
  if _lib is None:
    path = os.environ.get("SNNQP_DIAG_LIB") or LIB_PATH
    if not os.path.exists(path):
      raise ImportError(
          "HIP extension %s not found: build it with `python __graft_entry__.py` "
          "(hipcc --offload-arch=gfx950); there is no CPU fallback." % path)
    import torch  # noqa: F401  (first: libsnnqp must share torch's HIP runtime)
    handle = ctypes.CDLL(path)
    if path != LIB_PATH:
      import sys
      print("snnquantprune_amd: DIAGNOSTIC library %s" % path, file=sys.stderr)
    for name, (res, args) in _PROTOTYPES.items():
      fn = getattr(handle, name)
      fn.restype = res
      fn.argtypes = args
    if handle.snnqp_version() != ABI_VERSION:
      raise ImportError(
          "%s has ABI version %d, this binding was written against %d (include/snnqp.h "
          "SNNQP_VERSION): rebuild it with `python __graft_entry__.py`"
          % (path, handle.snnqp_version(), ABI_VERSION))
    _lib = handle
  return _lib


def build_flags() -> str:
  return lib().snnqp_build_flags().decode("utf-8", "replace")


def require_product_build():
  """Raises unless the loaded library is the product build (no diagnostic switches)."""
  flags = build_flags()
  if flags or os.environ.get("SNNQP_DIAG_LIB"):
    raise RuntimeError("libsnnqp is a diagnostic build (%s): rebuild with "
                       "`python snnquantprune_amd/csrc/build.py --force`" % (flags or "SNNQP_DIAG_LIB"))


def check(rc):
  if rc != 0:
    raise SnnqpError(rc, lib().snnqp_last_error().decode("utf-8", "replace"))
