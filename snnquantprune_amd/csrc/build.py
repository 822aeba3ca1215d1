"""Builds libsnnqp.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numeric
contract: every float op of the reference keeps its own rounding.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "runtime.hip", "quantize.hip", "spikes.hip", "frames.hip", "elementwise.hip",
           "generic_block.hip", "blocks.hip", "conv3x3_u8c2.hip", "conv3x3_bits.hip",
           "dense_mfma.hip", "dense_wide.hip", "dense_fp6.hip", "fseq_gemm.hip", "conv_gated.hip", "dense_gated.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function"]
# -fno-slp-vectorize: the SLP pass pairs adjacent float ops of the unrolled
# epilogues into v_pk_*_f32, which on gfx950 is no faster than two scalar ops
# and chains every pair through one register pair with s_nop hazards.
LIB = os.path.join(HERE, "libsnnqp.so")


def _stale(out, deps):
  if not os.path.exists(out):
    return True
  t = os.path.getmtime(out)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
  """The product build: fixed flags, in-tree objects.  Diagnostic / A-B builds never come
  through here (tools/diag_build.py compiles them into diag_build/<name>/)."""
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  headers = [os.path.join(HERE, h) for h in ("common.h", "kernels.h", "conv_tile.h")]
  headers.append(os.path.join(HERE, "..", "..", "include", "snnqp.h"))
  headers.append(os.path.abspath(__file__))
  objs, jobs = [], []
  for s in SOURCES:
    src = os.path.join(HERE, s)
    obj = os.path.join(HERE, s.replace(".hip", ".o"))
    objs.append(obj)
    if force or _stale(obj, [src] + headers):
      jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])

  def run(cmd):
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)

  with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(run, jobs))
  if jobs or force or _stale(LIB, objs):
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
  return LIB


if __name__ == "__main__":
  build(force="--force" in sys.argv)
