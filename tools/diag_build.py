"""Diagnostic / A-B builds of libsnnqp.so, kept OUT of the product tree.

  python tools/diag_build.py NAME [--patch tools/diag/x.patch ...] [--files a.hip,b.hip] -- -DFOO=1 ...

copies snnquantprune_amd/csrc/ to diag_build/NAME/src/, applies the patches there
(ablations and probes live as patches under tools/diag/, not as #if blocks in the product
kernels), compiles the listed files (default: every file a patch touched, plus api.hip)
with the extra flags and links diag_build/NAME/libsnnqp.so against the product objects of
the other files.  The library reports the switches in snnqp_build_flags(); load it with
SNNQP_DIAG_LIB=diag_build/NAME/libsnnqp.so (tests and the default bench refuse it).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snnquantprune_amd.csrc import build as product  # noqa: E402


def main(argv):
  if "--" in argv:
    k = argv.index("--")
    argv, flags = argv[:k], argv[k + 1:]
  else:
    flags = []
  name, patches, files = argv[0], [], []
  i = 1
  while i < len(argv):
    if argv[i] == "--patch":
      patches.append(os.path.abspath(argv[i + 1])); i += 2
    elif argv[i] == "--files":
      files += argv[i + 1].split(","); i += 2
    else:
      raise SystemExit("unknown argument %s" % argv[i])
  out = os.path.join(ROOT, "diag_build", name)
  src = os.path.join(out, "src")
  shutil.rmtree(out, ignore_errors=True)
  os.makedirs(src)
  for f in os.listdir(product.HERE):
    if f.endswith((".hip", ".h")):
      shutil.copy(os.path.join(product.HERE, f), src)
  # the sources include "../../include/snnqp.h": from diag_build/NAME/src that is diag_build/include
  inc = os.path.join(ROOT, "diag_build", "include")
  os.makedirs(inc, exist_ok=True)
  shutil.copy(os.path.join(ROOT, "include", "snnqp.h"), inc)
  touched = set()
  for p in patches:
    before = {f: open(os.path.join(src, f)).read() for f in os.listdir(src)}
    subprocess.check_call(["patch", "-p1", "-d", src, "-i", p])
    for f in os.listdir(src):
      if f in before and open(os.path.join(src, f)).read() != before[f]:
        touched.add(f)
  if any(f.endswith(".h") for f in touched):
    touched = set(product.SOURCES)
  todo = sorted((set(files) | {f for f in touched if f.endswith(".hip")} | {"api.hip"}))
  product.build(verbose=False)                      # product objects for the rest
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  desc = " ".join(flags + ["patch:" + os.path.basename(p) for p in patches]) or "diag:" + name
  objs, jobs = [], []
  for s in product.SOURCES:
    if s in todo:
      obj = os.path.join(out, s.replace(".hip", ".o"))
      jobs.append([hipcc, *product.FLAGS, "-w", *flags, "-DSNNQP_BUILD_FLAGS=\"%s\"" % desc,
                   "-c", os.path.join(src, s), "-o", obj])
    else:
      obj = os.path.join(product.HERE, s.replace(".hip", ".o"))
    objs.append(obj)
  with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(subprocess.check_call, jobs))
  lib = os.path.join(out, "libsnnqp.so")
  subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
  print(lib)


if __name__ == "__main__":
  main(sys.argv[1:])
